"""Accumulation checkpoints and screenshots.

The reference has no restartable renders; its only persistence is the GUI's screenshot button (Gui.cs:28-33 ->
Framebuffer.cs:67-82: read the displayed RGBA8 image back, flip it vertically because GL rows are bottom-up, write a
PNG).  On top of pt_read_result / pt_write_result this module adds
  * save_checkpoint / load_checkpoint: the raw RGBA32F accumulation image + frame index + the parameters the image is
    only valid for, so that a long progressive render continues bit-identically after a restart;
  * save_screenshot: the tone-mapped RGBA8 image (pt_present_rgba8) as a PNG, flipped like the reference's.
File format (little endian): 8-byte magic "PTCKPT1\\0", int32 x 8 (width, image height, y0, rows, band_rows, band_world,
band_rank, frame index), int32 x 2 (ray depth, spp), float32 x 2 (focal length, aperture), then rows*width*4 float32.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

MAGIC = b"PTCKPT1\0"
_HEADER = struct.Struct("<8s8i2i2f")


class CheckpointError(ValueError):
    pass


def write_checkpoint_file(path, image: np.ndarray, *, width, height, y0, rows, band_rows=0, band_world=1, band_rank=0,
                          frame_index, ray_depth, spp, focal_length, aperture) -> None:
    img = np.ascontiguousarray(image, dtype=np.float32)
    if img.shape != (rows, width, 4):
        raise CheckpointError(f"image shape {img.shape} does not match rows x width x 4 = {(rows, width, 4)}")
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, width, height, y0, rows, band_rows, band_world, band_rank, frame_index, ray_depth, spp,
                             float(focal_length), float(aperture)))
        f.write(img.tobytes())


def read_checkpoint_file(path):
    """-> (header dict, (rows, width, 4) float32 image)"""
    with open(path, "rb") as f:
        raw = f.read(_HEADER.size)
        if len(raw) != _HEADER.size:
            raise CheckpointError("truncated checkpoint header")
        magic, width, height, y0, rows, band_rows, band_world, band_rank, frame, depth, spp, focal, aperture = _HEADER.unpack(raw)
        if magic != MAGIC:
            raise CheckpointError("not a mi355pt checkpoint (bad magic)")
        if width <= 0 or rows < 0 or frame < 0:
            raise CheckpointError("corrupt checkpoint header")
        data = f.read()
    if len(data) != rows * width * 16:
        raise CheckpointError(f"checkpoint payload is {len(data)} bytes, expected {rows * width * 16}")
    hdr = dict(width=width, height=height, y0=y0, rows=rows, band_rows=band_rows, band_world=band_world, band_rank=band_rank,
               frame_index=frame, ray_depth=depth, spp=spp, focal_length=focal, aperture=aperture)
    return hdr, np.frombuffer(data, dtype=np.float32).reshape(rows, width, 4).copy()


def save_checkpoint(path, tracer) -> None:
    """Dump `tracer`'s tile (all of the image on one GPU) with everything needed to validate a later load."""
    write_checkpoint_file(path, tracer.Result, width=tracer.Width, height=tracer.Height, y0=tracer.y0, rows=tracer.rows,
                          band_rows=tracer.band_rows, band_world=tracer.band_world, band_rank=tracer.band_rank,
                          frame_index=tracer.FrameIndex,
                          ray_depth=tracer.RayDepth, spp=tracer.SPP, focal_length=tracer.FocalLength,
                          aperture=tracer.ApertureDiameter)


def load_checkpoint(path, tracer, strict: bool = True) -> dict:
    """Restore image + frame index into `tracer`.  The tile geometry must match; with `strict` the integrator parameters
    must match too (an image accumulated with other parameters is not a sample of the same estimator)."""
    hdr, img = read_checkpoint_file(path)
    # the whole tile geometry must match: under block-cyclic ownership every rank has y0 = 0 and often the same row
    # count, so band height / world size / rank are what tell one rank's rows from another's
    geo = dict(width=tracer.Width, height=tracer.Height, y0=tracer.y0, rows=tracer.rows, band_rows=tracer.band_rows)
    if tracer.band_rows:
        geo.update(band_world=tracer.band_world, band_rank=tracer.band_rank)
    for k, v in geo.items():
        if hdr[k] != v:
            raise CheckpointError(f"checkpoint {k} = {hdr[k]} but the renderer has {v}")
    if strict:
        par = dict(ray_depth=tracer.RayDepth, spp=tracer.SPP)
        for k, v in par.items():
            if hdr[k] != v:
                raise CheckpointError(f"checkpoint {k} = {hdr[k]} but the renderer has {v} (pass strict=False to override)")
        if np.float32(hdr["focal_length"]) != np.float32(tracer.FocalLength) or np.float32(hdr["aperture"]) != np.float32(tracer.ApertureDiameter):
            raise CheckpointError("checkpoint lens parameters differ from the renderer's (pass strict=False to override)")
    tracer.WriteResult(img, hdr["frame_index"])
    return hdr


def encode_png(rgba8: np.ndarray, flip_vertically: bool = True) -> bytes:
    """Minimal PNG (8-bit RGB, no alpha: the reference saves the opaque displayed image).  `rgba8` is (H, W, 3|4) uint8 with
    row 0 = bottom of the image (GL order); flip_vertically=True writes it top-down like Framebuffer.cs:79."""
    img = np.ascontiguousarray(rgba8[..., :3], dtype=np.uint8)
    if flip_vertically:
        img = img[::-1]
    h, w = img.shape[:2]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def decode_png_rgb8(data: bytes) -> np.ndarray:
    """Inverse of encode_png for the files it writes (filter type 0 only) — used by the tests."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        (n,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(tag + body) & 0xFFFFFFFF), "PNG chunk CRC"
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 2)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3).copy()


def save_screenshot(path, tracer) -> None:
    """Gui.cs:28-33 screenshot: the displayed (tone-mapped) image as PNG, flipped vertically (Framebuffer.cs:67-82)."""
    with open(path, "wb") as f:
        f.write(encode_png(tracer.Present(), flip_vertically=True))
