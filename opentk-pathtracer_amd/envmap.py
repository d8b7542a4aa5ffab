"""Procedural environment cubemaps (synthetic stand-ins; the reference's JPEG faces are not redistributed).

The reference's sky box is 6 x 2048^2 SRGB8_ALPHA8 faces in the order posx, negx, posy, negy, posz, negz
(/root/reference/OpenTK-PathTracer/src/MainWindow.cs:177-187, loader src/Helper.cs:18-50); its default
environment is the 256^2 RGBA32F atmosphere cube (MainWindow.cs:174,189).  These generators produce cubes of
the same shape/format from closed-form functions of the texel-centre direction, so that any size is reproducible
from a seed on both the build container and the GPU box.
"""
from __future__ import annotations

import numpy as np

F = np.float32
FORMAT_RGBA32F = 0
FORMAT_SRGB8_A8 = 1


def face_directions(size: int) -> np.ndarray:
    """Texel-centre directions, shape (6, size, size, 3), GL cube face convention (GL 4.5 spec table 8.19):
    +X: (1,-t,-s)  -X: (-1,-t,s)  +Y: (s,1,t)  -Y: (s,-1,-t)  +Z: (s,-t,1)  -Z: (-s,-t,-1), s,t in [-1,1],
    row index <-> t, column index <-> s."""
    c = (np.arange(size, dtype=np.float64) + 0.5) / size * 2.0 - 1.0
    s, t = np.meshgrid(c, c)  # s varies along columns, t along rows
    one = np.ones_like(s)
    faces = [(one, -t, -s), (-one, -t, s), (s, one, t), (s, -one, -t), (s, -t, one), (-s, -t, -one)]
    d = np.stack([np.stack(f, axis=-1) for f in faces])
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def _sky_radiance(d: np.ndarray) -> np.ndarray:
    """Smooth gradient + a sun lobe + a few low-frequency lobes; linear radiance, shape (..., 3)."""
    y = d[..., 1]
    up = np.clip(y, 0.0, 1.0)[..., None]
    down = np.clip(-y, 0.0, 1.0)[..., None]
    horizon = np.array([0.75, 0.80, 0.90])
    zenith = np.array([0.15, 0.35, 0.85])
    ground = np.array([0.22, 0.20, 0.17])
    col = horizon * (1.0 - up) + zenith * up
    col = col * (1.0 - down) + ground * down
    sun = np.array([0.35, 0.55, -0.76])
    sun = sun / np.linalg.norm(sun)
    mu = np.clip((d * sun).sum(-1), 0.0, 1.0)[..., None]
    col = col + np.array([1.0, 0.9, 0.7]) * (mu ** 64.0) * 4.0 + np.array([0.3, 0.25, 0.2]) * mu ** 4.0
    wob = 0.06 * np.sin(7.0 * d[..., 0:1] + 3.0 * d[..., 2:3]) * np.cos(5.0 * d[..., 1:2] - 2.0 * d[..., 0:1])
    return np.clip(col * (1.0 + wob), 0.0, None)


def synthetic_sky_rgba32f(size: int, scale: float = 1.0) -> np.ndarray:
    """(6, size, size, 4) float32, alpha = 1 — stand-in for the RGBA32F atmosphere cube."""
    rgb = _sky_radiance(face_directions(size)) * scale
    out = np.ones((6, size, size, 4), dtype=F)
    out[..., :3] = rgb.astype(F)
    return out


def synthetic_sky_srgb8(size: int) -> np.ndarray:
    """(6, size, size, 4) uint8 sRGB-encoded, alpha = 255 — stand-in for the SRGB8_ALPHA8 sky box."""
    rgb = np.clip(_sky_radiance(face_directions(size)), 0.0, 1.0)
    enc = np.where(rgb <= 0.0031308, rgb * 12.92, 1.055 * np.power(rgb, 1.0 / 2.4) - 0.055)
    out = np.full((6, size, size, 4), 255, dtype=np.uint8)
    out[..., :3] = np.clip(np.rint(enc * 255.0), 0, 255).astype(np.uint8)
    return out


def tiny_test_cube(size: int = 2) -> np.ndarray:
    """(6, size, size, 4) float32 with every texel distinct: value = 10*face + row*size + col in R,
    G = -R, B = 0.5 — for sampler unit tests (face selection, bilinear weights, seams, corners)."""
    out = np.ones((6, size, size, 4), dtype=F)
    for f in range(6):
        v = (10.0 * f + np.arange(size * size, dtype=np.float64).reshape(size, size)).astype(F)
        out[f, :, :, 0] = v
        out[f, :, :, 1] = -v
        out[f, :, :, 2] = 0.5
    return out
