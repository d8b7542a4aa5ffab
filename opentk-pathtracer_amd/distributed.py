"""Multi-GPU path: one process per GPU, the image tiled in contiguous row blocks, ONE gather at present time.

The reference is single-GPU (one GL context); this module is the scale-out the MI355X build adds.  The integrator
has no inter-pixel dependency — a pixel's value depends only on (x, y, frame, W, H) and read-only inputs
(compute.glsl:104-129) — so rank g of G owns rows [g*H/G, (g+1)*H/G) and renders them with GLOBAL pixel coordinates
(pt_set_tile); accumulation stays resident on the owning GPU across frames and nothing is exchanged per frame.
Only presenting / reading back the image needs communication: one `gather` of the row blocks to the presenting
rank over RCCL (xGMI: every peer has its own link to the root, so the G-1 sends proceed in parallel).

torch is plumbing here: it owns the tile buffer that the HIP kernel accumulates into (pt_bind_result_buffer) so
that torch.distributed can hand the same memory to RCCL without a copy.
"""
from __future__ import annotations

import os


def row_block(height: int, rank: int, world: int) -> tuple[int, int]:
    """Rows [y0, y0+rows) owned by `rank`: contiguous, covering, differing by at most one row."""
    y0 = rank * height // world
    y1 = (rank + 1) * height // world
    return y0, y1 - y0


def max_rows(height: int, world: int) -> int:
    return max(row_block(height, r, world)[1] for r in range(world))


def interleaved_rows(height: int, rank: int, world: int, band_rows: int) -> list[int]:
    """Image rows owned by `rank` under block-cyclic ownership (pt_set_interleaved_tile), in storage order: bands
    rank, rank + world, ... of `band_rows` rows each.  Row cost varies ~2x between sky and floor rows, so contiguous
    blocks leave the GPU with the floor rows as the straggler; interleaving 16-row bands balances the ranks."""
    rows = []
    b = rank
    while b * band_rows < height:
        rows.extend(range(b * band_rows, min((b + 1) * band_rows, height)))
        b += world
    return rows


def max_interleaved_rows(height: int, world: int, band_rows: int) -> int:
    return max(len(interleaved_rows(height, r, world, band_rows)) for r in range(world))


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    backend 'nccl' is RCCL on ROCm; falls back to 'gloo' only when there is no GPU (CPU tests)."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def attach_tile(tracer, height: int, rank: int, world: int, device=None, band_rows: int = 0):
    """Give `tracer` (a PathTracer) its rows (a contiguous block, or block-cyclic bands when band_rows > 0) and a
    torch-owned accumulation buffer padded to the largest share, so that all ranks gather equal-sized tensors.
    Returns the (max_rows, W, 4) float32 tensor."""
    import torch

    if band_rows and world > 1:
        tracer.SetInterleavedTile(rank, world, band_rows)
        pad = max_interleaved_rows(height, world, band_rows)
    else:
        y0, rows = row_block(height, rank, world)
        tracer.SetTile(y0, rows)
        pad = max_rows(height, world)
    buf = torch.zeros((pad, tracer.Width, 4), dtype=torch.float32, device=device if device is not None else "cuda")
    # the zero-fill ran on torch's stream; the library renders on its own streams: finish it before the buffer is bound
    if buf.is_cuda:
        torch.cuda.current_stream(buf.device).synchronize()
    tracer.BindResultBuffer(buf.data_ptr(), buf.numel() * 4)
    return buf


def present(tile, height: int, rank: int, world: int, dst: int = 0, band_rows: int = 0):
    """Gather the ranks' rows on rank `dst` and return the assembled (height, W, 4) tensor there (None elsewhere).
    `tile` is this rank's (max_rows, W, 4) tensor (CPU tensor with gloo, CUDA tensor with nccl/RCCL)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return tile[:height]
    if rank == dst:
        parts = [torch.empty_like(tile) for _ in range(world)]
        dist.gather(tile, gather_list=parts, dst=dst)
        out = torch.empty((height,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        for r in range(world):
            if band_rows:
                rows = interleaved_rows(height, r, world, band_rows)
                idx = torch.as_tensor(rows, dtype=torch.long, device=tile.device)
                out.index_copy_(0, idx, parts[r][:len(rows)])
            else:
                y0, n = row_block(height, r, world)
                out[y0:y0 + n] = parts[r][:n]
        return out
    dist.gather(tile, gather_list=None, dst=dst)
    return None


class _DeviceBytes:
    """Zero-copy view of `nbytes` of device memory at `ptr` for torch.as_tensor (CUDA array interface, version 2)."""

    def __init__(self, ptr: int, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def postprocessed_tile(tracer, pad_rows: int | None = None):
    """This rank's tone-mapped RGBA8 rows (PostProcessing/fragment.glsl: ACES + gamma, run by the library on the GPU) as
    a torch uint8 CUDA tensor of shape (pad_rows or rows, W, 4) — what `present_rgba8` gathers."""
    import torch

    ptr, nbytes = tracer.PostProcessDevice()
    rows, w = tracer.rows, tracer.Width
    assert nbytes >= rows * w * 4
    torch.cuda.init()  # the array-interface path does not initialise torch's lazy CUDA state by itself
    view = torch.as_tensor(_DeviceBytes(ptr, (rows, w, 4)), device="cuda")
    tracer.Synchronize()  # the post-process kernel ran on the library's stream; torch reads on its own
    pad = rows if pad_rows is None else pad_rows
    out = torch.zeros((pad, w, 4), dtype=torch.uint8, device=view.device)
    out[:rows].copy_(view)
    return out


def present_rgba8(tracer, height: int, rank: int, world: int, dst: int = 0, band_rows: int = 0):
    """Multi-GPU present of the DISPLAYED image: every rank tone-maps its own rows on its GPU and the gather moves
    4 bytes per pixel instead of 16 (the reference presents an RGBA8 image, ScreenEffect.cs:29-37)."""
    pad = (max_interleaved_rows(height, world, band_rows) if band_rows and world > 1 else max_rows(height, world))
    return present(postprocessed_tile(tracer, pad), height, rank, world, dst=dst, band_rows=band_rows)
