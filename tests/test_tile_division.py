"""The tile prologue's division by a host-side magic number (csrc/pt_kernel_common.hpp: div_magic / fast_divmod): the estimate
mul_hi(n, floor(2^32 / d)) is the quotient or one less for every n < 2^31, so ONE correction step makes it exact.  The kernels use it for
ticket -> (frame of the batch, tile) and tile -> (row, column); a wrong quotient would render a tile at the wrong place (the GPU parity
tests on ragged and one-tile images would see it) — this is the arithmetic on its own."""
import numpy as np


def div_magic(d: int) -> int:
    return 0xFFFFFFFF if d <= 1 else (1 << 32) // d


def fast_divmod(n, d, magic):
    q = (n.astype(np.uint64) * np.uint64(magic)) >> np.uint64(32)
    r = n.astype(np.int64) - q.astype(np.int64) * d
    fix = r >= d
    return q.astype(np.int64) + fix, r - fix * d


def test_one_correction_step_is_enough():
    rng = np.random.RandomState(11)
    ds = [1, 2, 3, 5, 7, 8, 30, 135, 240, 405, 4050, 8100, 32400, 129600, 4095 * 4095, (1 << 24) - 1, (1 << 31) - 1]
    ds += [int(x) for x in rng.randint(1, 1 << 22, 200)]
    for d in ds:
        n = np.concatenate([rng.randint(0, (1 << 31) - 1, 4000).astype(np.int64),
                            np.array([0, 1, d - 1, d, d + 1, 2 * d - 1, (1 << 31) - 1, ((1 << 31) - 1) // d * d], dtype=np.int64)])
        n = np.clip(n, 0, (1 << 31) - 1)
        q, r = fast_divmod(n, d, div_magic(d))
        assert np.array_equal(q, n // d) and np.array_equal(r, n % d), d


def test_every_ticket_of_the_benchmark_launches():
    # 1080p and 4K tilings, 64- and 256-frame batches: every (frame, tile) ticket decodes to itself
    for tiles_x, tiles_y, frames in ((240, 135, 64), (480, 270, 64), (240, 17, 256), (1, 1, 256), (5, 5, 200)):
        per_frame = tiles_x * tiles_y
        t = np.arange(per_frame * frames, dtype=np.int64)
        fj, tile = fast_divmod(t, per_frame, div_magic(per_frame))
        ty, tx = fast_divmod(tile, tiles_x, div_magic(tiles_x))
        assert np.array_equal((fj * tiles_y + ty) * tiles_x + tx, t)
        assert tx.max() == tiles_x - 1 and ty.max() == tiles_y - 1 and fj.max() == frames - 1
