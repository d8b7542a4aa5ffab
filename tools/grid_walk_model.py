"""CPU model of the sphere-grid walk of large scenes (development aid; no GPU needed): how many vector instructions a 64-lane
wavefront spends in the walk under different control structures, for the 256-sphere scene of BASELINE configs[2].

The grid is the library's own (pt_debug_build_sphere_grid, csrc/pt_sphere_grid.hpp); rays are secondary rays of that scene as
the integrator produces them (origins on sphere surfaces and on the room's walls, cosine-weighted directions; the exact ray
distribution matters little for the ratios).  Every ray is walked cell by cell exactly as ray_trace_t<GRID> does (3D-DDA, tests
in list order, stop when the accepted hit lies before the cell's exit), recording per cell visit how many spheres it tested.
Then, for random groups of 64 rays:

  lock-step rounds (today)   sum over rounds r of (STEP + TEST * max over lanes of tests in round r)
  flattened state machine    max over lanes of (tests + cells) iterations of (STEP + TEST)
  refill from a job pool     lanes that finish pull the next ray of a pool of 64 + R rays (R = rays waiting in the ring);
                             cost per ray = phase length / rays, phase length from a greedy list schedule of the rounds
  continuous refill          the same loop never drains: every lane always has a ray (bound: both code paths issue every iteration)
  ideal                      mean over lanes of (TEST * tests + STEP * cells): every lane always busy

STEP / TEST are the measured instruction counts of the kernel's cell step and sphere test (~20 / ~30 VALU).
python tools/grid_walk_model.py [rays] [seed]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

STEP, TEST, SETUP = 20.0, 30.0, 60.0
pkg = g.load_package()
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 400
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)

sc = pkg.scene.stress_scene(256)
objs = np.frombuffer(sc.ubo_bytes(), np.float32).copy()
lib = C.CDLL(pkg.native.LIB_PATH)
header, box = (C.c_int * 5)(), (C.c_float * 10)()
packed = (C.c_ubyte * 4096)()
lib.pt_debug_build_sphere_grid.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_ubyte), C.c_int]
nbytes = lib.pt_debug_build_sphere_grid(objs.ctypes.data_as(C.POINTER(C.c_float)), 256, header, box, packed, 4096)
assert header[4] == 1, "no grid for this scene"
dims = np.array(header[:3]); lo = np.array(box[0:3], np.float64); hi = np.array(box[3:6], np.float64)
cells = int(dims.prod())
raw = bytes(packed[:nbytes])
starts = np.frombuffer(raw[:2 * (cells + 1)], np.uint16).astype(np.int64)
refs = np.frombuffer(raw[2 * (cells + 1):2 * (cells + 1) + int(starts[-1])], np.uint8).astype(np.int64)
cell_size = (hi - lo) / dims
centres = objs[:256 * 20].reshape(256, 20)[:, :3].astype(np.float64)
radii = objs[:256 * 20].reshape(256, 20)[:, 3].astype(np.float64)
print(f"grid {dims.tolist()} cells, {len(refs)} references ({len(refs) / cells:.2f} per cell), box {lo.round(1).tolist()} .. {hi.round(1).tolist()}")


def cosine_dir(n):
    z = 2.0 * rng.rand(len(n)) - 1.0
    a = 2.0 * np.pi * rng.rand(len(n))
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    v = n + np.stack([r * np.cos(a), r * np.sin(a), z], 1)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


# secondary rays: half leave a sphere's surface, half a wall of the room (40 x 25 x 25 around (0, 0, -12.5): MainWindow.cs:208-213)
half = n_rays // 2
si = rng.randint(0, 256, half)
nrm = rng.randn(half, 3); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
o1 = centres[si] + nrm * (radii[si, None] + 1e-3)
d1 = cosine_dir(nrm)
room_lo, room_hi = np.array([-20.0, -12.5, -25.0]), np.array([20.0, 12.5, 0.0])
axis = rng.randint(0, 3, n_rays - half); side = rng.randint(0, 2, n_rays - half)
o2 = room_lo + rng.rand(n_rays - half, 3) * (room_hi - room_lo)
n2 = np.zeros((n_rays - half, 3))
for k in range(n_rays - half):
    o2[k, axis[k]] = room_hi[axis[k]] if side[k] else room_lo[axis[k]]
    n2[k, axis[k]] = -1.0 if side[k] else 1.0
d2 = cosine_dir(n2)
O, D = np.concatenate([o1, o2]), np.concatenate([d1, d2])
perm = rng.permutation(n_rays)
O, D = O[perm], D[perm]


def walk(o, d):
    """-> list of tests per visited cell (in order) for one ray, exactly the kernel's stop rule."""
    inv = np.where(np.abs(d) > 1e-18, 1.0 / np.where(d == 0, 1.0, d), np.where(d < 0, -1e18, 1e18))
    t0, t1 = (lo - o) * inv, (hi - o) * inv
    tn = max(0.0, np.minimum(t0, t1).max()); tf = np.maximum(t0, t1).min()
    if tn > tf:
        return []
    p = o + d * tn
    c = np.clip(((p - lo) / cell_size).astype(int), 0, dims - 1)
    pos = d >= 0
    m = (lo + (c + pos) * cell_size - o) * inv
    dt = np.abs(cell_size * inv)
    left = np.where(pos, dims - 1 - c, c)
    T = np.inf
    out = []
    while True:
        ci = (c[2] * dims[1] + c[1]) * dims[0] + c[0]
        lst = refs[starts[ci]:starts[ci + 1]]
        out.append(len(lst))
        for j in lst:
            oc = o - centres[j]
            b = d @ oc; cc = oc @ oc - radii[j] ** 2; disc = b * b - cc
            if disc >= 0:
                sq = np.sqrt(disc); ta, tb = -b - sq, -b + sq
                if tb > 0:
                    t = tb if ta < 0 else ta
                    if (ta < 0) or ta < T:
                        T = min(T, t) if ta >= 0 else t
        ax = int(np.argmin(m))
        if T <= m[ax] or left[ax] <= 0:
            return out
        m[ax] += dt[ax]; left[ax] -= 1; c[ax] += 1 if pos[ax] else -1


per_ray = [walk(O[i], D[i]) for i in range(n_rays)]
cells_v = np.array([len(x) for x in per_ray]); tests_v = np.array([sum(x) for x in per_ray])
print(f"{n_rays} rays: cells per ray mean {cells_v.mean():.2f} (max {cells_v.max()}), sphere tests per ray mean {tests_v.mean():.2f} (max {tests_v.max()}); "
      f"{(cells_v == 0).mean() * 100:.1f} % miss the grid's box")

lock, flat, ideal, slow_c, slow_t, sum_max = [], [], [], [], [], []
pool = {32: [], 64: []}
for w in range(n_rays // 64):
    grp = per_ray[w * 64:(w + 1) * 64]
    rounds = max(len(x) for x in grp)
    mx = [max((x[r] if r < len(x) else 0) for x in grp) for r in range(rounds)]
    lock.append(sum(STEP + TEST * m_ for m_ in mx))
    sum_max.append(sum(mx))
    flat.append(max(sum(x) + len(x) for x in grp) * (STEP + TEST))
    ideal.append(np.mean([TEST * sum(x) + STEP * len(x) for x in grp]))
    slow_c.append(rounds); slow_t.append(max(sum(x) for x in grp))
for R in pool:
    for w in range(n_rays // (64 + R)):
        jobs = per_ray[w * (64 + R):(w + 1) * (64 + R)]
        # greedy: 64 lanes, each job costs its own serial time (flattened per-lane cost), lanes pull the next job when free;
        # the wavefront executes both code paths while any lane needs them: phase length = max lane finish time in iterations
        cost = [sum(x) + len(x) + (SETUP / (STEP + TEST)) for x in jobs]
        lanes = np.zeros(64)
        for cst in cost:
            k = int(np.argmin(lanes)); lanes[k] += cst
        pool[R].append(lanes.max() * (STEP + TEST) / len(jobs) * 64)  # per 64 rays
lock, flat, ideal = np.array(lock), np.array(flat), np.array(ideal)
print(f"per wavefront of 64 rays: slowest lane {np.mean(slow_c):.1f} cells / {np.mean(slow_t):.1f} tests; sum over rounds of the longest list {np.mean(sum_max):.1f}")
print(f"VALU instructions per 64 rays (model: cell step {STEP:.0f}, sphere test {TEST:.0f}):")
print(f"  ideal (all lanes busy)              {ideal.mean():8.0f}")
print(f"  lock-step rounds (today)            {lock.mean():8.0f}   lane utilisation {ideal.mean() / lock.mean() * 100:.0f} %")
print(f"  flattened per-lane state machine    {flat.mean():8.0f}   lane utilisation {ideal.mean() / flat.mean() * 100:.0f} %")
cont = (tests_v.sum() + cells_v.sum() + n_rays * SETUP / (STEP + TEST)) * (STEP + TEST) / n_rays * 64 / 64
print(f"  continuous refill (no phase end)    {cont:8.0f}   lane utilisation {ideal.mean() / cont * 100:.0f} %   (flattened loop that never drains: a finished lane takes "
      f"the next ray at once, hits go to a queue; bound = both code paths issued every iteration)")
for R, v in pool.items():
    print(f"  refill from a pool of 64 + {R:2d} rays   {np.mean(v):8.0f}   lane utilisation {ideal.mean() / np.mean(v) * 100:.0f} %   (flattened loop, {SETUP:.0f}-instruction set-up per ray)")
