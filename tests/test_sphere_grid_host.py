"""CPU tests of the sphere-grid builder (csrc/pt_sphere_grid.hpp, through the library's host-only test export): the structure
the GPU walk relies on.  The bit-exact end-to-end checks are the `-m gpu` tests in test_gpu_sphere_grid.py."""
import ctypes as C

import numpy as np
import pytest


def build(native_lib, scene):
    if not hasattr(native_lib, "pt_debug_build_sphere_grid"):
        pytest.skip("library built without the host-only grid export")
    fn = native_lib.pt_debug_build_sphere_grid
    fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_void_p, C.c_int]
    objs = np.frombuffer(scene.ubo_bytes(), dtype=np.float32).copy()
    header, box = (C.c_int * 5)(), (C.c_float * 10)()
    packed = np.zeros(4096, np.uint8)
    n = fn(objs.ctypes.data, scene.num_spheres, header, box, packed.ctypes.data, packed.nbytes)
    dims = np.array(header[0:3])
    cells = int(dims.prod())
    starts = packed[:2 * (cells + 1)].view(np.uint16).astype(int)
    refs = packed[2 * (cells + 1):2 * (cells + 1) + header[3]].astype(int)
    return dict(valid=bool(header[4]), dims=dims, nrefs=header[3], lo=np.array(box[0:3]), hi=np.array(box[3:6]), center=np.array(box[6:9]),
                reach2=box[9], starts=starts, refs=refs, nbytes=n)


def walk(g, o, d):
    """the cells a ray visits, in order (float64 3D-DDA) -> list of cell indices"""
    inv = 1.0 / np.where(np.abs(d) > 1e-18, d, np.where(d < 0, -1e-18, 1e-18))
    t0, t1 = (g["lo"] - o) * inv, (g["hi"] - o) * inv
    tn, tf = max(0.0, np.minimum(t0, t1).max()), np.maximum(t0, t1).min()
    if not tn <= tf:
        return []
    cell = (g["hi"] - g["lo"]) / g["dims"]
    c = np.clip(((o + d * tn - g["lo"]) / cell).astype(int), 0, g["dims"] - 1)
    pos = d >= 0
    tmax = (g["lo"] + (c + pos) * cell - o) * inv
    td = np.abs(cell * inv)
    out = []
    while True:
        out.append(((c[2] * g["dims"][1] + c[1]) * g["dims"][0] + c[0], tmax.min()))
        a = int(np.argmin(tmax))
        c[a] += 1 if pos[a] else -1
        if c[a] < 0 or c[a] >= g["dims"][a]:
            return out
        tmax[a] += td[a]


@pytest.mark.parametrize("kind", ["stress256", "random200", "tiny", "offset"])
def test_every_hit_is_listed_where_the_walk_finds_it(pkg, native_lib, kind):
    """For random rays from inside the grid's reach: the nearest sphere hit (float64 geometry) is listed in a cell the ray visits
    no later than the cell that contains the hit point — the property the walk's stop criterion needs; lists are ascending,
    duplicate-free and inside the LDS budget."""
    S = pkg.scene
    rng = np.random.RandomState(5)
    if kind == "stress256":
        sc = S.stress_scene()
    else:
        sc = S.Scene()
        n, rad, off = {"random200": (200, (0.2, 1.5), 0.0), "tiny": (256, (0.005, 0.05), 0.0), "offset": (150, (0.3, 1.0), 5000.0)}[kind]
        for i in range(n):
            sc.spheres.append(S.Sphere((rng.uniform([-18, -11, -20], [18, 11, 0]) + off).astype(np.float32), np.float32(rng.uniform(*rad)), i,
                                       S.Material()))
        sc.cuboids = S.default_cuboids()
    g = build(native_lib, sc)
    assert g["valid"] and g["dims"].prod() <= 256 and g["nrefs"] <= 1024 and g["nbytes"] % 4 == 0
    cells = int(g["dims"].prod())
    assert g["starts"][0] == 0 and g["starts"][cells] == g["nrefs"] and (np.diff(g["starts"]) >= 0).all()
    for c in range(cells):
        lst = g["refs"][g["starts"][c]:g["starts"][c + 1]]
        assert (np.diff(lst) > 0).all(), "ascending, no duplicates"
    C_ = np.array([s.position for s in sc.spheres], np.float64)
    R_ = np.array([abs(float(s.radius)) for s in sc.spheres], np.float64)
    reach = np.sqrt(g["reach2"])
    hits = 0
    for _ in range(1500):
        o = g["center"] + rng.randn(3) / np.sqrt(3) * reach * rng.rand()
        if ((o - g["center"]) ** 2).sum() > g["reach2"]:
            continue
        if rng.rand() < 0.7:  # aimed at (or just past) a random sphere, so that small spheres are hit and grazed too
            k = rng.randint(len(C_))
            d = C_[k] + rng.randn(3) * R_[k] * 0.6 - o
        else:
            d = rng.randn(3)
        d /= np.linalg.norm(d)
        oc = o - C_
        b = oc @ d
        c = (oc * oc).sum(1) - R_ ** 2
        disc = b * b - c
        ok = disc >= 0
        t1 = np.where(ok, -b - np.sqrt(np.where(ok, disc, 0)), np.inf)
        t1[t1 < 0] = np.inf  # (origins inside a sphere: covered below)
        j = int(np.argmin(t1))
        inside = np.where(c < 0)[0]
        visited = walk(g, o, d)
        if inside.size:  # every sphere that contains the origin is listed in the first cell of the walk
            first = g["refs"][g["starts"][visited[0][0]]:g["starts"][visited[0][0] + 1]]
            assert set(inside) <= set(first)
        if not np.isfinite(t1[j]):
            continue
        hits += 1
        seen = set()
        for cell, texit in visited:
            seen |= set(g["refs"][g["starts"][cell]:g["starts"][cell + 1]])
            if texit >= t1[j]:
                break
        assert j in seen, f"sphere {j} (t1 = {t1[j]}) not listed up to the cell of its hit point"
    assert hits > 100


def test_grid_is_refused_for_small_huge_and_non_finite_scenes(pkg, native_lib):
    S = pkg.scene
    assert not build(native_lib, S.default_scene())["valid"]           # 48 spheres: the in-order loop is as fast
    sc = S.stress_scene()
    sc.spheres[7].position = np.array([np.nan, 0, 0], np.float32)
    assert not build(native_lib, sc)["valid"]
    sc = S.stress_scene()
    for s in sc.spheres:
        s.radius = np.float32(30.0)                                     # every sphere in every cell: lists over budget
    assert not build(native_lib, sc)["valid"]
    sc = S.stress_scene()
    for s in sc.spheres:
        s.position = np.array([1.0, 2.0, 3.0], np.float32)
        s.radius = np.float32(0.0)                                      # a single point: no extent
    assert not build(native_lib, sc)["valid"]


def test_set_rule_equals_in_order_loop_for_any_visiting_order():
    """The argument of ray_trace_t<GRID> in executable form.  Given per-sphere (valid, t1, t2) the reference visits spheres in
    index order and accepts `valid and t1 < T` with T := t1 < 0 ? t2 : t1 (compute.glsl:226-247).  The grid walk sees the spheres
    that contain the origin first, in ascending order (first cell), everything else in ANY order and possibly several times,
    and applies: index > L and (t1 < T, or t1 == T and lower index than an outside winner).  Both must agree — including equal
    t1 (duplicate spheres), several containing spheres, and a containing sphere's exit t2 tying with an outside entry."""
    rng = np.random.RandomState(9)
    FLT_MAX = np.float32(3.4028234663852886e38)
    for trial in range(4000):
        n = rng.randint(1, 24)
        vals = np.float32(rng.choice([0.5, 1.0, 1.5, 2.0, 2.5, 3.0], n))  # few distinct values: ties are common
        inside = rng.rand(n) < 0.25
        valid = rng.rand(n) < 0.8
        t1 = np.where(inside, -vals, vals).astype(np.float32)
        t2 = (np.abs(t1) + np.float32(rng.choice([0.0, 0.5, 1.0, 2.0], n))).astype(np.float32)  # t2 >= |t1| > 0; tangent hits have t1 == t2
        # the reference
        T, winner = FLT_MAX, -1
        for i in range(n):
            if valid[i] and t1[i] < T:
                T, winner = (t2[i] if t1[i] < 0 else t1[i]), i
        # the walk: first cell = every containing sphere plus a random subset of the others, ascending; then the rest, shuffled, with repeats
        first = sorted(set(np.nonzero(inside)[0]) | set(rng.choice(n, rng.randint(0, n + 1), replace=False)))
        later = list(rng.permutation(n)) + list(rng.choice(n, rng.randint(0, n + 1)))
        gT, gw, L = FLT_MAX, -1, -1
        for phase, seq in ((True, first), (False, later)):
            for j in seq:
                if not valid[j]:
                    continue
                if t1[j] < 0:
                    if phase:
                        L, gT, gw = j, t2[j], j
                    continue  # (later cells never list a containing sphere that the first cell does not: nothing to do)
                if j > L and (t1[j] < gT or (t1[j] == gT and gw != L and j < gw)):
                    gT, gw = t1[j], j
        assert (gw, gT) == (winner, T), (trial, winner, T, gw, gT)


def test_grid_walk_cost_model_runs_and_orders_the_control_structures():
    """tools/grid_walk_model.py (the numbers behind DESIGN.md section 3.5 / 9: why the lock-step walk was not replaced by a flattened
    state machine or a per-phase job pool) stays runnable on the library's own grid builder, and its ordering holds: a walk loop that
    never drains < pooled refill < lock-step rounds < flattened state machine, all above the ideal."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "grid_walk_model.py"), "6400", "3"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1500:]
    num = lambda label: float(re.search(re.escape(label) + r"\s+(\d+)", p.stdout).group(1))
    ideal, lock, flat, cont = num("ideal (all lanes busy)"), num("lock-step rounds (today)"), num("flattened per-lane state machine"), num("continuous refill (no phase end)")
    pool64 = num("refill from a pool of 64 + 64 rays")
    assert ideal < cont < pool64 < lock < flat, p.stdout
