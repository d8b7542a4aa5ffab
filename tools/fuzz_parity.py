"""Randomised HIP-vs-oracle parity fuzzing (development helper): random scenes (counts, sizes from tiny to huge, nested and
overlapping spheres, random materials incl. glass), random cameras (inside objects too), lens, depth, spp, image size,
frame count, batch size and (round 2) group handles over 1-3 parts; every image must equal the oracle bit for bit.
FUZZ_FOCUS=pipelining: tiny images, many frames per launch.  FUZZ_FOCUS=grid: 64-256 spheres at 1 spp (the sphere grid of large
scenes): planar / clustered / tiny / far-from-origin layouts, duplicate and degenerate spheres, cameras inside spheres and far away.
FUZZ_BIAS=group_spp (with FUZZ_FOCUS=pipelining): always several samples per pixel and a group handle.
MI355PT_LIB=opentk-pathtracer_amd/libmi355pt_audit.so runs it on the audit build: the kernels' own hand-over audit is read after every case.
A mismatch keeps its evidence under gpurun_out/fuzz_failures/ (both reads, both oracle runs, all inputs).
python tools/fuzz_parity.py [cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
oracle = g.load_oracle().Oracle()
S = pkg.scene
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
F = np.float32

def rand_material():
    kind = rng.randint(5)
    if kind == 0:
        return S.Material(albedo=rng.rand(3), emissiv=rng.rand(3) * (rng.rand() < 0.3) * 3)
    if kind == 1:
        return S.Material(albedo=rng.rand(3), specular_chance=rng.rand(), specular_roughness=rng.rand() * (rng.rand() < 0.7))
    if kind == 2:
        return S.Material(albedo=S.vec3(1.0), absorbance=rng.rand(3) * 2, specular_chance=0.02 + 0.1 * rng.rand(), ior=1.0 + rng.rand(),
                          refraction_chance=0.9 * rng.rand(), refraction_roughness=rng.rand() * (rng.rand() < 0.5))
    return S.Material(albedo=rng.rand(3), specular_chance=rng.rand() * 0.5, specular_roughness=rng.rand(), ior=1 + rng.rand(),
                      refraction_chance=rng.rand() * 0.5, refraction_roughness=rng.rand(), absorbance=rng.rand(3))

ONLY = int(os.environ["FUZZ_ONLY"]) if os.environ.get("FUZZ_ONLY") else None  # replay one case of a run (FUZZ_REPEAT times)
REPEAT = int(os.environ.get("FUZZ_REPEAT", "1"))
JITTER = np.random.RandomState(int(os.environ["FUZZ_JITTER"])) if os.environ.get("FUZZ_JITTER") else None  # random pauses between pt_render calls
GRID = os.environ.get("FUZZ_FOCUS") == "grid"  # many spheres, one sample: the sphere-grid traversal of large scenes

def rand_scene():
    sc = S.Scene()
    ns = int(rng.choice([0, 1, 3, 17, 48, 64, 65, 100, 200, 256]))
    nc = int(rng.choice([0, 1, 7, 20, 64]))
    scale = float(rng.choice([0.05, 0.6, 2.0, 8.0, 40.0]))
    lo, hi, off = np.array([-18, -11, -20], F), np.array([18, 11, 0], F), np.zeros(3, F)
    clusters = None
    if GRID:
        ns = int(rng.choice([64, 65, 80, 100, 128, 160, 200, 255, 256]))
        scale = float(rng.choice([0.01, 0.05, 0.3, 0.6, 0.6, 1.0, 2.0, 4.0]))
        kind = rng.randint(6)
        if kind == 1:    # everything in one plane / one line
            hi = lo + (hi - lo) * np.array([[1, 1, 0], [1, 0, 0], [0, 1, 1]][rng.randint(3)], F)
        elif kind == 2:  # far from the origin: coarse fp32 spacing
            off = rng.uniform(-1, 1, 3).astype(F) * F(rng.choice([100.0, 1000.0, 20000.0]))
        elif kind == 3:  # a few tight clusters
            clusters = rng.uniform(lo, hi, (int(rng.choice([2, 5, 12])), 3)).astype(F)
        elif kind == 4:  # a tiny scene
            lo, hi = lo * F(0.01), hi * F(0.01)
            scale *= 0.01
    for i in range(ns):
        if GRID and i > 0 and rng.rand() < 0.08:  # an exact copy of an earlier sphere's geometry: equal t1, the lower index must win
            src = sc.spheres[rng.randint(i)]
            sc.spheres.append(S.Sphere(src.position.copy(), src.radius, i, rand_material()))
            continue
        if clusters is not None:
            pos = (clusters[rng.randint(len(clusters))] + rng.randn(3).astype(F) * F(1.5 * scale) + off).astype(F)
        else:
            pos = (rng.uniform(lo, hi).astype(F) + off).astype(F)
        r = F(scale * rng.uniform(0.2, 1.5))
        if GRID and rng.rand() < 0.03:
            r = F(rng.choice([0.0, -0.7, 1e-4, 30.0]))  # degenerate, negative (only r*r is used), tiny, one giant among the small
        sc.spheres.append(S.Sphere(pos, r, i, rand_material()))
    sc.fuzz_offset = off
    if rng.rand() < 0.5:
        sc.cuboids = S.default_cuboids()[:min(nc, 7)]
        for c in sc.cuboids:
            c.position = (np.asarray(c.position, F) + off).astype(F)
    for i in range(len(sc.cuboids), nc):
        c = (rng.uniform([-18, -11, -20], [18, 11, 0]).astype(F) + off).astype(F)
        sc.cuboids.append(S.Cuboid(c, rng.uniform(0.2, 6.0, 3).astype(F), i, rand_material()))
    return sc

def audit_violations(pt):
    """> 0 only with the -DPT_AUDIT build of the library (MI355PT_LIB=.../libmi355pt_audit.so): its kernels mirror every pixel
    read-modify-write with a device-scope atomic side word and log resolves that ran out of order or on a stale colour."""
    import ctypes as C
    fn = getattr(pt._lib, "pt_debug_audit_read", None)
    if fn is None:
        return 0
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.c_int]
    fn.restype = C.c_int
    n = fn(pt._h, None, 0)
    return n if n > 0 else 0


bad = 0
grids_used = 0
t0 = time.time()
for case in range(cases):
    sc = rand_scene()
    W, H = int(rng.choice([8, 33, 64, 120, 200])), int(rng.choice([8, 17, 40, 72, 113]))
    depth, spp = int(rng.choice([0, 1, 2, 5, 8, 20])), int(rng.choice([1, 1, 1, 2, 3, 4, 7]))
    frames, batch = int(rng.choice([1, 2, 5, 9, 33])), int(rng.choice([1, 2, 32, 64]))
    if os.environ.get("FUZZ_FOCUS") == "pipelining":  # tiny images, many frames in one pipelined launch, several samples
        W, H = int(rng.choice([8, 8, 16, 33])), int(rng.choice([8, 8, 17, 40]))
        spp = int(rng.choice([1, 2, 3, 4, 7]))
        frames, batch = int(rng.choice([33, 64, 70])), 64
    parts = int(rng.choice([0, 0, 0, 1, 2, 3]))  # > 0: a group handle over `parts` copies of device 0 (pt_create_multi)
    if os.environ.get("FUZZ_BIAS") == "group_spp":  # the class of the one unexplained mismatch: several samples, a group handle
        spp = int(rng.choice([2, 3, 4]))
        parts = int(rng.choice([2, 3]))
    if parts and H < parts:
        parts = 0
    cam_pos = rng.uniform([-19, -12, -22], [19, 12, 2])
    if GRID:
        spp = 1
        depth = int(rng.choice([2, 5, 8, 20]))
        W, H = int(rng.choice([33, 64, 120])), int(rng.choice([17, 40, 72]))
        where = rng.randint(4)
        if where == 1 and sc.num_spheres:  # inside (or at the surface of) a sphere
            sp = sc.spheres[rng.randint(sc.num_spheres)]
            cam_pos = np.asarray(sp.position, np.float64) - sc.fuzz_offset + rng.uniform(-1, 1, 3) * abs(float(sp.radius))
        elif where == 2:                   # far outside: beyond the grid's reach (in-order loop)
            cam_pos = cam_pos * float(rng.choice([3.0, 10.0, 100.0]))
    cam_pos = cam_pos + getattr(sc, "fuzz_offset", np.zeros(3))
    cam = pkg.camera.Camera(position=tuple(float(v) for v in cam_pos),
                            look_x=float(rng.uniform(-180, 180)), look_y=float(rng.uniform(-85, 85)))
    focal, aperture = float(rng.choice([0.5, 5.0, 20.0, 200.0])), float(rng.choice([0.0, 0.14, 2.0, 15.0]))
    env = pkg.envmap.synthetic_sky_rgba32f(16) if rng.rand() < 0.7 else pkg.envmap.synthetic_sky_srgb8(16)
    basic = pkg.camera.basic_data_ubo(cam, W, H)
    band = int(rng.choice([0, 8, 16])) if parts > 1 and rng.rand() < 0.5 else None
    if ONLY is not None and case != ONLY:  # (the random stream is consumed above: the selected case is exactly the one of a full run)
        continue
    pt = pkg.PathTracer(env, W, H, depth, spp, focal, aperture, **({"devices": [0] * parts} if parts else {}))
    if band is not None:
        pt.SetPartition(band)
    pt.SetFrameBatch(batch)
    pt.UploadScene(sc); pt.UploadBasicData(basic)
    if GRID:  # how many cases really walk a grid (the others fall back to the in-order loop: too few / too big spheres)
        import ctypes as C
        info = (C.c_int * 5)()
        pt._lib.pt_debug_sphere_grid.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        pt._lib.pt_debug_sphere_grid(pt._h, info)
        grids_used += int(info[4])
    desc = (f"ns={sc.num_spheres} nc={sc.num_cuboids} {W}x{H} depth={depth} spp={spp} frames={frames} batch={batch} parts={parts} band={band} "
            f"focal={focal} aperture={aperture}")
    if os.environ.get("FUZZ_VERBOSE"):
        print(f"case {case}: {desc}", flush=True)
    want = None
    for rep in range(REPEAT):
        if rep and os.environ.get("FUZZ_FRESH"):  # a new handle per repetition (fresh allocations, first launches of a handle)
            pt.Dispose()
            pt = pkg.PathTracer(env, W, H, depth, spp, focal, aperture, **({"devices": [0] * parts} if parts else {}))
            if band is not None:
                pt.SetPartition(band)
            pt.SetFrameBatch(batch)
            pt.UploadScene(sc); pt.UploadBasicData(basic)
        elif rep:
            pt.ResetRenderer()
        try:
            for _ in range(frames):
                pt.Render()
                if JITTER is not None:  # vary how the library groups frames into launches (it launches when the GPU is idle)
                    t_end = time.perf_counter() + float(JITTER.choice([0, 0, 0, 5e-6, 2e-5, 1e-4, 4e-4]))
                    while time.perf_counter() < t_end: pass
            got = pt.Result
        except Exception as e:  # an error code from the library is a failure of the case, not of the fuzzer
            bad += 1
            print(f"case {case}: {e}: {desc}", flush=True)
            break
        if want is None:
            want = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_spheres=sc.num_spheres, num_cuboids=sc.num_cuboids, ray_depth=depth,
                                 spp=spp, focal_length=focal, aperture=aperture, num_frames=frames)
        same = (got.view(np.uint32) == want.view(np.uint32)).all(-1)
        nviol = audit_violations(pt)
        if nviol:
            print(f"case {case}: {nviol} hand-over AUDIT violation(s) reported by the kernels: {desc}", flush=True)
            bad += 1
        if not same.all():
            bad += 1
            ys, xs = np.nonzero(~same)
            print(f"case {case}" + (f" (repetition {rep})" if REPEAT > 1 else "") + f": {int((~same).sum())}/{same.size} pixels differ: {desc}; first at x={xs[0]} y={ys[0]}: "
                  f"got {got[ys[0], xs[0]]} want {want[ys[0], xs[0]]}", flush=True)
            # keep the evidence (round 2 lost it): what was read, a second read of the same image (a transient read-back error looks
            # different from a wrong accumulation), the oracle's image, the oracle's image computed AGAIN single-threaded, all inputs
            again = pt.Result
            want1 = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_spheres=sc.num_spheres, num_cuboids=sc.num_cuboids, ray_depth=depth,
                                  spp=spp, focal_length=focal, aperture=aperture, num_frames=frames, threads=1)
            os.makedirs("gpurun_out/fuzz_failures", exist_ok=True)
            path = f"gpurun_out/fuzz_failures/seed{sys.argv[2] if len(sys.argv) > 2 else 1}_case{case}_rep{rep}.npz"
            np.savez_compressed(path, got=got, got_second_read=again, want=want, want_single_thread=want1, basic=np.frombuffer(basic, np.uint8),
                                objects=np.frombuffer(sc.ubo_bytes(), np.uint8), env=env, desc=np.array(desc))
            print(f"    second read identical to the first: {bool((again.view(np.uint32) == got.view(np.uint32)).all())}; oracle single-threaded identical "
                  f"to multi-threaded: {bool((want1.view(np.uint32) == want.view(np.uint32)).all())}; second read equals the oracle: "
                  f"{bool((again.view(np.uint32) == want.view(np.uint32)).all())}; saved {path}", flush=True)
    pt.Dispose()
print(f"{cases} cases, {bad} with differences, {time.time() - t0:.1f} s" + (f", {grids_used} of them walked a sphere grid" if GRID else ""))
sys.exit(1 if bad else 0)
