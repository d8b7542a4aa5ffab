"""Randomised HIP-vs-oracle parity fuzzing (development helper): random scenes (counts, sizes from tiny to huge, nested and
overlapping spheres, random materials incl. glass), random cameras (inside objects too), lens, depth, spp, image size,
frame count, batch size and (round 2) group handles over 1-3 parts; every image must equal the oracle bit for bit.   python tools/fuzz_parity.py [cases] [seed]"""
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

def rand_scene():
    sc = S.Scene()
    ns = int(rng.choice([0, 1, 3, 17, 48, 64, 65, 100, 200, 256]))
    nc = int(rng.choice([0, 1, 7, 20, 64]))
    scale = float(rng.choice([0.05, 0.6, 2.0, 8.0, 40.0]))
    for i in range(ns):
        pos = rng.uniform([-18, -11, -20], [18, 11, 0]).astype(F)
        sc.spheres.append(S.Sphere(pos, F(scale * rng.uniform(0.2, 1.5)), i, rand_material()))
    if rng.rand() < 0.5:
        sc.cuboids = S.default_cuboids()[:min(nc, 7)]
    for i in range(len(sc.cuboids), nc):
        c = rng.uniform([-18, -11, -20], [18, 11, 0]).astype(F)
        sc.cuboids.append(S.Cuboid(c, rng.uniform(0.2, 6.0, 3).astype(F), i, rand_material()))
    return sc

bad = 0
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
    if parts and H < parts:
        parts = 0
    cam = pkg.camera.Camera(position=tuple(float(v) for v in rng.uniform([-19, -12, -22], [19, 12, 2])),
                            look_x=float(rng.uniform(-180, 180)), look_y=float(rng.uniform(-85, 85)))
    focal, aperture = float(rng.choice([0.5, 5.0, 20.0, 200.0])), float(rng.choice([0.0, 0.14, 2.0, 15.0]))
    env = pkg.envmap.synthetic_sky_rgba32f(16) if rng.rand() < 0.7 else pkg.envmap.synthetic_sky_srgb8(16)
    basic = pkg.camera.basic_data_ubo(cam, W, H)
    pt = pkg.PathTracer(env, W, H, depth, spp, focal, aperture, **({"devices": [0] * parts} if parts else {}))
    if parts > 1 and rng.rand() < 0.5:
        pt.SetPartition(int(rng.choice([0, 8, 16])))
    pt.SetFrameBatch(batch)
    pt.UploadScene(sc); pt.UploadBasicData(basic)
    desc = (f"ns={sc.num_spheres} nc={sc.num_cuboids} {W}x{H} depth={depth} spp={spp} frames={frames} batch={batch} parts={parts} "
            f"focal={focal} aperture={aperture}")
    if os.environ.get("FUZZ_VERBOSE"):
        print(f"case {case}: {desc}", flush=True)
    try:
        for _ in range(frames): pt.Render()
        got = pt.Result
    except Exception as e:  # an error code from the library is a failure of the case, not of the fuzzer
        bad += 1
        print(f"case {case}: {e}: {desc}", flush=True)
        pt.Dispose()
        continue
    pt.Dispose()
    want = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_spheres=sc.num_spheres, num_cuboids=sc.num_cuboids, ray_depth=depth,
                         spp=spp, focal_length=focal, aperture=aperture, num_frames=frames)
    same = (got.view(np.uint32) == want.view(np.uint32)).all(-1)
    if not same.all():
        bad += 1
        print(f"case {case}: {int((~same).sum())}/{same.size} pixels differ: ns={sc.num_spheres} nc={sc.num_cuboids} {W}x{H} depth={depth} spp={spp} "
              f"frames={frames} batch={batch} parts={parts} focal={focal} aperture={aperture}", flush=True)
print(f"{cases} cases, {bad} with differences, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
