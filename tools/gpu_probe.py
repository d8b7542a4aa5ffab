"""Ad-hoc GPU probe: parity vs oracle on several configs + quick timing per kernel variant. (development helper)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); po = g.load_oracle()
oracle = po.Oracle()
cam = pkg.camera.Camera()
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])]
_cache = {}

def run(name, sc, W, H, depth, frames, env, spp=1, compare=True, variant=0, time_it=True):
    basic = pkg.camera.basic_data_ubo(cam, W, H)
    pt = pkg.PathTracer(env, W, H, depth, spp, 20.0, 0.14)
    pt.SetVariant(variant)
    pt.UploadScene(sc); pt.UploadBasicData(basic)
    for _ in range(frames): pt.Render()
    got = pt.Result
    msg = f"v{variant} {name}: {W}x{H} d{depth} spp{spp} f{frames}"
    if compare:
        key = (name, W, H, depth, spp, frames)
        if key not in _cache:
            _cache[key] = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_spheres=sc.num_spheres, num_cuboids=sc.num_cuboids, ray_depth=depth, spp=spp, num_frames=frames)
        want = _cache[key]
        same = (got.view(np.uint32) == want.view(np.uint32)).all(-1)
        msg += f" | bit-identical {same.mean()*100:.4f}% ({(~same).sum()} differ)"
        if not same.all():
            ys, xs = np.nonzero(~same)
            for y, x in list(zip(ys, xs))[:3]: msg += f"\n    ({x},{y}) hip={got[y,x]} oracle={want[y,x]}"
    if time_it:
        pt.ResetRenderer()
        for _ in range(20): pt.Render()
        pt.Synchronize()
        n, best = 150, 1e9
        for rep in range(4):
            pt.TimerBegin()
            for _ in range(n): pt.Render()
            best = min(best, pt.TimerEnd() / n)
        ms = best
        msg += f" | {ms:.4f} ms/frame (best of 4x{n}), {W*H*spp/ms/1e3:.1f} Msamples/s"
    print(msg, flush=True)
    pt.Dispose()

sky = pkg.envmap.synthetic_sky_rgba32f(64)
srgb = pkg.envmap.synthetic_sky_srgb8(64)
for v in variants:
    run("default", pkg.scene.default_scene(), 128, 72, 8, 2, sky, variant=v, time_it=False)
    run("default-srgb-odd", pkg.scene.default_scene(), 131, 75, 8, 2, srgb, variant=v, time_it=False)
    run("randmat-spp3", pkg.scene.random_material_scene(), 256, 144, 13, 2, sky, spp=3, variant=v, time_it=False)
    run("depth0", pkg.scene.default_scene(), 40, 24, 0, 1, sky, variant=v, time_it=False)
    run("default-1080p", pkg.scene.default_scene(), 1920, 1080, 8, 2, sky, variant=v)
    run("stress-1080p", pkg.scene.stress_scene(), 1920, 1080, 8, 1, sky, compare=False, variant=v)
    run("glass-1080p", pkg.scene.glass_scene(), 1920, 1080, 32, 1, sky, compare=False, variant=v)
    run("default-4k", pkg.scene.default_scene(), 3840, 2160, 8, 1, sky, compare=False, variant=v)
