"""Ad-hoc GPU probe: parity vs oracle on several configs + quick timing. (development helper, not a test)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); po = g.load_oracle()
oracle = po.Oracle()
cam = pkg.camera.Camera()

def run(name, sc, W, H, depth, frames, env, spp=1, compare=True, variant=0):
    basic = pkg.camera.basic_data_ubo(cam, W, H)
    pt = pkg.PathTracer(env, W, H, depth, spp, 20.0, 0.14)
    pt.SetVariant(variant)
    pt.UploadScene(sc); pt.UploadBasicData(basic)
    for _ in range(frames): pt.Render()
    got = pt.Result
    msg = f"{name}: {W}x{H} d{depth} spp{spp} f{frames} v{variant}"
    if compare:
        want = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_spheres=sc.num_spheres, num_cuboids=sc.num_cuboids, ray_depth=depth, spp=spp, num_frames=frames)
        same = (got.view(np.uint32) == want.view(np.uint32)).all(-1)
        msg += f" bit-identical pixels {same.mean()*100:.4f}% ({(~same).sum()} differ) maxabs {np.nanmax(np.abs(got-want)):.3g}"
        if not same.all():
            ys, xs = np.nonzero(~same)
            for y, x in list(zip(ys, xs))[:5]: msg += f"\n    ({x},{y}) hip={got[y,x,:3]} oracle={want[y,x,:3]}"
    # timing
    pt.ResetRenderer()
    for _ in range(3): pt.Render()
    pt.Synchronize()
    n = 20
    pt.TimerBegin()
    for _ in range(n): pt.Render()
    ms = pt.TimerEnd() / n
    msg += f" | {ms:.4f} ms/frame, {W*H*spp/ms/1e3:.1f} Msamples/s"
    print(msg, flush=True)
    pt.Dispose()

sky = pkg.envmap.synthetic_sky_rgba32f(64)
srgb = pkg.envmap.synthetic_sky_srgb8(64)
tiny = pkg.envmap.tiny_test_cube(4)
run("default", pkg.scene.default_scene(), 128, 72, 8, 2, sky)
run("default-srgb", pkg.scene.default_scene(), 128, 72, 8, 2, srgb)
run("default-odd", pkg.scene.default_scene(), 131, 75, 8, 1, sky)
run("empty-tiny", pkg.scene.Scene(), 256, 144, 2, 1, tiny)
run("stress", pkg.scene.stress_scene(), 256, 144, 8, 1, sky)
run("glass32", pkg.scene.glass_scene(), 256, 144, 32, 1, sky)
run("randmat-spp3", pkg.scene.random_material_scene(), 256, 144, 13, 2, sky, spp=3)
run("default-1080p", pkg.scene.default_scene(), 1920, 1080, 8, 2, sky)
run("stress-1080p", pkg.scene.stress_scene(), 1920, 1080, 8, 1, sky, compare=False)
run("glass-1080p", pkg.scene.glass_scene(), 1920, 1080, 32, 1, sky, compare=False)
