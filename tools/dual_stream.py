"""Experiment: hide the frame-end drain by splitting the image into row blocks rendered by independent persistent
kernels on separate streams (handles), oversubscribed so that one block's drain is covered by the other's backlog."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
W, H = 1920, 1080
sc, cam = pkg.scene.default_scene(), pkg.camera.Camera()
env = pkg.envmap.synthetic_sky_rgba32f(64)
basic = pkg.camera.basic_data_ubo(cam, W, H)
def make(y0, rows, variant):
    pt = pkg.PathTracer(env, W, H, 8, 1, 20.0, 0.14)
    pt.SetVariant(variant); pt.UploadScene(sc); pt.UploadBasicData(basic); pt.SetTile(y0, rows)
    return pt
def run(parts, variant, frames=200):
    bounds = [i * H // parts for i in range(parts + 1)]
    pts = [make(bounds[i], bounds[i+1]-bounds[i], variant) for i in range(parts)]
    for _ in range(10):
        for p in pts: p.Render()
    for p in pts: p.Synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        for p in pts: p.Render()
    for p in pts: p.Synchronize()
    dt = (time.perf_counter() - t0) / frames * 1e3
    print(f"parts {parts} variant {variant}: {dt:.4f} ms/frame  {W*H/dt/1e3:.0f} Msamples/s", flush=True)
    for p in pts: p.Dispose()
for variant in (13, 14, 12):
    for parts in (1, 2, 3, 4):
        run(parts, variant)
