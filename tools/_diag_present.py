import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
W, H = 1920, 1080
sc, cam = pkg.scene.default_scene(), pkg.camera.Camera()
pt = pkg.PathTracer(pkg.envmap.synthetic_sky_rgba32f(64), W, H, 8, 1, 20.0, 0.14)
pt.UploadScene(sc); pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, W, H))
t = time.perf_counter()
while time.perf_counter() - t < 0.15:
    for _ in range(64): pt.Render()
    pt.Synchronize()
def run(name, fn, n):
    for i in range(4): fn(i)
    pt.Synchronize()
    t = time.perf_counter()
    for i in range(n): fn(i)
    pt.Synchronize()
    print(f"{name:50s} {(time.perf_counter()-t)*1e3/n:7.3f} ms", flush=True)
def nb(i):
    pt.Render(); pt.PresentAsync(i & 1)
    if i > 0: pt.PresentWait((i - 1) & 1)
def only_present(i):
    pt.PresentAsync(i & 1)
    if i > 0: pt.PresentWait((i - 1) & 1)
def only_render_sync(i):
    pt.Render(); pt.Synchronize()
pt.SetFrameBatch(1)
def only_render(i):
    pt.Render()
run("render only, batch 1, no sync", only_render, 200)
run("render + Synchronize per frame", only_render_sync, 100)
run("present async only (tone map + copy)", only_present, 100)
run("render + async present", nb, 100)
