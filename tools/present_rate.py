"""Interactive-mode rates (development helper): one Render() + one read-back per displayed frame, i.e. what the
reference's loop does when it shows every frame (MainWindow.cs:43-52) — nothing can be pipelined.  RGBA32F read-back vs
the fused RGBA8 present.  (Page-locked destinations were measured too: no difference to pageable numpy arrays, the
runtime already moves 33 MB at ~48 GB/s.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
W, H = 1920, 1080
sc, cam = pkg.scene.default_scene(), pkg.camera.Camera()
pt = pkg.PathTracer(pkg.envmap.synthetic_sky_rgba32f(64), W, H, 8, 1, 20.0, 0.14)
pt.UploadScene(sc); pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, W, H))
for _ in range(64): pt.Render()
pt.Synchronize()
def rate(name, fn, n=100):
    for _ in range(5): fn()
    t = time.perf_counter()
    for _ in range(n): fn()
    ms = (time.perf_counter() - t) * 1e3 / n
    print(f"{name:58s} {ms:7.3f} ms per displayed frame  {W * H / ms / 1e3:8.1f} Msamples/s")
pageable_f, pageable_b = np.empty((H, W, 4), np.float32), np.empty((H, W, 4), np.uint8)
def go(read, dst):
    def f():
        pt.Render(); read(dst)
    return f
rate("render + pt_read_result (RGBA32F)", go(pt.ReadInto, pageable_f))
rate("render + pt_present_rgba8 (ACES + gamma, RGBA8)", go(pt.PresentInto, pageable_b))
