"""What an idle gap before a short timed region costs: kernel time (HIP events on the library's stream) of ONE flush of n frames, after
`busy` ms of continuous rendering and `gap` ms of host sleep.  The driver's command (`bench.py --steps 20 --warmup 5`) times 2.3 ms of GPU
work right after a barrier + synchronize; the steady state of the same workload is ~7 % faster per frame.  Is that the launch (fill +
drain of one flush) or the GPU's clock state?   Usage: python tools/clock_probe.py [--json gpurun_out/clock_probe.json]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
ap = argparse.ArgumentParser()
ap.add_argument("--json", default=None)
a = ap.parse_args()
pkg = g.load_package()
W, H = 1920, 1080
sc, cam = pkg.scene.default_scene(), pkg.camera.Camera()
pt = pkg.PathTracer(None, W, H, 8, 1, 20.0, 0.14)
pt.EnvironmentMap = pkg.AtmosphericScatterer(256, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)
pt.UploadScene(sc); pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, W, H))
def busy(ms):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        for _ in range(64): pt.Render()
    pt.Synchronize()
def flush(n):
    pt.TimerBegin()
    for _ in range(n): pt.Render()
    return pt.TimerEnd()
out = []
for busy_ms in (80, 400):
    for gap_ms in (0, 0.2, 1, 5, 20, 100):
        for n in (20, 64):
            v = []
            for rep in range(5):
                busy(busy_ms)
                if gap_ms: time.sleep(gap_ms / 1e3)
                v.append(flush(n))
            v.sort()
            out.append({"busy_ms": busy_ms, "gap_ms": gap_ms, "frames": n, "kernel_ms_median": round(v[2], 4), "per_frame": round(v[2] / n, 5), "min": round(v[0], 4), "max": round(v[4], 4)})
            print(out[-1], flush=True)
# back-to-back flushes without any gap: the n-th flush of 20 frames in a row (each timed on its own)
busy(200)
seq = [round(flush(20) / 20, 5) for _ in range(12)]
print("12 flushes of 20 frames back to back (ms per frame):", seq)
if a.json:
    json.dump({"runs": out, "back_to_back_20": seq}, open(a.json, "w"), indent=1)
