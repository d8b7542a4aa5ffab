"""Per-wavefront timeline of the persistent kernel (development helper)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
lib = pkg.native.load()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
sc, cam = pkg.scene.default_scene(), pkg.camera.Camera()
pt = pkg.PathTracer(pkg.envmap.synthetic_sky_rgba32f(64), W, H, 8, 1, 20.0, 0.14)
pt.SetVariant(variant); pt.UploadScene(sc); pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, W, H))
for _ in range(5): pt.Render()
lib.pt_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.pt_debug_timeline(pt._h, None, 0)
pt.Render(); pt.Synchronize()
bpc = 4 if variant == 0 else variant - 9
nw = 256 * bpc * 4
buf = np.zeros((nw, 4), np.uint64)
lib.pt_debug_timeline(pt._h, buf.ctypes.data_as(C.c_void_p), nw)
buf = buf[buf[:, 2] > 0]
t0 = buf[:, 0].min()
us = lambda x: (x.astype(np.float64) - float(t0)) / 100.0  # wall_clock64 = 100 MHz
start, exh, end, it = us(buf[:, 0]), us(buf[:, 1]), us(buf[:, 2]), buf[:, 3].astype(np.float64)
print(f"waves {len(buf)}  kernel span {end.max():.1f} us")
print(f"start   : min {start.min():.1f} median {np.median(start):.1f} max {start.max():.1f}")
print(f"exhaust : min {exh.min():.1f} median {np.median(exh):.1f} max {exh.max():.1f}")
print(f"end     : min {end.min():.1f} median {np.median(end):.1f} p90 {np.percentile(end,90):.1f} max {end.max():.1f}")
print(f"drain (end-exhaust): median {np.median(end-exh):.1f} mean {(end-exh).mean():.1f} max {(end-exh).max():.1f}")
print(f"iterations per wave: mean {it.mean():.1f} min {it.min():.0f} max {it.max():.0f} total {it.sum():.0f};  us/iteration {((end-start)/it).mean():.2f}")
busy = (end - start).sum() / (len(buf) * end.max())
print(f"wave residency (sum(end-start) / waves*span): {busy:.3f}")
