"""Debug probe (round 6): the present loop over frame-fed launches, with per-iteration stats and timings."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import configs
pkg = configs.pkg
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (224, 126)
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if W * H < 1000000:
    pkg.native.debug_set("feed_min_tiles", 0)
pkg.native.debug_set("feed_log", int(os.environ.get("FEED_LOG", "0")))
pkg.native.debug_set("feed_idle_us", int(sys.argv[4]) if len(sys.argv) > 4 else 20000)
w = configs.Workload("probe", "default", W, H, 8, "sky_f32_32")
sc, basic, objs, env, kw = configs.inputs(w)
pt = pkg.PathTracer(env, W, H, 8, 1, 20.0, 0.14)
pt.UploadScene(sc); pt.UploadBasicData(basic)
for _ in range(100): pt.Render()
pt.Synchronize()
log = []
for f in range(frames):
    t0 = time.perf_counter(); pt.Render(); t1 = time.perf_counter()
    idx = None
    if f >= 2:
        _, idx = pt.PresentWait(f % 2)
    t2 = time.perf_counter(); pt.PresentAsync(f % 2); t3 = time.perf_counter()
    st = pkg.native.debug_launch_stats(pt._h)
    log.append((f, idx, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, st["feed_open"], st["published"], st["feed_opens"], st["feed_idle"], st["launches"]))
for r in log:
    print("f %3d shown %s render %7.1f us wait %9.1f us present %7.1f us | open %d published %d opens %d idle %d launches %d" % r)
pt.Synchronize()
print(pkg.native.debug_handover_stats(pt._h))
