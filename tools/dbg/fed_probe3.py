"""Debug probe: displayed loop (bound device images) with frame-fed launches under debug present modes."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import configs
pkg = configs.pkg
import torch
W, H = int(sys.argv[1]), int(sys.argv[2])
pkg.native.debug_set("feed_min_tiles", 0)
for kv in sys.argv[3:]:
    k, v = kv.split("="); pkg.native.debug_set(k, int(v))
w = configs.Workload("probe", "default", W, H, 8, "sky_f32_32")
sc, basic, objs, env, kw = configs.inputs(w)
pt = pkg.PathTracer(env, W, H, 8, 1, 20.0, 0.14)
pt.UploadScene(sc); pt.UploadBasicData(basic)
for _ in range(200): pt.Render()
pt.Synchronize()
bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
torch.cuda.synchronize()
for s_, b_ in enumerate(bufs): pt.BindPresentImage(s_, b_.data_ptr(), b_.numel())
seen = [False, False]
waits = []
def show(i):
    pt.Render()
    t = time.perf_counter()
    if seen[i & 1]: pt.PresentWait(i & 1)
    waits.append((time.perf_counter() - t) * 1e6)
    pt.PresentAsync(i & 1); seen[i & 1] = True
for i in range(32): show(i)
pt.Synchronize()
s0 = pkg.native.debug_launch_stats(pt._h)
del waits[:]
n = int(os.environ.get("PROBE_FRAMES", "400")); t = time.perf_counter()
for i in range(n): show(i)
for s_ in range(2): pt.PresentWait(s_)
pt.Synchronize()
el = time.perf_counter() - t
s1 = pkg.native.debug_launch_stats(pt._h)
import numpy as np
wa = np.array(waits)
print(f"{sys.argv[3:]}: displayed {el * 1e3 / n:.4f} ms/frame; launches {s1['launches'] - s0['launches']} published {s1['published'] - s0['published']} idle {s1['feed_idle'] - s0['feed_idle']}; "
      f"PresentWait us: median {np.median(wa):.0f} p90 {np.quantile(wa, 0.9):.0f} max {wa.max():.0f}; first 12: {[int(x) for x in wa[:12]]}")
