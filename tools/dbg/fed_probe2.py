"""Debug probe (round 6): timing of the interactive modes with and without frame-fed launches."""
import sys, time, os
import numpy as np
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

def tracer():
    pt = pkg.PathTracer(env, W, H, 8, 1, 20.0, 0.14)
    pt.UploadScene(sc); pt.UploadBasicData(basic)
    for _ in range(200):
        pt.Render()
    pt.Synchronize()
    return pt

for feed in (1, 0, 1, 0):
    pkg.native.debug_set("feed", feed)
    pt = tracer()
    pt.SetFrameBatch(1)
    for _ in range(64): pt.Render()
    pt.Synchronize()
    s0 = pkg.native.debug_launch_stats(pt._h)
    n = 512; t = time.perf_counter()
    for _ in range(n): pt.Render()
    pt.Synchronize()
    el = time.perf_counter() - t
    s1 = pkg.native.debug_launch_stats(pt._h)
    print(f"feed={feed} batch1 render-only: {el * 1e3 / n:.4f} ms/frame; launches {s1['launches'] - s0['launches']} published {s1['published'] - s0['published']} idle {s1['feed_idle'] - s0['feed_idle']}")
    pt.SetFrameBatch(0)
    bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for s_, b_ in enumerate(bufs): pt.BindPresentImage(s_, b_.data_ptr(), b_.numel())
    seen = [False, False]
    def show(i):
        pt.Render()
        if seen[i & 1]: pt.PresentWait(i & 1)
        pt.PresentAsync(i & 1); seen[i & 1] = True
    for i in range(32): show(i)
    pt.Synchronize()
    s0 = pkg.native.debug_launch_stats(pt._h)
    n = 400; t = time.perf_counter()
    for i in range(n): show(i)
    for s_ in range(2): pt.PresentWait(s_)
    pt.Synchronize()
    el = time.perf_counter() - t
    s1 = pkg.native.debug_launch_stats(pt._h)
    print(f"feed={feed} displayed (bound device image): {el * 1e3 / n:.4f} ms/frame; launches {s1['launches'] - s0['launches']} published {s1['published'] - s0['published']} idle {s1['feed_idle'] - s0['feed_idle']}")
    for s_ in range(2): pt.BindPresentImage(s_, None)
    print("   handover:", pkg.native.debug_handover_stats(pt._h))
    pt.Dispose()
