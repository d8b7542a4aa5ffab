"""Debug probe (round 6): the displayed loop with 2 or 3 present slots (bound device images), classic and frame-fed.
   usage: present_slots_probe.py W H slots [knob=value ...]
   2 slots (bench.py's displayed_frame): Render(i); PresentWait(i & 1) [frame i - 2]; PresentAsync(i & 1)
   3 slots (host/pt_host_demo.cpp frame-loop): Render(i); PresentAsync(i % 3); PresentWait((i + 1) % 3) [frame i - 2]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import configs
pkg = configs.pkg
import torch
import numpy as np
W, H, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for kv in sys.argv[4:]:
    k, v = kv.split("="); pkg.native.debug_set(k, int(v))
w = configs.Workload("probe", "default", W, H, 8, "sky_f32_32")
sc, basic, objs, env, kw = configs.inputs(w)
pt = pkg.PathTracer(env, W, H, 8, 1, 20.0, 0.14)
pt.UploadScene(sc); pt.UploadBasicData(basic)
for _ in range(200): pt.Render()
pt.Synchronize()
bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(S)]
torch.cuda.synchronize()
for s_, b_ in enumerate(bufs): pt.BindPresentImage(s_, b_.data_ptr(), b_.numel())
seen = [False] * S
def show(i):
    pt.Render()
    if S == 2:
        if seen[i & 1]: pt.PresentWait(i & 1)
        pt.PresentAsync(i & 1); seen[i & 1] = True
    else:
        pt.PresentAsync(i % S); seen[i % S] = True
        if seen[(i + 1) % S]: pt.PresentWait((i + 1) % S)
for rep in range(3):
    for i in range(32): show(i)
    pt.Synchronize()
    s0 = pkg.native.debug_launch_stats(pt._h)
    n = 402; t = time.perf_counter()
    for i in range(n): show(i)
    for s_ in range(S):
        if seen[s_]: pt.PresentWait(s_)
    pt.Synchronize()
    el = time.perf_counter() - t
    s1 = pkg.native.debug_launch_stats(pt._h)
    print(f"{S} slots {sys.argv[4:]}: displayed {el * 1e3 / n:.4f} ms/frame; launches {s1['launches'] - s0['launches']} published {s1['published'] - s0['published']} idle {s1['feed_idle'] - s0['feed_idle']}", flush=True)
