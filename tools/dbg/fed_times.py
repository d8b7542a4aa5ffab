import sys, time, os, ctypes as C
os.environ["MI355PT_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfeedtimes.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import configs
pkg = configs.pkg
import torch, numpy as np
W, H = 1920, 1080
pkg.native.debug_set("feed_idle_us", 20000)
for kv in sys.argv[1:]:
    k, v = kv.split("="); pkg.native.debug_set(k, int(v))
w = configs.Workload("probe", "default", W, H, 8, "sky_f32_32")
sc, basic, objs, env, kw = configs.inputs(w)
pt = pkg.PathTracer(env, W, H, 8, 1, 20.0, 0.14)
pt.UploadScene(sc); pt.UploadBasicData(basic)
for _ in range(200): pt.Render()
pt.Synchronize()
NS = int(os.environ.get("SLOTS", "2"))
bufs = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(NS)]
torch.cuda.synchronize()
for s_, b_ in enumerate(bufs): pt.BindPresentImage(s_, b_.data_ptr(), b_.numel())
seen = [False] * NS
hostt = []
def show(i):
    pt.Render(); t1 = time.perf_counter()
    if seen[i % NS]: pt.PresentWait(i % NS)
    t2 = time.perf_counter()
    pt.PresentAsync(i % NS); seen[i % NS] = True
    hostt.append((t1, t2))
for i in range(40): show(i)
pt.Synchronize()
L = pkg.native.load()
out = (C.c_uint * 128)()
L.pt_debug_feed_times(pt._h, out)
done = np.array(out[:64], dtype=np.int64); pub = np.array(out[64:], dtype=np.int64)
print("device: frame-complete clock deltas (us):", [int((done[j] - done[j - 1]) % 2**32) / 100 for j in range(2, 24)])
print("device: publish seen -> that frame complete (us):", [int((done[j] - pub[j]) % 2**32) / 100 for j in range(2, 24)])
print("host: wait durations (us):", [round((b - a) * 1e6) for a, b in hostt[5:29]])
print("host: render-to-render (us):", [round((hostt[i + 1][0] - hostt[i][0]) * 1e6) for i in range(5, 29)])
