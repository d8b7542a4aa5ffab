import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import configs
pkg = configs.pkg
W, H = 1920, 1080
for kv in sys.argv[1:]:
    k, v = kv.split("="); pkg.native.debug_set(k, int(v))
w = configs.Workload("probe", "default", W, H, 8, "sky_f32_32")
sc, basic, objs, env, kw = configs.inputs(w)
pt = pkg.PathTracer(env, W, H, 8, 1, 20.0, 0.14)
pt.UploadScene(sc); pt.UploadBasicData(basic)
for _ in range(300): pt.Render()
pt.Synchronize()
pt.SetFrameBatch(1)
for k in (1, 2, 4, 16):
    for _ in range(10):
        for _ in range(k): pt.Render()
        pt.Synchronize()
    s0 = pkg.native.debug_launch_stats(pt._h)
    t = time.perf_counter(); n = 40
    for _ in range(n):
        for _ in range(k): pt.Render()
        pt.Synchronize()
    el = time.perf_counter() - t
    s1 = pkg.native.debug_launch_stats(pt._h)
    print(f"{sys.argv[1:]}: {k} Render() + Synchronize(): {el * 1e6 / n:.1f} us per group = {el * 1e6 / n / k:.1f} us per frame; fed opens {s1['feed_opens'] - s0['feed_opens']} published {s1['published'] - s0['published']}")
