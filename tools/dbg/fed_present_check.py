import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import configs
import __graft_entry__ as graft
pkg = configs.pkg
oracle = graft.load_oracle().Oracle()
W, H, frames = 224, 126, 12
slow = len(sys.argv) > 1 and sys.argv[1] == "slow"
pkg.native.debug_set("feed_min_tiles", 0)
pkg.native.debug_set("feed_idle_us", 200000)
w = configs.Workload("probe", "default", W, H, 8, "sky_f32_32")
sc, basic, objs, env, kw = configs.inputs(w)
acc = oracle.render(W, H, basic, objs, env, num_frames=frames, dump_each=True, **kw)
pt = pkg.PathTracer(env, W, H, 8, 1, 20.0, 0.14)
pt.UploadScene(sc); pt.UploadBasicData(basic)
shown = {}
for f in range(frames):
    pt.Render()
    if slow: time.sleep(0.002)
    if f >= 2:
        img, idx = pt.PresentWait(f % 2); shown[idx] = img.copy()
    pt.PresentAsync(f % 2)
    st = pkg.native.debug_launch_stats(pt._h)
    print(f, "open", st["feed_open"], "published", st["published"], "opens", st["feed_opens"], "idle", st["feed_idle"])
for s_ in ((frames - 2) % 2, (frames - 1) % 2):
    img, idx = pt.PresentWait(s_); shown[idx] = img.copy()
for idx in sorted(shown):
    want = oracle.postprocess(acc[idx - 1])[1]
    bad = (shown[idx] != want)
    print("frame", idx, "mismatching bytes per channel", bad.reshape(-1, 4).sum(0), "of", W * H, "; first bad pixels", np.argwhere(bad.any(-1))[:4].tolist())
    if bad.any():
        y, x = np.argwhere(bad.any(-1))[0]
        print("   got", shown[idx][y, x], "want", want[y, x], "acc", acc[idx - 1][y, x])
