"""Soak test of the frame pipelining (development helper): N consecutive frames at full size, sparse pixels compared
bit for bit with the oracle's frame-by-frame accumulation.  python tools/soak.py [frames] [W H] [scene depth spp [parts]]
MI355PT_LIB=opentk-pathtracer_amd/libmi355pt_audit.so: the kernels' own hand-over audit (every pixel read-modify-write mirrored by a
device-scope atomic side word) is read at the end — full-size images put the tag protocol under real cross-XCD load."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as g
import configs
pkg = g.load_package()
oracle = g.load_oracle().Oracle()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
scene, depth, spp = (sys.argv[4], int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else ("default", 8, 1)
parts = int(sys.argv[7]) if len(sys.argv) > 7 else 0  # > 0: a group handle over that many copies of device 0
w = configs.Workload("soak", scene, W, H, depth, "sky_f32_32", frames=frames, spp=spp)
sc, basic, objs, env, kw = configs.inputs(w)
pt = pkg.PathTracer(env, W, H, w.ray_depth, spp, w.focal_length, w.aperture, **({"devices": [0] * parts} if parts else {}))
pt.UploadScene(sc); pt.UploadBasicData(basic)
t = time.perf_counter()
for _ in range(frames): pt.Render()
pt.Synchronize()
dt = time.perf_counter() - t
got = pt.Result
rng = np.random.RandomState(3)
xy = np.stack([rng.randint(0, W, 384), rng.randint(0, H, 384)], 1)
want = None
for f in range(frames):
    want = oracle.render_pixels(W, H, basic, objs, env, xy, frame=f, last=want, **kw)
same = (got[xy[:, 1], xy[:, 0]].view(np.uint32) == want.view(np.uint32)).all(-1)
import ctypes as C
fn = getattr(pt._lib, "pt_debug_audit_read", None)
viol = None
if fn is not None:
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.c_int]; fn.restype = C.c_int
    viol = fn(pt._h, None, 0)
audit = "not an audit build" if viol in (None, -1000) else f"{viol} audit violations in {frames * W * H / 1e9:.2f} G audited resolves"
print(f"{frames} frames {W}x{H} {scene} depth {depth} spp {spp} parts {parts}: {dt / frames * 1e3:.4f} ms/frame, alpha==1: {(got[..., 3] == 1).all()}, "
      f"sparse pixels bit-identical to the oracle: {same.sum()}/{len(same)}; {audit}", flush=True)
sys.exit(0 if same.all() and (got[..., 3] == 1).all() and not (viol and viol > 0) else 1)
