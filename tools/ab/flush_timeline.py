"""Per-wavefront timestamps (start, tickets exhausted, end) of the launches of ONE isolated flush of n frames, from an instrumented scratch
build (tools/ab/libT.so: every instantiation writes the timeline buffer, launches keep pipelining).  Development helper.
Usage: MI355PT_LIB=tools/ab/libT.so python tools/ab/flush_timeline.py [n]"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pkg = g.load_package()
lib = pkg.native.load()
W, H = 1920, 1080
sc, cam = pkg.scene.default_scene(), pkg.camera.Camera()
pt = pkg.PathTracer(None, W, H, 8, 1, 20.0, 0.14)
pt.EnvironmentMap = pkg.AtmosphericScatterer(256, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)
pt.UploadScene(sc); pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, W, H))
lib.pt_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.pt_debug_timeline(pt._h, None, 0)   # allocate the buffer: from here on every launch writes it
for rep in range(3):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.1:
        for _ in range(64): pt.Render()
    pt.Synchronize()
    buf = np.zeros((4 * 8192, 4), np.uint64)
    pt.TimerBegin()
    for _ in range(n): pt.Render()
    ms = pt.TimerEnd()
    lib.pt_debug_timeline(pt._h, buf.ctypes.data_as(C.c_void_p), 4 * 8192)
    # the two (or three) launches of the flush are the ones with the LATEST end stamps per slot of launchSeq & 3
    slots = [buf[k * 8192:(k + 1) * 8192] for k in range(4)]
    slots = [s[s[:, 2] > 0] for s in slots]
    slots = [s for s in slots if len(s)]
    slots.sort(key=lambda s: s[:, 0].min())
    recent = slots[-2:] if n > 1 else slots[-1:]
    t0 = min(float(s[:, 0].min()) for s in recent)
    us = lambda x: (x.astype(np.float64) - t0) / 100.0
    print(f"--- flush of {n} frames: kernel time {ms * 1e3:.1f} us (HIP events)")
    for s in recent:
        st, ex, en, it = us(s[:, 0]), us(s[:, 1]), us(s[:, 2]), s[:, 3].astype(np.float64)
        ex = np.where(s[:, 1] > 0, ex, en)
        q = lambda v, p: np.percentile(v, p)
        print(f"launch with {len(s)} wavefronts ({len(s) // 4} workgroups)")
        print(f"  start    : min {st.min():8.1f}  p10 {q(st,10):8.1f}  median {q(st,50):8.1f}  p90 {q(st,90):8.1f}  max {st.max():8.1f}")
        print(f"  exhausted: min {ex.min():8.1f}  p10 {q(ex,10):8.1f}  median {q(ex,50):8.1f}  p90 {q(ex,90):8.1f}  max {ex.max():8.1f}")
        print(f"  end      : min {en.min():8.1f}  p10 {q(en,10):8.1f}  median {q(en,50):8.1f}  p90 {q(en,90):8.1f}  max {en.max():8.1f}")
        print(f"  drain per wavefront (end - exhausted): median {q(en-ex,50):.1f}  mean {(en-ex).mean():.1f}  p90 {q(en-ex,90):.1f}  max {(en-ex).max():.1f} us;  iterations per wavefront {it.mean():.0f}")
        # machine occupancy over time: wavefronts resident (started, not ended) at a few instants
        grid = np.linspace(0, en.max(), 24)
        res = [(int(((st <= t) & (en > t)).sum()), int(((st <= t) & (ex > t)).sum())) for t in grid]
        print("  t(us): resident / still drawing tickets: " + "  ".join(f"{t:.0f}:{r}/{d}" for t, (r, d) in zip(grid, res)))
