"""Loop statistics of the spp = 1 integrator on one rank's share (instrumented scratch build tools/ab/libT.so: loop iterations, iterations
without an active lane, parked-list services and their length, tile passes, try_resolve calls).  Development helper.
Usage: MI355PT_LIB=tools/ab/libT.so python tools/ab/share_loop_counts.py <world> [tune=value,...]"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pkg = g.load_package()
for kv in filter(None, (sys.argv[2] if len(sys.argv) > 2 else "").split(",")):
    k, v = kv.split("="); pkg.native.debug_set(k, int(v))
lib = pkg.native.load()
W, H = 1920, 1080
sc, cam = pkg.scene.default_scene(), pkg.camera.Camera()
pt = pkg.PathTracer(None, W, H, 8, 1, 20.0, 0.14)
pt.EnvironmentMap = pkg.AtmosphericScatterer(256, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)
pt.UploadScene(sc); pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, W, H))
if world > 1: pt.SetInterleavedTile(world // 2, world, 8)
for _ in range(256): pt.Render()
pt.Synchronize()
lib.pt_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.pt_debug_timeline(pt._h, None, 0)
frames = 640
pt.TimerBegin()
for _ in range(frames): pt.Render()
ms = pt.TimerEnd() / frames
buf = np.zeros((65536, 4), np.uint64)
lib.pt_debug_timeline(pt._h, buf.ctypes.data_as(C.c_void_p), 65536)
c = buf.reshape(-1)[250000:250008].astype(np.float64) / frames * world   # per whole-frame equivalent
print(f"world {world} [{sys.argv[2] if len(sys.argv) > 2 else ''}]: {ms:.4f} ms per share frame; per whole-frame equivalent: loop iterations {c[0]:.0f}, without an active lane {c[1]:.0f}, "
      f"parked-list services {c[2]:.0f} (mean length {c[3] / max(c[2], 1):.1f}), tile passes {c[4]:.0f}, try_resolve calls {c[5]:.0f}")
