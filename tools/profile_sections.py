"""Per-section cycle breakdown of the integrator's bounce iteration on PIPELINED launches (the mode bench.py measures).  Needs the
-DPT_PROFILE build of the tree (tools/ab/build_ab.sh P -DPT_PROFILE): every kernel builds with it, and the library keeps batching /
chaining while the counter buffer exists (csrc/pt_renderer.hpp, PT_TIMELINE_BLOCKS).
    MI355PT_LIB=tools/ab/libP.so python tools/profile_sections.py [default|stress|glass] [variant frames]
default: variant 0, 640 frames.  Variant 14 = one un-pipelined frame per launch.  PROFILE_SHARE=rank,world profiles one rank's share."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
lib = pkg.native.load()
scene_name = sys.argv[1] if len(sys.argv) > 1 else "default"
W, H = 1920, 1080
sc = {"default": pkg.scene.default_scene, "stress": pkg.scene.stress_scene, "glass": pkg.scene.glass_scene}[scene_name]()
cam = pkg.camera.Camera()
pt = pkg.PathTracer(pkg.envmap.synthetic_sky_rgba32f(64), W, H, 8 if scene_name != "glass" else 32, 1, 20.0, 0.14)
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pt.SetVariant(variant); pt.UploadScene(sc); pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, W, H))
if os.environ.get("PROFILE_SHARE"):  # "rank,world": one rank's block-cyclic share of the image (16-row bands), as bench.py --gpus N renders it
    rank, world = map(int, os.environ["PROFILE_SHARE"].split(","))
    pt.SetInterleavedTile(rank, world, 16)
    print(f"share: rank {rank} of {world}")
for _ in range(5 if variant else 128): pt.Render()
pt.Synchronize()
lib.pt_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.pt_debug_timeline(pt._h, None, 0)          # allocate + zero
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 640
for _ in range(frames): pt.Render()
pt.Synchronize()
buf = np.zeros((65536, 4), np.uint64)
lib.pt_debug_timeline(pt._h, buf.ctypes.data_as(C.c_void_p), 65536)
prof = buf.reshape(-1)[200000:200008].astype(np.float64) / frames
util = buf.reshape(-1)[200008:200011].astype(np.float64) / frames
if util[0] > 0:
    print(f"generic bounce iterations per frame {util[0]:.0f}, mean active lanes {util[1] / util[0]:.2f} of 64, "
          f"of which waiting for their pixel's previous frame {util[2] / util[0]:.3f}")
names = ["feed (refill/pop/adopt/donate)", "sphere pass", "cuboid pass (+3 rcp)", "winner material+normal", "Beer absorption",
         "BSDF", "tile pass (whole first bounce of a tile)", "RR + resolve + bookkeeping"]
tot = prof.sum()
print(f"scene {scene_name}, variant {variant}, {frames} frames: wave-cycles per frame (s_memtime ticks, summed over wavefronts): {tot:.3e}")
for n, v in zip(names, prof):
    print(f"  {n:34s} {100 * v / tot:6.2f} %")
