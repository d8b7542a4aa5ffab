"""One isolated flush of n frames under `rocprofv3 --kernel-trace`: which kernels run, when (development helper for profiles/r05/short_region_probe.log).
Usage: rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/flush_trace.py [n]; python tools/flush_trace.py --read <dir>"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    import csv
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tail = rows[-int(sys.argv[3]) if len(sys.argv) > 3 else -12:]
    t0 = int(tail[0]["Start_Timestamp"])
    for r in tail:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"{s/1e3:10.1f} us -> {e/1e3:10.1f} us  ({(e-s)/1e3:8.1f} us)  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8}  queue {r.get('Queue_Id','?'):>3}  {r['Kernel_Name'][:90]}")
    sys.exit(0)
import __graft_entry__ as g
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pkg = g.load_package()
W, H = 1920, 1080
sc, cam = pkg.scene.default_scene(), pkg.camera.Camera()
pt = pkg.PathTracer(None, W, H, 8, 1, 20.0, 0.14)
pt.EnvironmentMap = pkg.AtmosphericScatterer(256, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)
pt.UploadScene(sc); pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, W, H))
t = time.perf_counter()
while time.perf_counter() - t < 0.1:
    for _ in range(64): pt.Render()
pt.Synchronize()
pt.TimerBegin()
for _ in range(n): pt.Render()
print("flush of", n, "frames: kernel ms", pt.TimerEnd())
