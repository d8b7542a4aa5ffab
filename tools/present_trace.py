"""Timeline of the non-blocking present loop (development aid): run under
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ptrace -- python tools/present_trace.py [slots]
and summarise with tools/present_trace.py --summarise gpurun_out/ptrace"""
import os, sys, glob, csv
if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), "-", r.get("Stream_Id", "?")))
    rows.sort()
    n = len(rows)
    sel = rows[int(n * 0.6):int(n * 0.6) + 40]
    t0 = sel[0][0]
    for s, e, name, q, st in sel:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{q} s{st}  {name}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H = 1920, 1080
pt = pkg.PathTracer(None, W, H, 8, 1, 20.0, 0.14)
pt.EnvironmentMap = pkg.AtmosphericScatterer(256, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)
pt.UploadScene(pkg.scene.default_scene()); pt.UploadBasicData(pkg.camera.basic_data_ubo(pkg.camera.Camera(), W, H))
for _ in range(200): pt.Render()
pt.Synchronize()
seen = [False] * slots
for i in range(120):
    pt.Render()
    if seen[i % slots]:
        pt.PresentWait(i % slots)
    pt.PresentAsync(i % slots)
    seen[i % slots] = True
pt.Synchronize()
