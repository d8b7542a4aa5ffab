"""Print a compact timeline from a rocprofv3 kernel-trace CSV: start / end (us, relative) per kernel, last N rows."""
import csv, sys, glob
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"]
    short = "FED" if "ELb1EEEv" in name and "persistent" in name else ("integ" if "persistent" in name else ("gate" if "feed_wait" in name else ("tonemapT" if "tagged" in name else ("tonemap" if "postprocess" in name else name[:28]))))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{short:10s} q{r.get('Queue_Id', '?'):>3s} start {s:10.1f} end {e:10.1f} dur {e - s:9.1f} us")
