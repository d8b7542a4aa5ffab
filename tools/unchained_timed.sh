#!/bin/bash
# The headline workload's rocprofv3 --kernel-trace --stats pass with launches that do NOT overlap (--tune serial_launches=1) and the usual clock
# warm-up: the last 10 integrator launches are the timed region's 640 frames (10 launches of 64), so their average duration is directly
# comparable with bench.py's HIP-event kernel time of the same run (kernel_ms x frames_per_launch).  -> gpurun_out/<round>/<round>_default_stats_unchained_timed.json
R=/root/repo; RND=${1:-r04}; OUT=/tmp/unch_$$; mkdir -p $OUT $R/gpurun_out/$RND; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $R/bench.py --steps 640 --warmup 320 --steady-ms 0 --no-cpu-baseline --tune serial_launches=1 > $OUT/bench.json 2> $OUT/stats.log
python - "$OUT" "$R/gpurun_out/$RND" "$RND" <<'PY'
import csv, glob, json, sys
out, dst, rnd = sys.argv[1:4]
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
f = (glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True) + glob.glob(out + "/stats/*kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(f)) if "pt_integrate" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
last = dur[-10:]
bench = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
k = bench["roofline"]["kernel_ms"]; fpl = bench["roofline"]["frames_per_launch"]
res = {"what": "default workload with launches that do not overlap (--tune serial_launches=1): the timed region's 640 frames are the LAST 10 integrator "
               "launches of 64 frames; their rocprofv3 durations against bench.py's HIP-event kernel time of the same run",
       "integrator_launches_total": len(dur), "last_10_launch_durations_ns": last, "avg_ns_last_10": sum(last) / len(last),
       "ns_per_frame_last_10": sum(last) / 640.0, "bench_kernel_ms_same_run": k, "bench_frames_per_launch": fpl,
       "bench_ns_per_launch": k * 1e6 * fpl, "ratio_rocprof_to_events": (sum(last) / len(last)) / (k * 1e6 * fpl),
       "bench_value": bench["value"], "csrc_hash": g.load_package().native.csrc_hash()}
json.dump(res, open(f"{dst}/{rnd}_default_stats_unchained_timed.json", "w"), indent=1)
print(json.dumps({a: b for a, b in res.items() if a != "what"}))
PY
rm -rf $OUT
