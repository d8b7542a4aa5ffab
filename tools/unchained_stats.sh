#!/bin/bash
# The default workload's rocprofv3 --kernel-trace --stats pass once more with PT_CHAIN_WAIT_US=0 (run on the GPU box):
#   bash tools/unchained_stats.sh <round>          -> gpurun_out/<round>/<round>_default_{kernel_stats,stats}_unchained.{csv,json}
# With back-pressure chaining (the default) every launch waits beside its predecessor, so the SUM of launch durations in
# <round>_default_kernel_stats.csv is ~2x the elapsed time; without it the launches go behind each other and sum / frames is a time
# per frame that can be compared with bench.py's HIP-event kernel_ms of the same run.
export TMPDIR=/tmp
R=/root/repo; RD=${1:-r03}; OUT=$R/gpurun_out/prof_${RD}_unchained; DST=$R/gpurun_out/$RD
mkdir -p $OUT $DST
STEPS=${PROFILE_STEPS:-640}; WARM=${PROFILE_WARMUP:-320}
cd /tmp
PT_CHAIN_WAIT_US=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- \
  python $R/bench.py --steps $STEPS --warmup $WARM --clock-warmup-ms 0 --steady-ms 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/stats.log
python - <<PY
import csv, glob, json, sys
sys.path.insert(0, "$R")
import __graft_entry__ as g
frames = $STEPS + $WARM
f = (glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True) + glob.glob("$OUT/stats/*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
open("$DST/${RD}_default_kernel_stats_unchained.csv", "w").write(open(f).read())
r = [x for x in rows if "pt_integrate" in x["Name"]][0]
bench = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
out = {"what": "default workload, rocprofv3 --kernel-trace --stats with PT_CHAIN_WAIT_US=0 (tools/unchained_stats.sh): launches do not overlap, "
               "sum of durations / frames is comparable with bench.py's HIP-event kernel_ms of the same run",
       "frames": frames,
       "unchained": {"calls": int(r["Calls"]), "total_ns": int(r["TotalDurationNs"]), "avg_ns": float(r["AverageNs"]),
                     "ns_per_frame": int(r["TotalDurationNs"]) / frames,
                     "bench_kernel_ms_same_run": bench["roofline"].get("kernel_ms"), "bench_value": bench["value"]},
       "csrc_hash": g.load_package().native.csrc_hash()}
json.dump(out, open("$DST/${RD}_default_stats_unchained.json", "w"), indent=1)
print(json.dumps(out["unchained"]))
PY
find $OUT -name "*_kernel_trace.csv" -delete
