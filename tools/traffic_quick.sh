#!/bin/bash
# HBM-side traffic of the integrator kernel from the L2's memory-side request counters, exact by request size (no
# calibration needed): tools/traffic_quick.sh <tag> [bench args...]   -> gpurun_out/traffic_<tag>/summary.json
# Separate --pmc passes with --kernel-trace only (task rules); counters are averaged per launch and divided by the frames
# a launch renders.
export TMPDIR=/tmp
R=/root/repo
TAG=$1
OUT=$R/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp
STEPS=${TRAFFIC_STEPS:-384}
BENCH="python $R/bench.py --steps $STEPS --warmup 192 --no-cpu-baseline --clock-warmup-ms 0 --steady-ms 0 ${@:2}"
export TRAFFIC_TOTAL_FRAMES=$((STEPS + 192))
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o $1 -- $BENCH > $OUT/$1.log 2>&1; }
run rd "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
run wr "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"
if [ -n "$TRAFFIC_FULL" ]; then run fetch "FETCH_SIZE"; run write "WRITE_SIZE"; fi
python - "$OUT" "$TAG" "${@:2}" <<'PY'
import csv, collections, json, sys, os
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3:]
fb = 64
if "--frame-batch" in args: fb = int(args[args.index("--frame-batch") + 1])
if "--variant" in args and int(args[args.index("--variant") + 1]) != 0: fb = 1
acc = collections.defaultdict(list)
for n in ("rd", "wr", "fetch", "write"):
    p = os.path.join(out, n, f"{n}_counter_collection.csv")
    if not os.path.exists(p): continue
    for r in csv.DictReader(open(p)):
        if "pt_integrate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
# every frame of the run (warm-up + timed) goes through the integrator exactly once, however the library batches them into
# launches (the adaptive batching makes launches of different sizes): per-frame value = sum over all launches / frames
frames = int(os.environ["TRAFFIC_TOTAL_FRAMES"])
m = {k: sum(v) / frames for k, v in acc.items()}
px = 1920 * 1080
res = {"tag": tag, "args": args, "per_frame": m}
if "TCC_EA0_RDREQ_sum" in m:
    rd = 32 * m["TCC_EA0_RDREQ_32B_sum"] + 64 * m["TCC_EA0_RDREQ_64B_sum"] + 128 * m["TCC_EA0_RDREQ_128B_sum"]
    other = m["TCC_EA0_RDREQ_sum"] - m["TCC_EA0_RDREQ_32B_sum"] - m["TCC_EA0_RDREQ_64B_sum"] - m["TCC_EA0_RDREQ_128B_sum"]
    res["read_bytes"] = rd; res["read_requests_other_size"] = other
if "TCC_EA0_WRREQ_sum" in m:
    res["write_bytes"] = 64 * m["TCC_EA0_WRREQ_64B_sum"] + 32 * (m["TCC_EA0_WRREQ_sum"] - m["TCC_EA0_WRREQ_64B_sum"])
if "read_bytes" in res and "write_bytes" in res:
    res["total_bytes"] = res["read_bytes"] + res["write_bytes"]
    res["ratio_to_algorithmic"] = res["total_bytes"] / (32.0 * px)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps({k: (round(v) if isinstance(v, float) else v) for k, v in res.items() if k != "per_frame"}), {k: round(v) for k, v in m.items()})
PY
