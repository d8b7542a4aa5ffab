#!/bin/bash
# quick SQ counter pass for one bench variant (counters summed over all integrator launches / the 64 frames rendered):
# tools/pmc_quick.sh <tag> <bench args...>
export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/pmcq_$1
mkdir -p $OUT
cd /tmp
BENCH="python $R/bench.py --steps 59 --warmup 5 --clock-warmup-ms 0 --steady-ms 0 --no-cpu-baseline ${@:2}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -o sq -- $BENCH > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_BRANCH --output-format csv -d $OUT/sq2 -o sq2 -- $BENCH > $OUT/sq2.log 2>&1
python - <<PY
import csv, collections
for f in ("$OUT/sq/sq_counter_collection.csv", "$OUT/sq2/sq2_counter_collection.csv"):
    acc = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(f)):
        if "pt_integrate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(f"$1 {k:24s} {sum(v)/64:16.0f} per frame (64 frames in {len(v)} launches)")
PY
