#!/bin/bash
# Profiling recipe of a round (run on the GPU box via gpurun):  bash tools/profile_round.sh <tag> [bench args]
#   1. rocprofv3 --kernel-trace --stats            (per-kernel time; must agree with bench.py's HIP-event kernel_ms)
#   2. separate --pmc passes, each with --kernel-trace only (task rules): FETCH_SIZE | WRITE_SIZE | the L2's memory-side
#      request counters by request size (exact bytes: 32 / 64 / 128-byte reads, 32 / 64-byte writes) | SQ instruction mix |
#      SQ wait / busy
#   3. the FETCH / WRITE passes again on a CALIBRATION run (--depth 0: the kernel only reads + writes the accumulation
#      image, a known 16 B + 16 B per pixel) to calibrate the gfx950 FETCH_SIZE under-count on OUR access pattern, as
#      /opt/skills/guides/MI355X_MICROARCH.md prescribes
# The library batches frames into launches adaptively, so counters are SUMMED over all integrator launches of a run and
# divided by the frames the run renders (warm-up + timed steps, no clock warm-up).
# Raw output -> gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns it into the files committed under profiles/.
export TMPDIR=/tmp
R=/root/repo
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; print(g.load_package().native.csrc_hash())" > $OUT/csrc_hash.txt
cd /tmp
STEPS=${PROFILE_STEPS:-640}
WARM=${PROFILE_WARMUP:-320}
echo $((STEPS + WARM)) > $OUT/frames.txt
BENCH="python $R/bench.py --steps $STEPS --warmup $WARM --clock-warmup-ms 0 --steady-ms 0 --no-cpu-baseline ${@:2}"
CAL="python $R/bench.py --steps $STEPS --warmup $WARM --clock-warmup-ms 0 --steady-ms 0 --no-cpu-baseline --depth 0 ${@:2}"
# PROFILE_PASSES="stats fetch write sq" restricts the passes (a reduced re-profile after a host-only change); default: all
run() { name=$1; opts=$2; cmd=$3; if [ -n "$PROFILE_PASSES" ] && [[ " $PROFILE_PASSES " != *" $name "* ]]; then return; fi
        rocprofv3 --kernel-trace $opts --output-format csv -d $OUT/$name -o $name -- $cmd > $OUT/$name.log 2>&1; }
run stats "--stats" "$BENCH"
run fetch "--pmc FETCH_SIZE" "$BENCH"
run write "--pmc WRITE_SIZE" "$BENCH"
run rd "--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "$BENCH"
run wr "--pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" "$BENCH"
run sq "--pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "$BENCH"
run sq2 "--pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_BRANCH" "$BENCH"
if [ -z "$PROFILE_NO_CAL" ]; then
  run cal_fetch "--pmc FETCH_SIZE" "$CAL"
  run cal_write "--pmc WRITE_SIZE" "$CAL"
fi
python $R/bench.py $PROFILE_BENCH_EXTRA ${@:2} > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json
# only the small files travel back (the raw traces are large): per-kernel stats + counter tables, and the integrator's rows of
# the stats run's kernel trace (start / end of every launch: consecutive launches OVERLAP on two streams, see summarize_profile.py)
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/stats/**/*kernel_trace.csv", recursive=True) + glob.glob("$OUT/stats/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "pt_integrate" in r.get("Kernel_Name", ""):
            rows.append({k: r[k] for k in ("Kernel_Name", "Queue_Id", "Start_Timestamp", "End_Timestamp") if k in r})
    break
if rows:
    w = csv.DictWriter(open("$OUT/stats/integrator_launches.csv", "w", newline=""), fieldnames=list(rows[0]))
    w.writeheader(); w.writerows(rows)
PY
find $OUT -name "*_kernel_trace.csv" -delete
ls $OUT
