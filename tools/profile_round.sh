#!/bin/bash
# Profiling recipe of a round (run on the GPU box via gpurun):  bash tools/profile_round.sh <tag> [bench args]
#   1. rocprofv3 --kernel-trace --stats            (per-kernel time; must agree with bench.py's HIP-event kernel_ms)
#   2. separate --pmc passes: FETCH_SIZE | WRITE_SIZE | SQ instruction mix | SQ wait/busy   (never combined with traces
#      other than --kernel-trace, see the task's gpurun rules)
#   3. the same FETCH/WRITE passes on a CALIBRATION run (--depth 0: the kernel only reads + writes the accumulation
#      image, a known 16 B + 16 B per pixel) to calibrate the gfx950 FETCH_SIZE under-count on OUR access pattern
# Raw output -> gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns it into the files committed under profiles/.
export TMPDIR=/tmp
R=/root/repo
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; print(g.load_package().native.csrc_hash())" > $OUT/csrc_hash.txt
cd /tmp
BENCH="python $R/bench.py --steps 640 --warmup 320 --no-cpu-baseline ${@:2}"   # multiples of the 64-frame batch: every launch renders 32 frames
CAL="python $R/bench.py --steps 640 --warmup 320 --no-cpu-baseline --depth 0 ${@:2}"
run() { name=$1; opts=$2; cmd=$3; rocprofv3 --kernel-trace $opts --output-format csv -d $OUT/$name -o $name -- $cmd > $OUT/$name.log 2>&1; }
run stats "--stats" "$BENCH"
run fetch "--pmc FETCH_SIZE" "$BENCH"
run write "--pmc WRITE_SIZE" "$BENCH"
run sq "--pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "$BENCH"
run sq2 "--pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_BRANCH" "$BENCH"
run tcc "--pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "$BENCH"
run cal_fetch "--pmc FETCH_SIZE" "$CAL"
run cal_write "--pmc WRITE_SIZE" "$CAL"
python $R/bench.py ${@:2} > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
ls $OUT
