#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun): kernel-trace stats + separate PMC passes.
# Output lands in gpurun_out/ (scratch); the summaries worth judging are copied into profiles/ by hand.
set -x
export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/prof_${1:-r01}
mkdir -p $OUT
cd /tmp
BENCH="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline ${@:2}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -o sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
find $OUT -name "*.csv" | head -50
