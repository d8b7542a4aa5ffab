#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_3
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run() { # label lib env...
  local label=$1 lib=$2; shift 2
  for args in "" "--config C5" "--config C3"; do
    echo -n "$label [$args] "; env "$@" MI355PT_LIB=$R/tools/ab/lib$lib.so python bench.py --no-cpu-baseline --steady-ms 0 $args 2>/dev/null | val
  done
}
for rep in 1 2; do
run B6 B X=1
run B6lean B PT_FORCE_LEAN_LDS=1
run W7 W7 PT_BATCH_WG=7 PT_FORCE_LEAN_LDS=1
run W7p32 W7 PT_BATCH_WG=7 PT_FORCE_LEAN_LDS=1 PT_PARKED_MAX=32
run W8p24 W8 PT_BATCH_WG=8 PT_FORCE_LEAN_LDS=1 PT_PARKED_MAX=24
run W8p0 W8 PT_BATCH_WG=8 PT_FORCE_LEAN_LDS=1 PT_PARKED_MAX=0
done 2>&1 | tee gpurun_out/r4_3/ab.log
MI355PT_LIB=$R/tools/ab/libW7.so PT_BATCH_WG=7 PT_FORCE_LEAN_LDS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bit_exact or pipelining" 2>&1 | tail -2
