#!/bin/bash
# tools/ab/run_ab3.sh "A B ..." : bench A/B of library builds on one box (no tests)
R=/root/repo
cd $R
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
LIBS=${1:-"A B"}
for rep in 1 2; do
for L in $LIBS; do
  for args in "" "--config C3" "--config C5" "--spp 4 --steps 240 --warmup 80" "--frame-batch 1" "--steps 20 --warmup 5" "--strong-4k"; do
    echo -n "lib$L [$args] "; MI355PT_LIB=$R/tools/ab/lib$L.so timeout 120 python bench.py --no-cpu-baseline --steady-ms 0 $args 2>/dev/null | val
  done
done
done 2>&1 | tee gpurun_out/ab3.log
