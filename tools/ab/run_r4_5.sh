#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_5
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for L in A B R E; do
  for args in "" "--config C5" "--config C3" "--spp 4 --steps 240 --warmup 80"; do
    echo -n "lib$L [$args] "; MI355PT_LIB=$R/tools/ab/lib$L.so python bench.py --no-cpu-baseline --steady-ms 0 $args 2>/dev/null | val
  done
done
done 2>&1 | tee gpurun_out/r4_5/ab.log
