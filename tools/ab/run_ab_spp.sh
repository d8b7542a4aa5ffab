#!/bin/bash
# A/B of the spp > 1 kernels: tools/ab/libA.so vs libB.so
R=/root/repo
for rep in 1 2; do
for L in A B; do
  for args in "--spp 4 --steps 240 --warmup 80" "--spp 2 --steps 480 --warmup 160" "--config C3 --spp 4 --steps 240 --warmup 80"; do
    v=$(MI355PT_LIB=$R/tools/ab/lib$L.so python $R/bench.py --no-cpu-baseline --steady-ms 0 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "lib$L [$args] $v"
  done
done
done
