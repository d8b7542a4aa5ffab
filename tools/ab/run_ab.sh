#!/bin/bash
# A/B two builds of the library on the same box: tools/ab/libA.so vs libB.so (bench value, default + stress scene)
R=/root/repo
for rep in 1 2; do
for L in A B; do
  for args in "" "--scene stress256" "--scene glass --depth 32"; do
    v=$(MI355PT_LIB=$R/tools/ab/lib$L.so python $R/bench.py --no-cpu-baseline --steady-ms 0 $args | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "lib$L [$args] $v"
  done
done
done
