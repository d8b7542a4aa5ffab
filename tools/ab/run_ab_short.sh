#!/bin/bash
# A/B of the short timed region the driver uses (--steps 20 --warmup 5): tools/ab/libA.so vs the tree's library
R=/root/repo
for rep in 1 2 3; do
  for L in A tree; do
    if [ $L = tree ]; then unset MI355PT_LIB; else export MI355PT_LIB=$R/tools/ab/lib$L.so; fi
    v=$(python $R/bench.py --no-cpu-baseline --steady-ms 0 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$L K=20: $v"
  done
done
