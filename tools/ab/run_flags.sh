#!/bin/bash
R=/root/repo
for rep in 1 2; do
for L in F0 F1 F2 F4; do
  v=$(MI355PT_LIB=$R/tools/ab/lib$L.so python $R/bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "lib$L: $v"
done
done
