#!/bin/bash
# same-box A/B of the round-4 library (tools/ab/libR4.so, built from commit 9cf49ed) against the tree's: steady rate of the headline workload
# and of C3 / C5 / 4 spp
R=/root/repo; cd $R
for rep in 1 2 3; do
for L in tools/ab/libR4.so opentk-pathtracer_amd/libmi355pt.so; do
  for args in "" "--config C3" "--config C5" "--spp 4"; do
    v=$(MI355PT_LIB=$R/$L python bench.py --steps 640 --warmup 320 --no-cpu-baseline --steady-ms 800 --frame-batch 64 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f steady %.0f' % (d['value'], d['steady']['value']))")
    echo "$L [$args] $v"
  done
done
done
