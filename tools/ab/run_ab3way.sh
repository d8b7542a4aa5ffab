#!/bin/bash
# same-box A/B: tools/ab/run_ab3way.sh "<lib> <lib> ..." "<bench args>" ...   (value and steady of bench.py per library, three rounds)
R=/root/repo; cd $R
LIBS=$1; shift
for rep in 1 2 3; do
for L in $LIBS; do
  for args in "$@"; do
    v=$(MI355PT_LIB=$R/$L python bench.py --steps 640 --warmup 320 --no-cpu-baseline --steady-ms 800 --frame-batch 64 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f steady %.0f kernel_ms %.5f' % (d['value'], d['steady']['value'], d['roofline']['kernel_ms']))")
    echo "$L [$args] $v"
  done
done
done
