#!/bin/bash
# tools/ab/spp4_libs.sh "<lib> ..." "<bench args>" ...: steady rate at spp > 1 per library (two rounds)
R=/root/repo; cd $R
LIBS=$1; shift
for rep in 1 2; do for L in $LIBS; do for args in "$@"; do
  v=$(MI355PT_LIB=$R/$L python bench.py --steps 256 --warmup 64 --no-cpu-baseline --steady-ms 400 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f steady %.0f' % (d['value'], d['steady']['value']))")
  echo "$L [$args] $v"
done; done; done
