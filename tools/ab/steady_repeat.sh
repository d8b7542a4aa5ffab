#!/bin/bash
# tools/ab/steady_repeat.sh <runs> "<lib> ..." <bench args...>: timed + steady rate and the hand-over bound's counters per run
R=/root/repo; cd $R
N=$1; LIBS=$2; shift; shift
for L in $LIBS; do for i in $(seq 1 $N); do
  MI355PT_LIB=$R/$L python bench.py --steps 256 --warmup 64 --no-cpu-baseline --steady-ms 400 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', '%.0f steady %.0f' % (d['value'], d['steady']['value']), d.get('handover_bound'))"
done; done
