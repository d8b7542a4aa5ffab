#!/bin/bash
# tools/ab/spp4_tunes.sh "<bench args>" "<tune set>" ...: steady rate per set of --tune flags (two rounds), tree library
R=/root/repo; cd $R
ARGS=$1; shift
for rep in 1 2; do for t in "$@"; do
  v=$(python bench.py --steps 256 --warmup 64 --no-cpu-baseline --steady-ms 400 $ARGS $t 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f steady %.0f' % (d['value'], d['steady']['value']))")
  echo "[$ARGS] [$t] $v"
done; done
