#!/bin/bash
cd /root/repo
for extra in "$@"; do
python bench.py --config C3 --steps 512 --warmup 128 --no-cpu-baseline --steady-ms 300 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$extra] value %.0f  steady %.0f Msamples/s (%.4f ms)' % (d['value'], d['steady']['value'], d['steady']['ms_per_step']))"
done
