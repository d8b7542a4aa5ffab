#!/bin/bash
cd /root/repo
for extra in "$@"; do
python bench.py --spp 4 --steps 256 --warmup 64 --no-cpu-baseline --steady-ms 400 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[--spp 4 $extra] value %.0f  steady %.0f Msamples/s (%.4f ms per frame)' % (d['value'], d['steady']['value'], d['steady']['ms_per_step']))"
done
