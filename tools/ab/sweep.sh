#!/bin/bash
R=/root/repo
python $R/tools/gpu_probe.py 194 2>&1 | tail -9
b() { python $R/bench.py --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
echo "variant 0: $(b --variant 0)"
for q in 4 8 16; do for v in 174 194 193; do echo "chunk $q variant $v: $(PT_QUEUE_CHUNK=$q b --variant $v)"; done; done
