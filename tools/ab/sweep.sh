#!/bin/bash
R=/root/repo
b() { python $R/bench.py --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for v in 0 34 44 33; do echo "variant $v: $(b --variant $v)"; done
for q in 4 6 12; do echo "chunk $q: $(PT_QUEUE_CHUNK=$q b)"; done
