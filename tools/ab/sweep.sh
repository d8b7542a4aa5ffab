#!/bin/bash
R=/root/repo
b() { python $R/bench.py --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for fb in 16 24 32; do echo "batch $fb: $(b --frame-batch $fb --steps 384)"; done
