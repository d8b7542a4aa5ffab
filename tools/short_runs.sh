#!/bin/bash
# The driver's bench command is short (--steps 20 --warmup 5): total wall time and HIP-event time of K timed steps, for a
# few K, to separate the fixed cost of a timed region (ramp, drain, launch + sync latency) from the per-frame cost.
R=/root/repo
for K in ${KS:-5 10 20 40 80 160}; do
  for rep in 1 2 3; do
    python $R/bench.py --no-cpu-baseline --steady-ms 0 --steps $K --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['steps']
print(f\"K={k:4d} value {d['value']:9.1f}  wall {d['ms_per_step']*k*1e3:8.1f} us  events {d['roofline']['kernel_ms']*k*1e3:8.1f} us\")"
  done
done
