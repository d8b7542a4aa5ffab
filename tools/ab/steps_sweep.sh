#!/bin/bash
# tools/ab/steps_sweep.sh "<steps> ..." [bench args]: the timed region of one flush of n frames — kernel time (HIP events) and host time, total = ms_per_step x n
R=/root/repo; cd $R
STEPS=$1; shift
for rep in 1 2; do for n in $STEPS; do
  python bench.py --steps $n --warmup 5 --no-cpu-baseline --steady-ms 50 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=$n
r=d['roofline']
print('steps %4d: host %.4f ms/step (total %.3f ms)  kernel %.4f ms/step (total %.3f ms)  launches %s  frames/launch %s' % (n, d['ms_per_step'], d['ms_per_step']*n, r['kernel_ms'], r['kernel_ms']*n, r.get('launches_in_timed_region'), r.get('frames_per_launch')))"
done; done
