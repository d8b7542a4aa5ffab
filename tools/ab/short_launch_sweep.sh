#!/bin/bash
# per-frame launches (pt_set_frame_batch(1)) and displayed frames for several workgroup shapes of short launches:
#   tools/ab/short_launch_sweep.sh "waves:wg[:extra tune] ..."   e.g. "4:5 2:10 2:9:carry_last=0"
R=/root/repo; cd $R
for cfg in ${1:-4:5 2:10 2:9 1:20 1:18 1:16}; do
  IFS=: read w g extra <<< "$cfg"
  python bench.py --steps 256 --warmup 64 --no-cpu-baseline --steady-ms 50 --tune short_waves=$w --tune short_wg=$g ${extra:+--tune $extra} --tune log_launch=400 2>/tmp/sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('waves $w wg/CU $g $extra: per_frame_launch %.4f ms  displayed %.4f ms  (pipelined %.4f)' % (d['per_frame_launch']['ms_per_step'], d['displayed_frame']['ms_per_displayed_frame'], d['ms_per_step']))"
  grep "frames 1 " /tmp/sweep.err | tail -1 | cut -c1-200
done
