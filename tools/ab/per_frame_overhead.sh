#!/bin/bash
# what one launch per Render() costs against pipelined launches, per workload / tuning: tools/ab/per_frame_overhead.sh "<bench args>" ...
cd /root/repo
for extra in "$@"; do
python bench.py --steps 256 --warmup 64 --no-cpu-baseline --steady-ms 50 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$extra] per_frame_launch %.4f ms  displayed %.4f ms  steady %.4f  diff %.4f' % (d['per_frame_launch']['ms_per_step'], d['displayed_frame']['ms_per_displayed_frame'], d['steady']['ms_per_step'], d['per_frame_launch']['ms_per_step']-d['steady']['ms_per_step']))"
done
