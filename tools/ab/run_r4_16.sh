#!/bin/bash
cd /root/repo
time (timeout 600 python bench.py --gpus 8 --share-gpu --steps 64 --warmup 64 --steady-ms 0 --no-cpu-baseline > gpurun_out/b8.json 2> gpurun_out/b8.err); echo rc=$?
tail -c 1500 gpurun_out/b8.err; python - <<'PY'
import json
l=[x for x in open('/root/repo/gpurun_out/b8.json') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print({k:d[k] for k in ('value','ranks','n_gpus','ms_per_step','checks')}); print(d.get('configs3_4k',{}).get('checks'), d.get('in_process_group'))
PY
