#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_15
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for t in "" "--tune carry_last=0"; do
    for args in "" "--config C5" "--frame-batch 1" "--strong-4k" "--env sky2048"; do echo -n "[$t] [$args] "; timeout 120 python bench.py --no-cpu-baseline --steady-ms 0 $args $t 2>/dev/null | val; done
  done
done | tee gpurun_out/r4_15/ab.log
bash tools/traffic_quick.sh tree 2>&1 | tail -1 | cut -c1-250
bash tools/traffic_quick.sh tree_C5 --config C5 2>&1 | tail -1 | cut -c1-250
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -4
timeout 200 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so 3000 411 2>&1 | tail -1 | cut -c1-250
timeout 200 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit_chaos.so 1500 412 2>&1 | tail -1 | cut -c1-250
