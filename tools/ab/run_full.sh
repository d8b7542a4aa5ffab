#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/full
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -6 | cut -c1-300
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1
for L in "" _audit_chaos; do timeout 300 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so 4000 $((900 + ${#L})) 2>&1 | tail -1 | cut -c1-250; done
timeout 300 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so 2000 913 --multisample 2>&1 | tail -1 | cut -c1-250
timeout 300 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so 2000 914 --multisample --tune batch_pass_min_tiles=0 2>&1 | tail -1 | cut -c1-250
timeout 300 python tools/fuzz_parity.py 300 51 2>&1 | tail -1 | cut -c1-200
FUZZ_FOCUS=grid timeout 300 python tools/fuzz_parity.py 300 52 2>&1 | tail -1 | cut -c1-200
for args in "" "--config C3" "--config C5" "--spp 4 --steps 240 --warmup 80" "--config C3 --spp 4 --steps 240 --warmup 80" "--frame-batch 1" "--steps 20 --warmup 5" "--strong-4k" "--config C3 --tune log_launch=4"; do echo -n "[$args] "; timeout 120 python bench.py --no-cpu-baseline --steady-ms 0 $args 2>/dev/null | val; done | tee gpurun_out/full/bench.log
