#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_9
timeout 300 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so 1200 304 --multisample --tune batch_pass_min_tiles=0 2>&1 | tail -30 | tee gpurun_out/r4_9/stress.log
for i in 1 2 3; do timeout 100 python bench.py --no-cpu-baseline --steady-ms 0 --config C3 --spp 4 --steps 240 --warmup 80 2>&1 | tail -3 | cut -c1-300; echo "rc=$?"; done | tee gpurun_out/r4_9/c3spp4.log
