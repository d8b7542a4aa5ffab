#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_10
echo "== no slices"; timeout 200 tools/handover_stress.bin tools/ab/libN.so 1200 304 --multisample --tune batch_pass_min_tiles=0 2>&1 | tail -2 | cut -c1-300
echo "== slices, multisample without forcing the batch pass"; timeout 200 tools/handover_stress.bin tools/ab/libB.so 1200 304 --multisample 2>&1 | tail -2 | cut -c1-300
echo "== slices, spp 1 stress"; timeout 200 tools/handover_stress.bin tools/ab/libB.so 1500 301 2>&1 | tail -2 | cut -c1-300
echo "== slices, fuzz grid"; MI355PT_LIB=$R/tools/ab/libB.so FUZZ_FOCUS=grid timeout 400 python tools/fuzz_parity.py 400 11 2>&1 | tail -4 | cut -c1-300
