#!/bin/bash
# 1/8 share of 1080p under tuning knobs (tools/emulate_strong.py, EMULATE_ONLY=1920x1080:8, EMULATE_TUNE="key=value,...")
R=/root/repo; cd $R
for t in "" "parked_max=16" "parked_max=32" "batch_wg=5" "batch_wg=4" "queue_chunk=2" "queue_chunk=8" "carry_last=0" "tile_masks=0"; do
  echo -n "[$t] "; EMULATE_TUNE="$t" EMULATE_ONLY=1920x1080:8 timeout 200 python tools/emulate_strong.py gpurun_out/emu_tmp.json 2>&1 | grep world
done 2>&1 | tee gpurun_out/share_knobs.log
for sc in default stress; do MI355PT_LIB=$R/tools/ab/libQ.so timeout 200 python tools/profile_sections.py $sc 0 640 2>&1 | grep -v amdgpu; done 2>&1 | tee gpurun_out/sections_q.log
for L in P Q; do PROFILE_SHARE=3,8 MI355PT_LIB=$R/tools/ab/lib$L.so timeout 200 python tools/profile_sections.py default 0 2560 2>&1 | grep -v amdgpu; done 2>&1 | tee -a gpurun_out/sections_q.log
