#!/bin/bash
# 1/8 share of 1080p under tuning knobs (tools/emulate_strong.py, EMULATE_ONLY=1920x1080:8, EMULATE_TUNE="key=value,...")
R=/root/repo; cd $R
[ -n "$1" ] && export MI355PT_LIB=$R/tools/ab/lib$1.so
shift
for rep in 1 2; do
for t in "$@"; do
  echo -n "[$t] "; EMULATE_TUNE="$t" EMULATE_ONLY=${SHARE:-1920x1080:8} timeout 200 python tools/emulate_strong.py gpurun_out/emu_tmp.json 2>&1 | grep world
done
done 2>&1 | tee gpurun_out/share_knobs.log
