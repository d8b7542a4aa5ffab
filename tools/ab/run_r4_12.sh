#!/bin/bash
cd /root/repo
echo "== multisample forced"; timeout 120 tools/handover_stress.bin tools/ab/libB.so 1200 304 --multisample --tune batch_pass_min_tiles=0 2>&1 | tail -2 | cut -c1-300
echo "== multisample forced, other seed"; timeout 120 tools/handover_stress.bin tools/ab/libB.so 1500 999 --multisample --tune batch_pass_min_tiles=0 2>&1 | tail -2 | cut -c1-300
echo "== multisample default"; timeout 120 tools/handover_stress.bin tools/ab/libB.so 1200 305 --multisample 2>&1 | tail -2 | cut -c1-300
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do echo -n "C3 spp4: "; MI355PT_LIB=tools/ab/libB.so timeout 100 python bench.py --no-cpu-baseline --steady-ms 0 --config C3 --spp 4 --steps 240 --warmup 80 2>/dev/null | val; done
