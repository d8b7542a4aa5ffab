#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_14
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for L in S C; do
  for t in "" "--tune xcd_affine=1"; do
    for args in "" "--config C3"; do echo -n "lib$L [$t] [$args] "; MI355PT_LIB=$R/tools/ab/lib$L.so timeout 120 python bench.py --no-cpu-baseline --steady-ms 0 $args $t 2>/dev/null | val; done
  done
  done
done | tee gpurun_out/r4_14/ab.log
MI355PT_LIB=$R/tools/ab/libS.so bash tools/traffic_quick.sh affine_sc1 --tune xcd_affine=1 2>&1 | tail -1 | cut -c1-250
MI355PT_LIB=$R/tools/ab/libC.so bash tools/traffic_quick.sh carry 2>&1 | tail -1 | cut -c1-250
MI355PT_LIB=$R/tools/ab/libC.so bash tools/traffic_quick.sh carry_affine --tune xcd_affine=1 2>&1 | tail -1 | cut -c1-250
