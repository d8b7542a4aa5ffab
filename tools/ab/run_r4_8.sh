#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_8
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for L in A B; do
  for args in "" "--config C3" "--config C5" "--spp 4 --steps 240 --warmup 80" "--config C3 --spp 4 --steps 240 --warmup 80" "--frame-batch 1"; do
    echo -n "lib$L [$args] "; MI355PT_LIB=$R/tools/ab/lib$L.so timeout 120 python bench.py --no-cpu-baseline --steady-ms 0 $args 2>>gpurun_out/r4_8/err_$L.log | val
  done
done
done 2>&1 | tee gpurun_out/r4_8/ab.log
MI355PT_LIB=$R/tools/ab/libB.so timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -5 | tee gpurun_out/r4_8/pytest_B.log
