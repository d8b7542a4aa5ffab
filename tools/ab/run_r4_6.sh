#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_7
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
MI355PT_LIB=$R/tools/ab/libL82.so timeout 900 python -m pytest tests/test_gpu_sphere_grid.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -3 | tee gpurun_out/r4_7/parity_K4.log
for rep in 1 2; do
for L in A K6 K8 K12 L62 L83 L82 L122; do
  for args in "--config C3" "--config C3 --spp 4 --steps 240 --warmup 80"; do
    echo -n "lib$L [$args] "; MI355PT_LIB=$R/tools/ab/lib$L.so python bench.py --no-cpu-baseline --steady-ms 0 $args 2>gpurun_out/r4_7/err_$L.log | val
  done
done
done 2>&1 | tee gpurun_out/r4_7/ab.log
