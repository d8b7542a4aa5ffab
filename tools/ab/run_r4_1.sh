#!/bin/bash
# round 4, call 1: parity of the sphere-run loop (libB) + A/B against HEAD (libA) and the carry-last build (libC)
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_1
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
MI355PT_LIB=$R/tools/ab/libB.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sphere_grid.py -m gpu -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -4 | tee gpurun_out/r4_1/parity_B.log
for rep in 1 2; do
for L in A B C; do
  for args in "" "--config C3" "--config C5" "--frame-batch 1"; do
    echo -n "lib$L [$args] "; MI355PT_LIB=$R/tools/ab/lib$L.so python bench.py --no-cpu-baseline --steady-ms 0 $args | val
  done
done
done 2>&1 | tee gpurun_out/r4_1/ab.log
for L in A B; do MI355PT_LIB=$R/tools/ab/lib$L.so bash tools/pmc_quick.sh r4_1_$L 2>&1 | grep "INSTS_VALU\|INSTS_SALU\|INSTS_LDS\|WAVE_CYCLES\|BUSY_CYCLES\|ACTIVE_INST_VALU"; done | tee gpurun_out/r4_1/pmc.log
