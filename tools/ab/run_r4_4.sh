#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_4
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
MI355PT_LIB=$R/tools/ab/libB.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -2 | tee gpurun_out/r4_4/parity_B.log
for rep in 1 2; do
for L in A B; do
  for args in "" "--config C5" "--config C3"; do
    echo -n "lib$L [$args] "; MI355PT_LIB=$R/tools/ab/lib$L.so python bench.py --no-cpu-baseline --steady-ms 0 $args 2>/dev/null | val
  done
done
done 2>&1 | tee gpurun_out/r4_4/ab.log
for L in B; do MI355PT_LIB=$R/tools/ab/lib$L.so bash tools/pmc_quick.sh r4_4_$L 2>&1 | grep "INSTS_VALU\|INSTS_SALU\|INSTS_LDS\|INSTS_BRANCH\|WAVE_CYCLES"; done | tee gpurun_out/r4_4/pmc.log
