#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_2
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
MI355PT_LIB=$R/tools/ab/libB.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -2 | tee gpurun_out/r4_2/parity_B.log
for rep in 1 2; do
for L in A B; do
  for args in "" "--config C5" "--frame-batch 1" "--spp 4 --steps 240 --warmup 80"; do
    echo -n "lib$L [$args] "; MI355PT_LIB=$R/tools/ab/lib$L.so python bench.py --no-cpu-baseline --steady-ms 0 $args 2>/dev/null | val
  done
done
done 2>&1 | tee gpurun_out/r4_2/ab.log
for L in B; do MI355PT_LIB=$R/tools/ab/lib$L.so bash tools/pmc_quick.sh r4_2_$L 2>&1 | grep "INSTS_VALU\|INSTS_SALU\|INSTS_LDS\|WAVE_CYCLES\|BUSY_CYCLES\|ACTIVE_INST_VALU"; done | tee gpurun_out/r4_2/pmc.log
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"; lscpu | grep -i "model name\|socket\|thread\|core" 
