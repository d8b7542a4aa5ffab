#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_17
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -3
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1
for sc in default stress glass; do MI355PT_LIB=$R/tools/ab/libP.so timeout 200 python tools/profile_sections.py $sc 0 640 2>&1 | grep -v amdgpu; done | tee gpurun_out/r4_17/profile_sections.log
