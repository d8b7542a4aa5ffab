#!/bin/bash
# the GPU suite + smoke on the tree's product library (log under gpurun_out/suite/)
cd /root/repo
mkdir -p gpurun_out/suite
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -40 > gpurun_out/suite/pytest_gpu.log; tail -6 gpurun_out/suite/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2
