#!/bin/bash
# full-size soak on the audit builds + long oracle-checked fuzz sessions on the final sources -> gpurun_out/r04/audit_soak.log, fuzz_long.log
R=/root/repo; cd $R; mkdir -p gpurun_out/r04
A=$R/opentk-pathtracer_amd/libmi355pt_audit.so; AC=$R/opentk-pathtracer_amd/libmi355pt_audit_chaos.so
{ MI355PT_LIB=$A timeout 400 python tools/soak.py 3072 1920 1080 default 8 1
  MI355PT_LIB=$A timeout 400 python tools/soak.py 1536 1920 1080 stress256 8 1
  MI355PT_LIB=$A timeout 400 python tools/soak.py 1024 1920 1080 glass 32 1
  MI355PT_LIB=$A timeout 400 python tools/soak.py 512 1920 1080 default 8 4
  MI355PT_LIB=$A timeout 400 python tools/soak.py 1536 1920 1080 default 8 1 8
  MI355PT_LIB=$AC timeout 400 python tools/soak.py 768 1920 1080 default 8 1; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/audit_soak.log
{ echo "== FUZZ_FOCUS=pipelining FUZZ_BIAS=group_spp, audit build"; MI355PT_LIB=$A FUZZ_FOCUS=pipelining FUZZ_BIAS=group_spp timeout 500 python tools/fuzz_parity.py 4000 401
  echo "== FUZZ_FOCUS=pipelining, product library"; FUZZ_FOCUS=pipelining timeout 400 python tools/fuzz_parity.py 3000 402
  echo "== FUZZ_FOCUS=grid"; FUZZ_FOCUS=grid timeout 400 python tools/fuzz_parity.py 3000 403
  echo "== general"; timeout 400 python tools/fuzz_parity.py 2000 404; } 2>&1 | grep -v amdgpu.ids | grep "cases\|==" > gpurun_out/r04/fuzz_long.log
cat gpurun_out/r04/audit_soak.log gpurun_out/r04/fuzz_long.log
