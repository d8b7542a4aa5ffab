#!/bin/bash
# round 6, final sources: stress and fuzz on seeds no earlier run of the round used (profiles/r06/fresh_seeds.log)
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/r06
{ echo "== frame-fed launches on every image size (--tune feed_min_tiles=0), product / audit+chaos builds"
  for L in "" _audit_chaos; do timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so 4000 $((920 + ${#L})) --tune feed_min_tiles=0 | grep -v "^\.\.\."; done
  echo "== ... with the fused display on (--tune feed_display=1) and an idle budget of 1 us"
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit.so 3000 931 --tune feed_min_tiles=0 --tune feed_display=1 --tune feed_idle_us=1 | grep -v "^\.\.\."
  echo "== classic, all four builds"
  for L in "" _audit _chaos _audit_chaos; do timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so 3000 $((940 + ${#L})) | grep -v "^\.\.\."; done
  echo "== multisample (packed counters, early pass)"
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit_chaos.so 3000 951 --multisample --tune batch_pass_min_tiles=0 | grep -v "^\.\.\."
  echo "== fuzz general / pipelining / group_spp under audit / grid"
  timeout 600 python tools/fuzz_parity.py 400 401; FUZZ_FOCUS=pipelining timeout 600 python tools/fuzz_parity.py 300 402
  MI355PT_LIB=$R/opentk-pathtracer_amd/libmi355pt_audit.so FUZZ_FOCUS=pipelining FUZZ_BIAS=group_spp timeout 600 python tools/fuzz_parity.py 300 403
  FUZZ_FOCUS=grid timeout 600 python tools/fuzz_parity.py 600 404; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/fresh_seeds.log
grep "handover_stress:\|==\|cases," gpurun_out/r06/fresh_seeds.log
