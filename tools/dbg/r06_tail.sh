#!/bin/bash
# the part of `tools/round_profiles.sh run r06` that the 3,600 s limit of its GPU call cut off: the rest of the zero-budget stress and the
# atmosphere profile (appended to the same logs)
R=$(pwd); RND=r06; export TMPDIR=/tmp; mkdir -p gpurun_out/$RND
N=${STRESS_CASES:-12000}; NM=${STRESS_MS_CASES:-5000}
{ for L in _chaos _audit_chaos; do echo "== libmi355pt$L.so --tune handover_budget_ms=0"; timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so $((N / 2)) $((800 + ${#L})) --tune handover_budget_ms=0 | grep -v "^\.\.\."; done
  echo "== multisample, zero budget, batch-pass kernel forced onto tiny images"
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit_chaos.so $NM 811 --multisample --tune batch_pass_min_tiles=0 --tune handover_budget_ms=0 | grep -v "^\.\.\."
  echo "== multisample, zero budget, in-lane sample chain"
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so $NM 812 --multisample --tune handover_budget_ms=0 | grep -v "^\.\.\."
  echo "== fresh handle per case, 1 ms budget"
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so 1000 813 --fresh --tune handover_budget_ms=1 --tune handover_check_us=50 | grep -v "^\.\.\."; } > gpurun_out/$RND/handover_zero_budget_tail.log 2>&1
{ for sz in 256 1024 2048; do python tools/atmo_profile.py $sz $((sz > 1000 ? 3 : 16)); done
  ( cd /tmp; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/prof_${RND}_atmo -o pmc -- python $R/tools/atmo_profile.py 1024 2 > /dev/null 2>&1 )
  grep atmo $R/gpurun_out/prof_${RND}_atmo/pmc_counter_collection.csv > gpurun_out/$RND/atmosphere_pmc_1024.csv; } 2>&1 | grep -v amdgpu.ids > gpurun_out/$RND/atmosphere_profile.log
grep "handover_stress:\|==\|hand-over bound" gpurun_out/$RND/handover_zero_budget_tail.log; cat gpurun_out/$RND/atmosphere_profile.log
