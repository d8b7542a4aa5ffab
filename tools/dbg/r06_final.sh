#!/bin/bash
# round 6, after the ticket-accounting fix of the frame-fed launches (csrc changed: the profiles of the default workload and of the per-frame
# workload are taken again — the other kernels' code is unchanged —, and the whole stress / fuzz / GPU-suite evidence is run again)
R=$(pwd); RND=r06; export TMPDIR=/tmp; mkdir -p gpurun_out/$RND
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/$RND/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$RND/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/$RND/smoke.log 2>&1
prof() { local tag=${RND}_$1 key=$2; shift 2
  bash tools/profile_round.sh $tag "$@" > gpurun_out/$RND/${tag}_profile.log 2>&1
  python tools/summarize_profile.py $tag $RND $key ${RND}_default > gpurun_out/$RND/${tag}_summary.log 2>&1
  python bench.py $PROFILE_BENCH_EXTRA "$@" > gpurun_out/$RND/${tag}_bench.json 2> gpurun_out/$RND/${tag}_bench.err; }
prof default default_1920x1080_d8_spp1_atmosphere256_g1
export PROFILE_NO_CAL=1 PROFILE_BENCH_EXTRA="--no-cpu-baseline"
prof perframe default_1920x1080_d8_spp1_atmosphere256_g1_fb1 --frame-batch 1
unset PROFILE_NO_CAL PROFILE_BENCH_EXTRA
python bench.py --steps 20 --warmup 5 > gpurun_out/$RND/driver_command_bench.json 2> gpurun_out/$RND/driver_command_bench.err
bash tools/unchained_timed.sh $RND > gpurun_out/$RND/unchained_timed.log 2>&1
cp gpurun_out/$RND/${RND}_default_bench.json gpurun_out/$RND/bench_default.json 2>/dev/null
N=8000; NM=3000
{ for L in "" _audit _chaos _audit_chaos; do echo "== libmi355pt$L.so"; timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so $N $((700 + ${#L})) | grep -v "^\.\.\."; done
  echo "== round 6: frame-fed launches on every image size (--tune feed_min_tiles=0), idle budget 150 us and 1 us, zero hand-over budget, fused display"
  for L in "" _audit _audit_chaos; do timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so 6000 $((720 + ${#L})) --tune feed_min_tiles=0 | grep -v "^\.\.\."; done
  timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit.so 3000 731 --tune feed_min_tiles=0 --tune feed_idle_us=1 | grep -v "^\.\.\."
  timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit_chaos.so 3000 733 --tune feed_min_tiles=0 --tune feed_idle_us=1 | grep -v "^\.\.\."
  timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so 3000 732 --tune feed_min_tiles=0 --tune handover_budget_ms=0 | grep -v "^\.\.\."
  timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit.so 3000 931 --tune feed_min_tiles=0 --tune feed_display=1 --tune feed_idle_us=1 | grep -v "^\.\.\."
  timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so 4000 920 --tune feed_min_tiles=0 | grep -v "^\.\.\."
  echo "== multisample focus, batch-pass kernel forced onto tiny images (--tune batch_pass_min_tiles=0)"
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit_chaos.so $NM 711 --multisample --tune batch_pass_min_tiles=0 | grep -v "^\.\.\."
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so $NM 712 --multisample --tune batch_pass_min_tiles=0 | grep -v "^\.\.\."; } > gpurun_out/$RND/handover_stress.log 2>&1
{ echo "== general"; timeout 600 python tools/fuzz_parity.py 600 301; echo "== FUZZ_FOCUS=pipelining"; FUZZ_FOCUS=pipelining timeout 600 python tools/fuzz_parity.py 400 302;
  echo "== FUZZ_FOCUS=pipelining FUZZ_BIAS=group_spp under the audit build"; MI355PT_LIB=$R/opentk-pathtracer_amd/libmi355pt_audit.so FUZZ_FOCUS=pipelining FUZZ_BIAS=group_spp timeout 600 python tools/fuzz_parity.py 400 303;
  echo "== FUZZ_FOCUS=grid"; FUZZ_FOCUS=grid timeout 600 python tools/fuzz_parity.py 1000 304; } 2>&1 | grep -v amdgpu.ids > gpurun_out/$RND/fuzz.log
{ for L in "" _audit_chaos; do echo "== libmi355pt$L.so --tune handover_budget_ms=0"; timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so 4000 $((800 + ${#L})) --tune handover_budget_ms=0 | grep -v "^\.\.\."; done
  echo "== multisample, zero budget, batch-pass kernel forced onto tiny images"
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit_chaos.so 2000 811 --multisample --tune batch_pass_min_tiles=0 --tune handover_budget_ms=0 | grep -v "^\.\.\."; } > gpurun_out/$RND/handover_zero_budget_final.log 2>&1
tail -3 gpurun_out/$RND/pytest_gpu.log; grep -c "library error\|MISMATCH" gpurun_out/$RND/handover_stress.log gpurun_out/$RND/handover_zero_budget_final.log; grep "handover_stress:\|==" gpurun_out/$RND/handover_stress.log gpurun_out/$RND/handover_zero_budget_final.log | cut -c1-220; grep "cases,\|==" gpurun_out/$RND/fuzz.log; cat gpurun_out/$RND/unchained_timed.log | tail -3
