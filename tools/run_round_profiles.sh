#!/bin/bash
# Everything a round commits under profiles/<round>/ in ONE gpurun call:  bash tools/run_round_profiles.sh r03
# Per workload: tools/profile_round.sh (rocprofv3 --stats + separate --pmc passes), then tools/summarize_profile.py ON THE BOX (so that
# profiles/traffic.json / valu_insts.json there carry this build's numbers) and bench.py once more -> the committed *_bench.json and
# every line of bench_configs.jsonl carry roofline.traffic / roofline.valu_issue.  tools/collect_round_profiles.sh repeats the summary
# locally from the merged raw counters (only gpurun_out/ travels back).
R=/root/repo
RND=${1:-r03}
cd $R
mkdir -p gpurun_out/$RND
python -m pytest tests -m gpu -q > gpurun_out/$RND/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$RND/pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/$RND/smoke.log 2>&1
prof() { # tag, workload key, calibration tag, bench args...
  local tag=$1 key=$2 cal=$3; shift 3
  bash tools/profile_round.sh $tag "$@" > gpurun_out/$RND/${tag}_profile.log 2>&1
  python tools/summarize_profile.py $tag $RND $key $cal > gpurun_out/$RND/${tag}_summary.log 2>&1
  python bench.py "$@" > gpurun_out/$RND/${tag}_bench.json 2> gpurun_out/$RND/${tag}_bench.err
}
prof ${RND}_default default_1920x1080_d8_spp1_atmosphere256_g1 ${RND}_default
export PROFILE_NO_CAL=1
prof ${RND}_perframe default_1920x1080_d8_spp1_atmosphere256_g1_fb1 ${RND}_default --frame-batch 1
prof ${RND}_C3 stress256_1920x1080_d8_spp1_atmosphere256_g1 ${RND}_default --config C3
prof ${RND}_C5 glass_1920x1080_d32_spp1_atmosphere256_g1 ${RND}_default --config C5
export PROFILE_STEPS=192 PROFILE_WARMUP=64
prof ${RND}_spp4 default_1920x1080_d8_spp4_atmosphere256_g1 ${RND}_default --spp 4
prof ${RND}_tilewave default_1920x1080_d8_spp1_atmosphere256_g1_variant1 ${RND}_default --variant 1
unset PROFILE_NO_CAL PROFILE_STEPS PROFILE_WARMUP
bash tools/bench_configs.sh > gpurun_out/$RND/bench_configs.log 2>&1; cp gpurun_out/bench_configs.jsonl gpurun_out/$RND/
python bench.py --steps 20 --warmup 5 > gpurun_out/$RND/driver_command_bench.json 2> gpurun_out/$RND/driver_command_bench.err
python tools/emulate_strong.py gpurun_out/$RND/emulate_strong.json > gpurun_out/$RND/emulate_strong.log 2>&1
python tools/present_rate.py --json gpurun_out/$RND/present_rate.json > gpurun_out/$RND/present_rate.log 2>&1
python tools/present_rate.py --devices 0,0 --json gpurun_out/$RND/present_rate_group2.json > gpurun_out/$RND/present_rate_group2.log 2>&1
python bench.py --gpus 2 --share-gpu --steps 256 --warmup 128 > gpurun_out/$RND/bench_2ranks_one_gpu.json 2> gpurun_out/$RND/bench_2ranks_one_gpu.err
bash tools/short_runs.sh > gpurun_out/$RND/short_runs.log 2>&1
{ for L in "" _audit _chaos _audit_chaos; do echo "== libmi355pt$L.so"; timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so 12000 $((700 + ${#L})) | grep -v "^\.\.\."; done
  echo "== multisample focus, batch-pass kernel forced onto tiny images (PT_BATCH_PASS_MIN_TILES=0)"
  PT_BATCH_PASS_MIN_TILES=0 timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit_chaos.so 5000 711 --multisample | grep -v "^\.\.\."; } > gpurun_out/$RND/handover_stress.log 2>&1
{ echo "== general"; timeout 600 python tools/fuzz_parity.py 600 301; echo "== FUZZ_FOCUS=pipelining"; FUZZ_FOCUS=pipelining timeout 600 python tools/fuzz_parity.py 400 302;
  echo "== FUZZ_FOCUS=pipelining FUZZ_BIAS=group_spp under the audit build"; MI355PT_LIB=$R/opentk-pathtracer_amd/libmi355pt_audit.so FUZZ_FOCUS=pipelining FUZZ_BIAS=group_spp timeout 600 python tools/fuzz_parity.py 400 303;
  echo "== FUZZ_FOCUS=grid"; FUZZ_FOCUS=grid timeout 600 python tools/fuzz_parity.py 1000 304; } 2>&1 | grep -v amdgpu.ids > gpurun_out/$RND/fuzz.log
tail -3 gpurun_out/$RND/pytest_gpu.log; cat gpurun_out/$RND/handover_stress.log | grep "handover_stress:\|==" ; grep "cases,\|==" gpurun_out/$RND/fuzz.log; grep "ms per" gpurun_out/$RND/present_rate.log; tail -14 gpurun_out/$RND/bench_configs.log
