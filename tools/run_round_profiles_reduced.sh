#!/bin/bash
# Reduced variant of tools/run_round_profiles.sh for a re-profile after a HOST-ONLY change late in a round (the kernels are unchanged, the
# source hash is not): all six workloads, but only the passes bench.py's roofline needs for the secondary ones, no stress / fuzz / short-run
# logs (those of the full run stay, stamped with the hash they ran on).  bash tools/run_round_profiles_reduced.sh r03
R=/root/repo
RND=${1:-r03}
cd $R
mkdir -p gpurun_out/$RND
prof() { # tag, workload key, calibration tag, bench args...
  local tag=$1 key=$2 cal=$3; shift 3
  bash tools/profile_round.sh $tag "$@" > gpurun_out/$RND/${tag}_profile.log 2>&1
  python tools/summarize_profile.py $tag $RND $key $cal > gpurun_out/$RND/${tag}_summary.log 2>&1
  python bench.py $PROFILE_BENCH_EXTRA "$@" > gpurun_out/$RND/${tag}_bench.json 2> gpurun_out/$RND/${tag}_bench.err
}
prof ${RND}_default default_1920x1080_d8_spp1_atmosphere256_g1 ${RND}_default
export PROFILE_NO_CAL=1 PROFILE_PASSES="stats fetch write rd wr sq" PROFILE_BENCH_EXTRA="--no-cpu-baseline"
prof ${RND}_perframe default_1920x1080_d8_spp1_atmosphere256_g1_fb1 ${RND}_default --frame-batch 1
prof ${RND}_C3 stress256_1920x1080_d8_spp1_atmosphere256_g1 ${RND}_default --config C3
prof ${RND}_C5 glass_1920x1080_d32_spp1_atmosphere256_g1 ${RND}_default --config C5
export PROFILE_STEPS=192 PROFILE_WARMUP=64
prof ${RND}_spp4 default_1920x1080_d8_spp4_atmosphere256_g1 ${RND}_default --spp 4
prof ${RND}_tilewave default_1920x1080_d8_spp1_atmosphere256_g1_variant1 ${RND}_default --variant 1
unset PROFILE_NO_CAL PROFILE_STEPS PROFILE_WARMUP PROFILE_PASSES PROFILE_BENCH_EXTRA
bash tools/bench_configs.sh > gpurun_out/$RND/bench_configs.log 2>&1; cp gpurun_out/bench_configs.jsonl gpurun_out/$RND/
python bench.py --steps 20 --warmup 5 > gpurun_out/$RND/driver_command_bench.json 2> gpurun_out/$RND/driver_command_bench.err
python tools/present_rate.py --json gpurun_out/$RND/present_rate.json > gpurun_out/$RND/present_rate.log 2>&1
python -m pytest tests -m gpu -q > gpurun_out/$RND/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$RND/pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/$RND/smoke.log 2>&1
python tools/emulate_strong.py gpurun_out/$RND/emulate_strong.json > gpurun_out/$RND/emulate_strong.log 2>&1
python bench.py --gpus 2 --share-gpu --steps 256 --warmup 128 > gpurun_out/$RND/bench_2ranks_one_gpu.json 2> gpurun_out/$RND/bench_2ranks_one_gpu.err
tail -3 gpurun_out/$RND/pytest_gpu.log; grep "ms per" gpurun_out/$RND/present_rate.log; tail -14 gpurun_out/$RND/bench_configs.log
