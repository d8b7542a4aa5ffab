#!/bin/bash
# PMC passes for the bench_configs.sh workloads that tools/run_round_profiles*.sh do not profile, so that EVERY line of
# profiles/<round>/bench_configs.jsonl carries roofline.traffic / roofline.valu_issue:  bash tools/profile_extra_configs.sh r03
R=/root/repo; RND=${1:-r03}; cd $R; mkdir -p gpurun_out/$RND
export PROFILE_NO_CAL=1 PROFILE_PASSES="stats fetch write sq" PROFILE_BENCH_EXTRA="--no-cpu-baseline" PROFILE_STEPS=256 PROFILE_WARMUP=128
prof() { local tag=$1 key=$2; shift 2
  bash tools/profile_round.sh $tag "$@" > gpurun_out/$RND/${tag}_profile.log 2>&1
  python tools/summarize_profile.py $tag $RND $key ${RND}_default > gpurun_out/$RND/${tag}_summary.log 2>&1; }
prof ${RND}_sky2048 default_1920x1080_d8_spp1_sky2048_g1 --env sky2048
prof ${RND}_d13 default_1920x1080_d13_spp1_atmosphere256_g1 --depth 13
prof ${RND}_C3spp4 stress256_1920x1080_d8_spp4_atmosphere256_g1 --config C3 --spp 4
prof ${RND}_4k default_3840x2160_d8_spp1_atmosphere256_g1_strong4k --strong-4k
prof ${RND}_variant14 default_1920x1080_d8_spp1_atmosphere256_g1_variant14 --variant 14
prof ${RND}_C3nogrid stress256_1920x1080_d8_spp1_atmosphere256_g1_nogrid --config C3 --tune no_sphere_grid=1
unset PROFILE_NO_CAL PROFILE_PASSES PROFILE_BENCH_EXTRA PROFILE_STEPS PROFILE_WARMUP
bash tools/bench_configs.sh > gpurun_out/$RND/bench_configs.log 2>&1; cp gpurun_out/bench_configs.jsonl gpurun_out/$RND/
tail -13 gpurun_out/$RND/bench_configs.log
