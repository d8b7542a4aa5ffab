#!/bin/bash
# After `gpurun -- bash tools/run_round_profiles.sh r03` has merged gpurun_out/: summarise the raw counters locally (tracked files under
# profiles/) and copy the logs the round commits.   bash tools/collect_round_profiles.sh r03
R=/root/repo; RND=${1:-r03}; cd $R; mkdir -p profiles/$RND
s() { python tools/summarize_profile.py $1 $RND $2 ${RND}_default > /dev/null && cp gpurun_out/$RND/$1_bench.json profiles/$RND/$1_bench.json; }
s ${RND}_default default_1920x1080_d8_spp1_atmosphere256_g1
s ${RND}_perframe default_1920x1080_d8_spp1_atmosphere256_g1_fb1
s ${RND}_C3 stress256_1920x1080_d8_spp1_atmosphere256_g1
s ${RND}_C5 glass_1920x1080_d32_spp1_atmosphere256_g1
s ${RND}_spp4 default_1920x1080_d8_spp4_atmosphere256_g1
s ${RND}_tilewave default_1920x1080_d8_spp1_atmosphere256_g1_variant1
for f in pytest_gpu.log bench_configs.jsonl driver_command_bench.json emulate_strong.json present_rate.json present_rate_group2.json bench_2ranks_one_gpu.json short_runs.log handover_stress.log fuzz.log smoke.log; do
  [ -f gpurun_out/$RND/$f ] && cp gpurun_out/$RND/$f profiles/$RND/$f
done
ls profiles/$RND
