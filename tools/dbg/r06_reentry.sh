#!/bin/bash
# round 6, re-entry: the container of the previous session was replaced before the outputs of tools/dbg/r06_final.sh were committed, so
# the default workload's profile (stats + PMC passes + calibration), the driver's command line, the non-overlapped rocprof check and the GPU
# suite are taken again on the final sources (csrc_hash 0773348c8d51), most valuable first: the GPU budget left is one short call.
R=$(pwd); RND=r06; export TMPDIR=/tmp; mkdir -p gpurun_out/$RND
prof() { local tag=${RND}_$1 key=$2; shift 2
  bash tools/profile_round.sh $tag "$@" > gpurun_out/$RND/${tag}_profile.log 2>&1
  python tools/summarize_profile.py $tag $RND $key ${RND}_default > gpurun_out/$RND/${tag}_summary.log 2>&1
  python bench.py $PROFILE_BENCH_EXTRA "$@" > gpurun_out/$RND/${tag}_bench.json 2> gpurun_out/$RND/${tag}_bench.err; }
date +%s > gpurun_out/$RND/reentry_t0.txt
prof default default_1920x1080_d8_spp1_atmosphere256_g1
cp gpurun_out/$RND/${RND}_default_bench.json gpurun_out/$RND/bench_default.json 2>/dev/null
date +%s > gpurun_out/$RND/reentry_t1.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/$RND/driver_command_bench.json 2> gpurun_out/$RND/driver_command_bench.err
bash tools/unchained_timed.sh $RND > gpurun_out/$RND/unchained_timed.log 2>&1
date +%s > gpurun_out/$RND/reentry_t2.txt
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/$RND/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$RND/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > gpurun_out/$RND/smoke.log 2>&1
date +%s > gpurun_out/$RND/reentry_t3.txt
export PROFILE_NO_CAL=1 PROFILE_BENCH_EXTRA="--no-cpu-baseline"
prof perframe default_1920x1080_d8_spp1_atmosphere256_g1_fb1 --frame-batch 1
unset PROFILE_NO_CAL PROFILE_BENCH_EXTRA
{ echo "== libmi355pt.so"; timeout 200 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so 3000 1700 | grep -v "^\.\.\."
  echo "== frame-fed launches on every image size"; timeout 200 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit.so 3000 1720 --tune feed_min_tiles=0 | grep -v "^\.\.\."; } > gpurun_out/$RND/handover_stress_reentry.log 2>&1
tail -3 gpurun_out/$RND/pytest_gpu.log; tail -c 300 gpurun_out/$RND/driver_command_bench.json; tail -2 gpurun_out/$RND/unchained_timed.log; grep "handover_stress:" gpurun_out/$RND/handover_stress_reentry.log
