#!/bin/bash
# Everything a round commits under profiles/<round>/ in ONE gpurun call:  bash tools/run_round_profiles.sh r02
R=/root/repo
RND=${1:-r02}
cd $R
mkdir -p gpurun_out/$RND
python -m pytest tests -m gpu -q > gpurun_out/$RND/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$RND/pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/$RND/smoke.log 2>&1
bash tools/profile_round.sh ${RND}_default
PROFILE_NO_CAL=1 bash tools/profile_round.sh ${RND}_perframe --frame-batch 1
PROFILE_NO_CAL=1 bash tools/profile_round.sh ${RND}_C3 --config C3
PROFILE_NO_CAL=1 bash tools/profile_round.sh ${RND}_C5 --config C5
PROFILE_NO_CAL=1 PROFILE_STEPS=192 PROFILE_WARMUP=64 bash tools/profile_round.sh ${RND}_spp4 --spp 4
PROFILE_NO_CAL=1 PROFILE_STEPS=192 PROFILE_WARMUP=64 bash tools/profile_round.sh ${RND}_tilewave --variant 1
bash tools/bench_configs.sh > gpurun_out/$RND/bench_configs.log 2>&1; cp gpurun_out/bench_configs.jsonl gpurun_out/$RND/
python tools/emulate_strong.py gpurun_out/$RND/emulate_strong.json > gpurun_out/$RND/emulate_strong.log 2>&1
python tools/present_rate.py --json gpurun_out/$RND/present_rate.json > gpurun_out/$RND/present_rate.log 2>&1
python tools/present_rate.py --devices 0,0 --json gpurun_out/$RND/present_rate_group2.json > gpurun_out/$RND/present_rate_group2.log 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-gpu --steps 256 --warmup 128 > gpurun_out/$RND/bench_2ranks_one_gpu.json 2> gpurun_out/$RND/bench_2ranks_one_gpu.err
bash tools/short_runs.sh > gpurun_out/$RND/short_runs.log 2>&1
{ echo "== general"; timeout 600 python tools/fuzz_parity.py 600 101; echo "== FUZZ_FOCUS=pipelining"; FUZZ_FOCUS=pipelining timeout 600 python tools/fuzz_parity.py 300 102;
  echo "== FUZZ_FOCUS=grid"; FUZZ_FOCUS=grid timeout 600 python tools/fuzz_parity.py 1000 103; } 2>&1 | grep -v amdgpu.ids > gpurun_out/$RND/fuzz.log
tail -3 gpurun_out/$RND/pytest_gpu.log; cat gpurun_out/$RND/fuzz.log; cat gpurun_out/$RND/present_rate.log | grep "ms per"; cat gpurun_out/$RND/bench_configs.log | tail -9
