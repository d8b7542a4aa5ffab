#!/bin/bash
# Round-5 acceptance runs of the hand-over bound with the library's DEFAULT knobs (no chain_wait_us workaround):
#   tools/ab/handover_repeat.sh <out-dir> [two-rank runs = 200] [pytest repeats = 50]
#   1. tools/ab/run_2rank.sh N: `bench.py --gpus 2 --share-gpu` (two processes oversubscribing one GPU, launches chained beside their
#      predecessors: the configuration that ended in error -5 about once in 15 runs in round 4) N times;
#   2. tests/test_gpu_round3.py::test_bench_self_spawns_ranks_and_refuses_missing_devices M times in a row.
R=/root/repo; cd $R
OUT=${1:-gpurun_out/handover_repeat}; mkdir -p $OUT
N=${2:-200}; M=${3:-50}
python -c "import __graft_entry__ as g; print('csrc_hash', g.load_package().native.csrc_hash())" > $OUT/two_rank_repeat.log
bash tools/ab/run_2rank.sh $N tree >> $OUT/two_rank_repeat.log 2>&1
tail -2 $OUT/two_rank_repeat.log
bad=0
for i in $(seq 1 $M); do
  timeout 600 python -m pytest tests/test_gpu_round3.py -q -m gpu -k test_bench_self_spawns_ranks_and_refuses_missing_devices > /tmp/spawn.log 2>&1 || { bad=$((bad+1)); echo "repeat $i FAILED: $(tail -3 /tmp/spawn.log | tr '\n' ' ')"; }
done > $OUT/self_spawn_repeat.log 2>&1
echo "test_bench_self_spawns_ranks_and_refuses_missing_devices: $bad of $M repeats failed" | tee -a $OUT/self_spawn_repeat.log
