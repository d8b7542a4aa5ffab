mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s1/pytest.log
timeout 300 python tools/present_rate.py --json gpurun_out/s1/present_rate.json > gpurun_out/s1/present_rate.log 2>&1
timeout 300 python tools/present_rate.py --devices 0,0 --json gpurun_out/s1/present_rate_group2.json > gpurun_out/s1/present_rate_group2.log 2>&1
timeout 600 python bench.py > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --share-gpu --steps 256 --warmup 128 > gpurun_out/s1/bench_share2.json 2> gpurun_out/s1/bench_share2.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > /root/repo/gpurun_out/s1/counters.txt 2>&1)
tail -3 gpurun_out/s1/pytest.log; cat gpurun_out/s1/present_rate.log | tail -6; tail -c 400 gpurun_out/s1/bench.json
