mkdir -p gpurun_out/s3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s3/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s3/pytest.log
python tools/_diag_present.py > gpurun_out/s3/diag.log 2>&1
timeout 300 python tools/present_rate.py --json gpurun_out/s3/present_rate.json > gpurun_out/s3/present_rate.log 2>&1
tail -3 gpurun_out/s3/pytest.log; tail -4 gpurun_out/s3/diag.log; tail -5 gpurun_out/s3/present_rate.log
