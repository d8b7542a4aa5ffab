mkdir -p gpurun_out/s2
python tools/_diag_present.py > gpurun_out/s2/diag.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /root/repo/gpurun_out/s2/trace -o t -- python /root/repo/tools/_diag_present.py > /root/repo/gpurun_out/s2/trace.log 2>&1
cd /root/repo; ls gpurun_out/s2/trace; cat gpurun_out/s2/diag.log | tail -5
# keep only the tail of big traces
for f in gpurun_out/s2/trace/*trace.csv; do tail -n 3000 $f > $f.tail; rm $f; done
