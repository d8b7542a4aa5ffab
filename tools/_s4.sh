mkdir -p gpurun_out/s4
env | grep -i "^HSA\|^HIP\|^GPU_\|^ROC\|^AMD" > gpurun_out/s4/env.txt
for v in "X=1" "HSA_ENABLE_SDMA=1" "GPU_FORCE_BLIT_COPY_SIZE=0" "HSA_ENABLE_SDMA=0" "GPU_MAX_HW_QUEUES=8"; do
  echo "== $v" >> gpurun_out/s4/log.txt
  env $v timeout 120 python tools/present_rate.py --frames 300 2>&1 | grep "async\|blocking" >> gpurun_out/s4/log.txt
done
cat gpurun_out/s4/env.txt; cat gpurun_out/s4/log.txt
