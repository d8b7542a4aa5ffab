#!/bin/bash
# Hand-over stress on the GPU box:  bash tools/run_stress.sh <cases per library> <seed> [extra handover_stress args]
# Runs tools/handover_stress.bin against the product library and its audit / chaos / audit+chaos builds; logs under gpurun_out/stress/.
R=/root/repo; N=${1:-2000}; SEED=${2:-1}; shift; shift
cd $R; mkdir -p gpurun_out/stress
for L in "" _audit _chaos _audit_chaos; do
  LOG=gpurun_out/stress/stress${L:-_product}_seed${SEED}.log
  timeout ${STRESS_TIMEOUT:-900} tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so $N $SEED "$@" > $LOG 2>&1
  echo "rc=$?" >> $LOG
  echo "== libmi355pt$L.so"; grep -v "^\.\.\." $LOG | tail -${STRESS_TAIL:-12}
done
