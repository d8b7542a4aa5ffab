#!/bin/bash
# round 6, last GPU seconds: the zero-budget audit + chaos stress with the tool's site-91 classification (late starts behind an abandonment
# are the design, not violations), then the tests of the hand-over bound and round 6's own tests as far as the remaining budget reaches
mkdir -p gpurun_out/r06
timeout 60 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit_chaos.so 600 513 --tune handover_budget_ms=0 2>&1 | grep -v "^\.\.\." | tail -6 > gpurun_out/r06/site91_check.log; echo "rc=${PIPESTATUS[0]}" >> gpurun_out/r06/site91_check.log
cat gpurun_out/r06/site91_check.log
timeout 110 python -m pytest tests/test_gpu_handover_bound.py tests/test_gpu_round6.py -m gpu -q -x > gpurun_out/r06/pytest_gpu_tail.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06/pytest_gpu_tail.log
tail -4 gpurun_out/r06/pytest_gpu_tail.log
