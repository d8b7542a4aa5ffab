#!/bin/bash
# SQ counters of the integrator per whole-frame equivalent: the whole 1080p image against one rank's 1/4 and 1/8 share (tools/emulate_strong.py,
# fixed frame counts).  tools/ab/share_counters.sh  ->  stdout
export TMPDIR=/tmp; R=/root/repo; cd /tmp
for world in 1 4 8; do
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" \
              "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_WAIT_INST_LDS"; do
    rm -rf /tmp/sc; EMULATE_FIXED_WARMUP=1 EMULATE_ONLY=1920x1080:$world rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/sc -o sc -- python $R/tools/emulate_strong.py /tmp/es.json > /tmp/sc.log 2>&1
    python - <<PY
import csv, collections, glob
world = $world
ranks = len({0, world // 2, world - 1})
frames = ranks * 832 / world          # whole-frame equivalents rendered
acc = collections.defaultdict(float)
for f in glob.glob("/tmp/sc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_integrate" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()): print(f"world {world}  {k:22s} {v / frames / 1e6:12.3f} M per whole-frame equivalent")
PY
  done
done
