#!/bin/bash
# one SQ counter pass per (world, tune set): tools/ab/share_counters2.sh "world:tune,tune ..." ...
export TMPDIR=/tmp; R=/root/repo; cd /tmp
for cfg in "$@"; do
  world=${cfg%%:*}; tune=${cfg#*:}
  rm -rf /tmp/sc; EMULATE_TUNE="$tune" EMULATE_FIXED_WARMUP=1 EMULATE_ONLY=1920x1080:$world rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d /tmp/sc -o sc -- python $R/tools/emulate_strong.py /tmp/es.json > /tmp/sc.log 2>&1
  python - <<PY
import csv, collections, glob
world = $world
ranks = len({0, world // 2, world - 1})
frames = ranks * 832 / world
acc = collections.defaultdict(float)
for f in glob.glob("/tmp/sc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_integrate" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
g = lambda k: acc[k] / frames / 1e6
print("world %d [%-28s] VALU %7.2f  SALU %6.2f  VMEM_RD %6.3f  VMEM_WR %6.3f  LDS %5.2f  SMEM %5.2f  BRANCH %5.2f  BUSY %6.2f   (M per whole-frame equivalent)" % (world, "$tune", g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), g("SQ_INSTS_LDS"), g("SQ_INSTS_SMEM"), g("SQ_INSTS_BRANCH"), g("SQ_BUSY_CYCLES")))
PY
done
