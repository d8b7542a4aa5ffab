#!/bin/bash
# PMC counters of a 1/8 share against the whole 1080p image (per-kernel sums of the integrator launches)
R=/root/repo; cd /tmp; export TMPDIR=/tmp
for sh in 1920x1080:1 1920x1080:8; do
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    O=/tmp/pmc_$$_$RANDOM; mkdir -p $O
    EMULATE_FIXED_WARMUP=4 EMULATE_ONLY=$sh timeout 300 rocprofv3 --pmc $pass --output-format csv -d $O -o p -- python $R/tools/emulate_strong.py /tmp/emu.json > $O/log 2>&1
    python - "$O" "$sh" <<'PY'
import csv, glob, sys, collections
o, sh = sys.argv[1:3]
f = glob.glob(o + "/**/*counter_collection.csv", recursive=True)
tot = collections.Counter(); n = 0
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "pt_integrate" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
print(sh, {k: f"{v:.4g}" for k, v in sorted(tot.items())})
PY
    rm -rf $O
  done
done 2>&1 | tee $R/gpurun_out/share_pmc.log
