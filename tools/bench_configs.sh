#!/bin/bash
# BASELINE.json's other configs through bench.py (one JSON line each) -> gpurun_out/bench_configs.jsonl
R=/root/repo
O=$R/gpurun_out/bench_configs.jsonl
: > $O
python $R/bench.py --no-cpu-baseline --steady-ms 0 >> $O                                       # C2: default scene, 1080p, 8 bounces
python $R/bench.py --no-cpu-baseline --steady-ms 0 --config C3 >> $O                     # C3: 256-sphere scene (fills the UBO)
python $R/bench.py --no-cpu-baseline --steady-ms 0 --config C5 >> $O              # C5: glass-heavy, 32 bounces, atmosphere env
python $R/bench.py --no-cpu-baseline --steady-ms 0 --env sky2048 >> $O                         # default scene with a 2048^2 sRGB sky cube
python $R/bench.py --no-cpu-baseline --steady-ms 0 --depth 13 >> $O                            # the reference's shipped default depth
python $R/bench.py --no-cpu-baseline --steady-ms 0 --spp 4 --steps 240 --warmup 80 >> $O                   # 4 samples per pixel per frame
python $R/bench.py --no-cpu-baseline --steady-ms 0 --config C3 --spp 4 --steps 240 --warmup 80 >> $O       # C3 at 4 samples per pixel
python $R/bench.py --no-cpu-baseline --steady-ms 0 --config C3 --tune no_sphere_grid=1 >> $O                   # C3 with the reference's in-order sphere loop (no grid)
python $R/bench.py --no-cpu-baseline --steady-ms 0 --tune carry_last=0 >> $O                    # default scene, every resolve loads its pixel (the kernel of rounds 1-3)
python $R/bench.py --no-cpu-baseline --steady-ms 0 --frame-batch 1 >> $O                       # one launch per frame (2 overlapping row stripes)
python $R/bench.py --no-cpu-baseline --steady-ms 0 --strong-4k >> $O                          # 3840x2160 on one GPU
python $R/bench.py --no-cpu-baseline --steady-ms 0 --variant 1 >> $O                           # tile-per-wave kernel (reference mapping)
python $R/bench.py --no-cpu-baseline --steady-ms 0 --variant 14 >> $O                          # persistent kernel, one launch per frame
python - <<PY
import json
for l in open("$O"):
    d = json.loads(l); print(d["value"], d["ms_per_step"], d["config"]["workload"][:90], "variant", d["config"]["kernel_variant"])
PY
