#!/bin/bash
# build an A/B variant of the library:  tools/ab/build_ab.sh <letter> [extra hipcc flags, e.g. -DPT_EXPERIMENT]
R=/root/repo; C=${AB_SRC:-$R/opentk-pathtracer_amd/csrc}; L=$1; shift   # AB_SRC: an experimental copy of csrc (keeps the tree's hash, and the profiles stamped with it, valid)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared -fvisibility=hidden "$@" \
  $C/pt_integrate_persistent.hip $C/pt_integrate_multisample.hip $C/pt_helper_kernels.hip $C/mi355pt.cpp $C/mi355pt_multi.cpp -o $R/tools/ab/lib$L.so 2>&1 | grep -v "warning\|^$" | head -20
ls -la $R/tools/ab/lib$L.so
