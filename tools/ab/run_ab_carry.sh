R=/root/repo
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for L in "" tools/ab/libC.so; do
  for args in "" "--frame-batch 1" "--config C3" "--config C5"; do
    echo -n "lib[${L:-tree}] [$args] "; MI355PT_LIB=${L:+$R/$L} python $R/bench.py --no-cpu-baseline --steady-ms 0 $args | val
  done
done
done
for L in "" tools/ab/libC.so; do echo -n "lib[${L:-tree}] 1/8 share: "; MI355PT_LIB=${L:+$R/$L} EMULATE_ONLY=1920x1080:8 python tools/emulate_strong.py gpurun_out/exp/es.json | tail -1; done
MI355PT_LIB=$R/tools/ab/libC.so python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_abi_round2.py -m gpu -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -2
