cd /root/repo
for rep in 1 2; do
for lib in "" tools/ab/libMSold.so; do
  for cfg in "--spp 4" "--config C3 --spp 4" "--spp 2" "--config C3" ""; do
    L=${lib:-tree}
    v=$(MI355PT_LIB=${lib:+/root/repo/$lib} timeout 300 python bench.py $cfg --steps 192 --warmup 64 --no-cpu-baseline --steady-ms 400 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['steady']['value'])")
    echo "$L | $cfg | $v"
  done
done
done
for gc in 0 1 2; do
  timeout 300 python bench.py --config C3 --steps 192 --warmup 64 --no-cpu-baseline --steady-ms 400 --tune grid_carry=$gc --tune log_launch=3 2> /tmp/gc.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C3 grid_carry=$gc', d['value'], d['steady']['value'])"
  grep "frames 64" /tmp/gc.err | tail -n 1
done
