cd /root/repo
for rep in 1 2; do
for lib in "" tools/ab/libMSold.so tools/ab/libMS_noearly.so; do
  for cfg in "--config C3 --spp 4" "--config C3 --spp 2" "--spp 4"; do
    L=${lib:-tree}
    v=$(MI355PT_LIB=${lib:+/root/repo/$lib} timeout 300 python bench.py $cfg --steps 192 --warmup 64 --no-cpu-baseline --steady-ms 400 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['steady']['value'])")
    echo "$L | $cfg | $v"
  done
done
done
