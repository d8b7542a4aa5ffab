b() { python /root/repo/bench.py --no-cpu-baseline --steps 640 --warmup 128 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"; }
echo "C3 default heuristic: $(b --config C3)"
echo "default scene brute: $(b)   grid: $(PT_GRID_MIN_SPHERES=32 b)"
echo "glass scene brute: $(b --config C5)   grid: $(PT_GRID_MIN_SPHERES=32 b --config C5)"
for n in 64 96 128 192; do
echo "stress $n spheres: brute $(BENCH_STRESS_SPHERES=$n PT_NO_SPHERE_GRID=1 b --scene stress256)  grid $(BENCH_STRESS_SPHERES=$n PT_GRID_MIN_SPHERES=32 b --scene stress256)"
done
