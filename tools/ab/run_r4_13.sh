#!/bin/bash
R=/root/repo
cd $R
mkdir -p gpurun_out/r4_13
val() { python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo "== parity, affine"; cat > /tmp/conftest_tune.py <<'PY'
PY
timeout 300 python - <<'PY' 2>&1 | tail -5
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as g, configs
pkg = g.load_package(); oracle = g.load_oracle().Oracle()
pkg.native.debug_set('xcd_affine', 1)
bad = 0
for w in [configs.Workload("a", "default", 256, 144, 8, "sky_f32_32", frames=70), configs.Workload("b", "glass", 640, 360, 12, "sky_f32_32", frames=40),
          configs.Workload("c", "default", 64, 40, 8, "sky_f32_32", frames=130), configs.Workload("d", "stress256", 320, 200, 8, "sky_f32_32", frames=66)]:
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
    pt.UploadScene(sc); pt.UploadBasicData(basic)
    for _ in range(w.frames): pt.Render()
    got = pt.Result; pt.Dispose()
    want = oracle.render(w.width, w.height, basic, objs, env, num_frames=w.frames, **kw)
    same = (got.view(np.uint32) == want.view(np.uint32)).all(-1)
    print(w.name, w.scene, w.width, w.height, w.frames, "differing pixels:", int((~same).sum())); bad += int((~same).sum())
print("TOTAL differing", bad)
PY
for rep in 1 2; do
  for t in "" "--tune xcd_affine=1"; do
    for args in "" "--config C3" "--config C5"; do echo -n "[$t] [$args] "; timeout 120 python bench.py --no-cpu-baseline --steady-ms 0 $args $t 2>/dev/null | val; done
  done
done | tee gpurun_out/r4_13/ab.log
bash tools/traffic_quick.sh base 2>&1 | tail -1 | cut -c1-600
bash tools/traffic_quick.sh affine --tune xcd_affine=1 2>&1 | tail -1 | cut -c1-600
