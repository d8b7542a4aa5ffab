"""Strong scaling emulated on ONE GPU: render a single rank's share (block-cyclic 16-row bands) of the 1920x1080 and the
3840x2160 image for world sizes 1 / 2 / 4 / 8 and report the slowest of three ranks (first, middle, last).  Ranks are
independent (no data-path collective), so the N-GPU frame time is the slowest rank's time; the gather happens only at
present time.  -> gpurun_out/<file>.json (committed under profiles/ by the round's profile script)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
BAND = int(os.environ.get("EMULATE_BAND", "8"))  # rows per band of the block-cyclic split (bench.py: BAND)
out = {"csrc_hash": pkg.native.csrc_hash(), "band_rows": BAND, "results": {}}
sc, cam = pkg.scene.default_scene(), pkg.camera.Camera()
only = os.environ.get("EMULATE_ONLY")  # e.g. "1920x1080:8" (tuning runs)
for kv in filter(None, os.environ.get("EMULATE_TUNE", "").split(",")):  # e.g. "parked_max=16,batch_wg=5" (tuning runs)
    k, v = kv.split("=")
    pkg.native.debug_set(k, int(v))
    out.setdefault("tuning", {})[k] = int(v)
for (W, H) in ((1920, 1080), (3840, 2160)):
    for world in (1, 2, 4, 8):
        if only and only != f"{W}x{H}:{world}":
            continue
        worst = 0.0
        for rank in sorted({0, world // 2, world - 1}):
            pt = pkg.PathTracer(None, W, H, 8, 1, 20.0, 0.14)
            pt.EnvironmentMap = pkg.AtmosphericScatterer(256, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)
            pt.UploadScene(sc); pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, W, H))
            if world > 1:
                pt.SetInterleavedTile(rank, world, BAND)
            t = time.perf_counter()
            fixed = int(os.environ.get("EMULATE_FIXED_WARMUP", "0"))  # counter runs: a known number of frames (n rounds of 64) instead of 80 ms
            for _ in range(fixed):
                for _ in range(64): pt.Render()
                pt.Synchronize()
            while not fixed and time.perf_counter() - t < 0.08:
                for _ in range(64): pt.Render()
                pt.Synchronize()
            steps = 640
            for _ in range(128): pt.Render()
            pt.Synchronize()
            pt.TimerBegin()
            for _ in range(steps): pt.Render()
            ms = pt.TimerEnd() / steps
            worst = max(worst, ms)
            pt.Dispose()
        out["results"][f"{W}x{H}_world{world}"] = {"ms_per_frame_slowest_rank": round(worst, 5),
                                                   "predicted_msamples_per_s": round(W * H / worst / 1e3, 1)}
        print(f"{W}x{H} world {world}: {worst:.4f} ms per frame (slowest rank) -> {W * H / worst / 1e3:.0f} Msamples/s", flush=True)
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/emulate_strong.json"
os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
