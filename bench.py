#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: Msamples/s of the path-tracing integrator on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a torchrun environment: bench.py starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Either way N > 1 means one process per GPU over RCCL, and the run FAILS if the box has fewer than N devices (it never
degrades to fewer GPUs silently; --share-gpu is the explicit one-GPU debug mode).  `n_gpus` is the number of distinct devices that
rendered (gathered from the ranks), `ranks` the RCCL world size.

A "step" is one PathTracer.Render() — one pass of the integrator over the whole image with the inputs (scene
UBO, camera UBO, environment cube, accumulation image) already resident in HBM; frames accumulate progressively
exactly as in the reference (src/Render/PathTracer.cs:114-123).  The library launches consecutive Render() calls as one
kernel that pipelines up to --frame-batch frames; every frame is computed in full inside the
timed region, which ends with pt_timer_end + pt_synchronize.  Defaults: 960 steps after 320 warm-up steps; before the
warm-up steps the GPU is kept busy for --clock-warmup-ms (80 ms, reported in the JSON line) because it needs ~40 ms of
load to reach its steady clock — that time is neither a step nor timed, and the accumulation restarts at frame 0 after it.

Workload = BASELINE.json's metric ("@1080p 8-bounce default scene, 1/2/4/8 GPU"):
  N = 1  BASELINE configs[1] (--config C2): default scene (48 spheres + 7 cuboids), 1920x1080, 8 bounces, 1 spp, the
         reference's default environment (256^2 RGBA32F atmosphere cube, computed by the atmosphere kernel).
  N > 1  the SAME 1920x1080 image row-tiled over the N GPUs (STRONG scaling, "scaling": "strong"): one process per GPU,
         block-cyclic 8-row bands (one tile row each) (floor rows cost ~2x sky rows), no data-path collective; the RCCL gather happens only at
         present time and is timed separately (`present_ms`).  value = all pixels x spp x steps / max-over-ranks wall time.
         The same run then measures BASELINE configs[3] — ONE 3840x2160 image over the N GPUs — and reports it inside the
         same JSON line as "configs3_4k" (the driver's contract is one line).
  --config C3 | C5 select BASELINE configs[2] (256-sphere scene) / configs[4] (glass scene, 32 bounces); --weak grows the
  image with N (~N x 2.07 Mpixel) instead; --strong-4k makes configs[3] the headline workload.

The JSON line also carries
  roofline     : the integrator kernel against the HBM roof, from ALGORITHMIC bytes (32 B per pixel per frame: one
                 float4 load + one float4 store of the accumulation image) and the kernel's average duration measured
                 with HIP events on the stream it runs on;  `valu` gives the roof that actually binds (fp32 vector ALU);
                 `traffic` / `valu_issue` come from the PMC passes committed under profiles/ and are dropped (null + a
                 note) when those passes were taken with different kernel sources than the library now built;
  cpu_baseline : the C restatement (oracle/, "port") timed on this box's host cores on a bounded sample of the same
                 workload (all threads, and one thread) — rank 0, N = 1 only — plus the REFERENCE'S OWN GLSL on Mesa
                 llvmpipe as recorded in the build container (profiles/reference_llvmpipe.json; llvmpipe does not exist on
                 the GPU box).  The oracle is used here ONLY as the thing timed, never by the GPU path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP32_VALU_PEAK_TF = 157.3  # same guide: peak FP32 vector
ALGO_BYTES_PER_PIXEL_FRAME = 32  # SURVEY.md section 8d / BASELINE.md section 4

WEAK_SIZES = {1: (1920, 1080), 2: (2720, 1530), 4: (3840, 2160), 8: (5440, 3060)}
# BASELINE.json configs by name: (scene, depth, env, BASELINE index)
CONFIGS = {"C2": ("default", 8, "atmosphere256", 1), "C3": ("stress256", 8, "atmosphere256", 2), "C5": ("glass", 32, "atmosphere256", 4)}
BAND = 8   # block-cyclic bands of ONE tile row (8 image rows) across ranks: row cost varies ~2x between sky and floor rows, and 1080 rows are
           # 135 tile rows — 8 GPUs own 16 or 17 of them (136 rows at most; 16-row bands gave the largest share 144 rows: 6.7 % over the mean)


def weak_image_size(n_gpus: int) -> tuple[int, int]:
    if n_gpus in WEAK_SIZES:
        return WEAK_SIZES[n_gpus]
    w = int(round(1920 * n_gpus ** 0.5 / 16)) * 16
    return w, w * 9 // 16


def flops_per_sample(mean_bounces: float, ns: int, nc: int) -> float:
    """SURVEY.md section 8d model: 100 + sum over bounces of (17 Ns + 24 Nc + 120)."""
    return 100.0 + mean_bounces * (17.0 * ns + 24.0 * nc + 120.0)


def load_profile_number(file_name: str, workload_key: str, csrc_hash: str):
    """A per-step number measured by the rocprofv3 PMC passes committed under profiles/ -> (value or None, note or None).
    Entries are stamped with the hash of the kernel sources they were measured on (tools/summarize_profile.py); a stale
    entry is NOT used."""
    p = os.path.join(ROOT, "profiles", file_name)
    if not os.path.exists(p):
        return None, None
    try:
        e = json.load(open(p)).get(workload_key)
    except Exception:
        return None, None
    if e is None:
        return None, None
    if not isinstance(e, dict) or e.get("csrc_hash") != csrc_hash:
        return None, f"profiles/{file_name} holds a measurement of other kernel sources for this workload: not used"
    return e["value"], None


def load_profile_entry(file_name: str, workload_key: str, csrc_hash: str) -> dict:
    """The whole entry behind load_profile_number (same staleness rule), or {}."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", file_name))).get(workload_key)
    except Exception:
        return {}
    return e if isinstance(e, dict) and e.get("csrc_hash") == csrc_hash else {}


# VALU issue roof: the 157.3 TFLOP/s FP32 vector peak = 1024 SIMDs x 32 lanes x 2 flops x 2.4 GHz, i.e. at best one
# wave64 instruction per 2 clocks per SIMD.  (A dependence-free v_fma_f32 stream sustains one per 2.63 clocks with
# 4 wavefronts per SIMD, tools/ubench2.hip; the integrator's mix — with instructions that retire early under an empty
# EXEC mask — runs at one per ~2.4.)
VALU_ISSUE_PEAK_GINST = 1024 * 2.4 / 2.0


def effective_cpus() -> tuple[int, str]:
    """CPUs this process may actually use: the scheduler affinity, capped by the container's CPU quota (cgroup v2 cpu.max / v1
    cfs quota).  The GPU boxes show 256 hardware threads to a container that is allowed 16 CPUs' worth of time — threads beyond the
    quota only add contention, and a baseline quoted per 256 'cores' would be a strawman."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} hardware threads visible"
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota is not None and quota < n:
        note += f", container CPU quota {quota:g} CPUs (cgroup)"
        n = max(1, int(quota + 0.5))
    return n, note


def cpu_baseline(pkg, scene, basic, env, width, height, depth, spp, workload_name):
    """Time the oracle ("port") on the host cores on a bounded sample: whole frames of the same workload until
    ~10 s have elapsed (at least one frame) on all threads, then a thin row block on ONE thread (~3 s)."""
    oracle = graft.load_oracle().Oracle()
    cores, cores_note = effective_cpus()
    kw = dict(num_spheres=scene.num_spheres, num_cuboids=scene.num_cuboids, ray_depth=depth, spp=spp)
    objs = scene.ubo_bytes()
    # warm the caches / page in with a thin row block
    oracle.render(width, height, basic, objs, env, y0=height // 2, rows=8, threads=cores, **kw)
    frames, t0 = 0, time.perf_counter()
    img = np.zeros((height, width, 4), np.float32)
    while True:
        oracle.render(width, height, basic, objs, env, frame_start=frames, num_frames=1, image=img, threads=cores, **kw)
        frames += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or frames >= 64:
            break
    # one thread: row blocks spread over the image (rows differ ~2x in cost), until ~3 s have elapsed
    rows1, t1, k = 0, time.perf_counter(), 0
    starts = [int(height * f) // 8 * 8 for f in (0.5, 0.1, 0.9, 0.3, 0.7, 0.2, 0.8, 0.4, 0.6, 0.0)]
    while True:
        y0 = min(starts[k % len(starts)] + 8 * (k // len(starts)), height - 8)
        oracle.render(width, height, basic, objs, env, y0=y0, rows=8, threads=1, **kw)
        rows1 += 8
        k += 1
        dt1 = time.perf_counter() - t1
        if dt1 > 3.0 or k >= 400:
            break
    _, st = oracle.render(width, height, basic, objs, env, y0=0, rows=height, threads=cores, want_stats=True, **kw)
    out = {
        "value": round(width * height * spp * frames / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"{frames} full frame(s) of the same workload ({width}x{height}, {depth} bounces, {spp} spp) in {dt:.1f} s, "
                  f"oracle/pt_oracle.c on {cores} threads (persistent pool, dynamic 4-row chunks; {cores_note})",
        "one_thread": {"value": round(width * rows1 * spp / dt1 / 1e6, 4), "unit": "Msamples/s", "cores": 1,
                       "sample": f"{rows1} rows ({k} blocks of 8 spread over the image) of one frame in {dt1:.1f} s"},
    }
    # the reference's own code on a CPU: its GLSL on llvmpipe, recorded in the build container (no llvmpipe on the GPU box)
    p = os.path.join(ROOT, "profiles", "reference_llvmpipe.json")
    if os.path.exists(p):
        try:
            rec = json.load(open(p))
            run = rec["runs"].get(workload_name)
            if run:
                out["reference_glsl_llvmpipe"] = {
                    "value": run["all_threads"]["msamples_per_s"], "unit": "Msamples/s", "cores": run["all_threads"]["threads"],
                    "ms_per_frame": run["all_threads"]["ms_per_frame"],
                    "one_thread": {"value": run["one_thread"]["msamples_per_s"], "ms_per_frame": run["one_thread"]["ms_per_frame"]},
                    "kind": "reference", "measured": "NOT in this run: recorded in the build container (8 vCPUs, "
                                                     f"{rec['provenance']['cpu']}), {rec['what']}",
                    "workload": workload_name}
        except Exception:
            pass
    return out, st["bounces"] / max(1, st["samples"]), st["env_lookups"] / max(1, st["samples"])


def self_spawn(args) -> int:
    """`python bench.py --gpus N` with no torchrun environment: start the N ranks here (one process per GPU, the same
    environment torchrun would set), pass rank 0's JSON line through, fail if any rank fails or the box has fewer devices."""
    import socket
    import subprocess

    pkg = graft.load_package()
    if not os.path.exists(pkg.native.LIB_PATH):
        graft.build()
    have = pkg.native.load().pt_device_count()
    if have < args.gpus and not args.share_gpu:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this box has {have} HIP device(s); refusing to report a {args.gpus}-GPU number "
                         f"from fewer GPUs (use --share-gpu for the one-GPU debug mode)")
    rc = 1
    for attempt in range(3):  # (the rendezvous port is found by bind + close: another process can take it before rank 0 listens — retry)
        t_start = time.time()
        rc = _spawn_ranks(args, socket, subprocess)
        if rc == 0 or time.time() - t_start > 20.0:
            break
    return rc


def _spawn_ranks(args, socket, subprocess) -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll ALL ranks: when one dies, the others may sit in a collective until the RCCL watchdog fires — end them at once
    rc, alive = 0, list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in alive:
                    q.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=960)
    ap.add_argument("--warmup", type=int, default=320)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="BASELINE.json config preset (scene, depth, env); default C2")
    ap.add_argument("--scene", default=None, choices=["default", "stress256", "glass"])
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--env", default=None, choices=["atmosphere256", "sky2048", "sky64"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--frame-batch", type=int, default=0,
                    help="frames one launch may pipeline (pt_set_frame_batch; 1 = one launch per Render(); 0 = the library's automatic "
                         "choice: 64, or 256 on a share of fewer than 12,000 tiles)")
    ap.add_argument("--clock-warmup-ms", type=float, default=80.0,
                    help="wall time of untimed rendering before the W warm-up steps (GPU clock ramp); 0 = none")
    ap.add_argument("--strong-4k", action="store_true",
                    help="make BASELINE configs[3] the headline workload: ONE 3840x2160 image row-tiled over the N GPUs")
    ap.add_argument("--weak", action="store_true", help="N > 1: grow the 16:9 image to ~N x 2.07 Mpixel (weak scaling)")
    ap.add_argument("--no-4k", action="store_true", help="N > 1: skip the extra configs[3] measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--steady-ms", type=float, default=500.0,
                    help="after the timed K steps, render the same workload for at least this long in the same process and report it "
                         "as `steady` (never the metric's value): what a short driver run looks like without its ramp and drain; 0 = off")
    ap.add_argument("--no-group-check", action="store_true",
                    help="N > 1: skip rendering the same frames through ONE in-process group handle (pt_create_multi over the N devices, "
                         "peer copies over xGMI) and comparing it bit for bit with the RCCL-gathered image")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="tuning knob of the library (csrc/pt_tuning.hpp) set through pt_debug_set before any renderer exists, e.g. "
                         "--tune no_sphere_grid=1; the library itself reads no environment variables")
    ap.add_argument("--share-gpu", action="store_true",
                    help="debug: all ranks use cuda:0 and rendezvous over gloo (validates the N>1 logic on a 1-GPU box; "
                         "RCCL cannot put two ranks on one device)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_spawn(args))
    preset = CONFIGS[args.config or "C2"]
    scene_name = args.scene or preset[0]
    depth = args.depth if args.depth is not None else preset[1]
    env_name = args.env or preset[2]
    is_preset = (scene_name, depth, env_name) == preset[:3] and args.spp == 1
    baseline_index = preset[3] if is_preset else None

    pkg = graft.load_package()
    if not os.path.exists(pkg.native.LIB_PATH):  # fresh checkout without built artefacts: hipcc is part of the image
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            graft.build()
        else:  # other ranks wait for rank 0's build instead of racing it
            t_wait = time.time()
            while not os.path.exists(pkg.native.LIB_PATH) and time.time() - t_wait < 300:
                time.sleep(1.0)
            time.sleep(2.0)
    from opentk_pathtracer_amd import distributed as D
    for kv in args.tune:
        key, _, val = kv.partition("=")
        pkg.native.debug_set(key, int(val))

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}")
    have = pkg.native.load().pt_device_count()
    if have < args.gpus and not args.share_gpu:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this box has {have} HIP device(s); refusing to report a {args.gpus}-GPU number from fewer GPUs")
    rank, world, local = (D.init_from_env(backend="gloo" if args.share_gpu else None) if args.gpus > 1
                          else (0, 1, int(os.environ.get("LOCAL_RANK", "0"))))
    if args.share_gpu:
        local = 0
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    import torch.distributed as dist

    scene = {"default": pkg.scene.default_scene, "stress256": pkg.scene.stress_scene, "glass": pkg.scene.glass_scene}[scene_name]()
    if os.environ.get("BENCH_STRESS_SPHERES") and scene_name == "stress256":  # tuning runs: the stress scene with fewer spheres
        scene = pkg.scene.stress_scene(int(os.environ["BENCH_STRESS_SPHERES"]))
    cam = pkg.camera.Camera()
    csrc_hash = pkg.native.csrc_hash()
    frames_per_launch = (args.frame_batch or 64) if args.variant == 0 else 1

    def group_handle_check(W, H, basic, warmup, steps, full):
        """The same W + K frames through ONE in-process pt_create_multi handle over all devices (peer copies over xGMI), compared bit for
        bit with the RCCL-gathered image `full` of the one-process-per-GPU path."""
        frames = warmup + steps
        devs = [0] * world if args.share_gpu else list(range(world))  # (--share-gpu: the same code on one device)
        gp = pkg.PathTracer(None, W, H, depth, args.spp, 20.0, 0.14, devices=devs)
        direct = gp.GatherIsDirect
        if not direct and not args.share_gpu:
            gp.Dispose()
            raise RuntimeError("pt_create_multi: no peer access between the devices (hipDeviceCanAccessPeer = 0) — the gather would be staged "
                               "through the host; refusing to measure it (pt_multi_gather_is_direct)")
        gp.SetVariant(args.variant)
        gp.SetFrameBatch(args.frame_batch)
        if env_name == "atmosphere256":
            gp.EnvironmentMap = pkg.AtmosphericScatterer(256, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), gp)
        elif env_name == "sky2048":
            gp.EnvironmentMap = pkg.envmap.synthetic_sky_srgb8(2048)
        else:
            gp.EnvironmentMap = pkg.envmap.synthetic_sky_rgba32f(64)
        gp.UploadScene(scene)
        gp.UploadBasicData(basic)
        for _ in range(warmup):
            gp.Render()
        gp.Synchronize()
        tg = time.perf_counter()
        gp.TimerBegin()
        for _ in range(steps):
            gp.Render()
        gk = gp.TimerEnd()
        gp.Synchronize()
        g_el = time.perf_counter() - tg
        tr = time.perf_counter()
        gimg = gp.Result  # peer copies to device 0 + one device-to-host copy
        g_read = (time.perf_counter() - tr) * 1e3
        same = bool(np.array_equal(gimg.view(np.uint32), full.cpu().numpy().view(np.uint32))) if full is not None else None
        group = {"devices": devs, "frames": frames, "value": round(W * H * args.spp * steps / g_el / 1e6, 2), "unit": "Msamples/s",
                 "ms_per_step": round(g_el * 1e3 / steps, 5), "kernel_ms_slowest_device": round(gk / steps, 5),
                 "read_result_ms": round(g_read, 3), "equals_rccl_gather_bit_for_bit": same, "gather_is_direct": direct,
                 "note": "ONE process, one pt_create_multi handle over all devices, run after the ranks' measurement while they idle"}
        gp.Dispose()
        return group

    def measure(W, H, steps, warmup, clock_warmup_ms, steady_ms=0.0, group_check=False):
        """One workload: create the renderer for this rank's rows of the W x H image, warm up, time `steps` Render() calls
        (barrier + synchronize on both sides, max over ranks), gather once.  -> dict (rank 0) with the raw measurements."""
        basic = pkg.camera.basic_data_ubo(cam, W, H)
        pt = pkg.PathTracer(None, W, H, depth, args.spp, 20.0, 0.14, device=local)
        pt.SetVariant(args.variant)
        pt.SetFrameBatch(args.frame_batch)
        if env_name == "atmosphere256":   # MainWindow.cs:174-175,189
            pt.EnvironmentMap = pkg.AtmosphericScatterer(256, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)
        elif env_name == "sky2048":       # MainWindow.cs:177-187 shape, synthetic content
            pt.EnvironmentMap = pkg.envmap.synthetic_sky_srgb8(2048)
        else:
            pt.EnvironmentMap = pkg.envmap.synthetic_sky_rgba32f(64)
        pt.UploadScene(scene)
        pt.UploadBasicData(basic)
        tile = D.attach_tile(pt, H, rank, world, device=torch.device("cuda", local), band_rows=BAND)
        rows = pt.rows

        def gather_image():
            if args.share_gpu and world > 1:  # gloo cannot gather CUDA tensors: stage through the host
                out = D.present(tile.cpu(), H, rank, world, band_rows=BAND)
                return out.cuda() if out is not None else None
            return D.present(tile, H, rank, world, band_rows=BAND)

        def sync_local():
            pt.Synchronize()
            torch.cuda.synchronize()

        def sync_all():  # local work done on every rank, then all ranks together (the barrier is a GPU collective on RCCL: drain it too)
            sync_local()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()

        # Clock warm-up (NOT counted as steps): the GPU needs ~40 ms of load to leave its idle clock; render for a fixed
        # wall time, then restart the accumulation so that the W warm-up steps and the K timed steps start from frame 0.
        if clock_warmup_ms > 0:
            t_w = time.perf_counter()
            while (time.perf_counter() - t_w) * 1e3 < clock_warmup_ms:
                for _ in range(16):
                    pt.Render()
                pt.Synchronize()
            pt.ResetRenderer()
        for _ in range(warmup):
            pt.Render()
        pt.Synchronize()  # every warm-up frame has reached the tile before torch / RCCL read it
        if world > 1:  # first RCCL call (communicator setup) outside the timed region; also validates the gather
            gather_image()
        sync_all()

        t0 = time.perf_counter()
        pt.TimerBegin()
        for _ in range(steps):
            pt.Render()
        kernel_ms_total = pt.TimerEnd()   # HIP events on the kernel's own stream; also drains it
        sync_local()
        elapsed = time.perf_counter() - t0  # this rank's K steps, from the common start; the job's time is the MAX over ranks (below)
        if world > 1:
            dist.barrier()                   # closing barrier: every rank has finished before anyone gathers

        t1 = time.perf_counter()
        full = gather_image()
        torch.cuda.synchronize()
        present_ms = (time.perf_counter() - t1) * 1e3

        times = torch.tensor([elapsed, kernel_ms_total / 1e3], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
        if world > 1:
            dist.all_reduce(times, op=dist.ReduceOp.MAX)
        elapsed_max, kernel_s_max = (float(v) for v in times.cpu())
        # which devices rendered: every rank reports the PCI bus id of the device it used (distinct ids = distinct GPUs)
        props = torch.cuda.get_device_properties(local)
        me = (local, str(getattr(props, "uuid", "")), str(getattr(props, "pci_bus_id", "")))
        ids = [None] * world
        if world > 1:
            dist.all_gather_object(ids, me)
        else:
            ids = [me]
        devices_used = len(set(ids))

        # ---- steady state of the same workload in the same process (never the metric's value): >= steady_ms of rendering
        steady = None
        if steady_ms > 0:
            sync_all()
            done, t_s = 0, time.perf_counter()
            pt.TimerBegin()
            while True:
                for _ in range(256):
                    pt.Render()
                done += 256
                if (time.perf_counter() - t_s) * 1e3 >= steady_ms and done >= 512:
                    break
            k_ms = pt.TimerEnd()
            sync_local()
            el = time.perf_counter() - t_s
            st = torch.tensor([el, k_ms / 1e3], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
            if world > 1:
                dist.all_reduce(st, op=dist.ReduceOp.MAX)
            el, k_s = (float(v) for v in st.cpu())
            steady = {"steps": done, "ms_per_step": round(el * 1e3 / done, 5), "kernel_ms": round(k_s * 1e3 / done, 5),
                      "value": round(W * H * args.spp * done / el / 1e6, 2), "unit": "Msamples/s",
                      "note": "same workload, same process, right after the timed region; the timed K-step region carries ~0.16 ms of "
                              "clock ramp and drain that a long run amortises"}

        # ---- what the reference's own frame loop gets (MainWindow.cs:40-69: Render -> post-process -> SwapBuffers every frame; never
        # the metric's value): (a) one launch per Render() (`pt_set_frame_batch(1)`), (b) every frame DISPLAYED: render + tone map into a
        # caller-owned device image (pt_present_bind_device_image: the library side of an interop present, nothing crosses PCIe)
        per_frame = displayed = None
        if steady_ms > 0 and world == 1:
            pt.SetFrameBatch(1)
            for _ in range(64):
                pt.Render()
            pt.Synchronize()
            n1, t_p = 512, time.perf_counter()
            pt.TimerBegin()
            for _ in range(n1):
                pt.Render()
            k_ms = pt.TimerEnd()
            sync_local()
            el = time.perf_counter() - t_p
            per_frame = {"steps": n1, "ms_per_step": round(el * 1e3 / n1, 5), "kernel_ms": round(k_ms / n1, 5),
                         "value": round(W * H * args.spp * n1 / el / 1e6, 2), "unit": "Msamples/s",
                         "note": "pt_set_frame_batch(1): every Render() is its own launch (chained single-frame launches), same process"}
            bufs = [torch.zeros((rows, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
            torch.cuda.synchronize()
            for s_, b_ in enumerate(bufs):
                pt.BindPresentImage(s_, b_.data_ptr(), b_.numel())
            seen = [False, False]

            def show(i):
                pt.Render()
                if seen[i & 1]:
                    pt.PresentWait(i & 1)
                pt.PresentAsync(i & 1)
                seen[i & 1] = True
            for i in range(16):
                show(i)
            pt.Synchronize()
            n2, t_d = 400, time.perf_counter()
            for i in range(n2):
                show(i)
            for s_ in range(2):
                pt.PresentWait(s_)
            sync_local()
            el = time.perf_counter() - t_d
            for s_ in range(2):
                pt.BindPresentImage(s_, None)
            displayed = {"frames": n2, "ms_per_displayed_frame": round(el * 1e3 / n2, 5), "value": round(W * H * args.spp * n2 / el / 1e6, 2),
                         "unit": "Msamples/s",
                         "note": "Render(); pt_present_rgba8_async into a bound device image, two slots (a slot is waited for before it is "
                                 "reused): ACES + gamma -> RGBA8 per frame, no host copy"}
            # (c) the reference's REAL frame loop: OnUpdateFrame re-uploads InvView and ViewPos on every focused update, moved or not
            # (MainWindow.cs:131-132: two SubData calls), then OnRenderFrame renders and shows the frame (MainWindow.cs:40-69).  The bytes are
            # unchanged here (a camera at rest, the progressive-accumulation case the metric is about): the library must treat them as no
            # input change — same pipelining, same cached tile masks as (b).  Called through ctypes with prebuilt pointers: the C# host's
            # DllImport costs less than Python's argument marshalling, which is the only thing this leg adds to (b) on the host.
            import ctypes as C_
            cam_blob = (C_.c_ubyte * 144).from_buffer_copy(bytes(basic))
            p_view, p_pos = C_.c_void_p(C_.addressof(cam_blob) + 64), C_.c_void_p(C_.addressof(cam_blob) + 128)
            up = pt._lib.pt_upload_basic_data

            def show_ref(i):
                if up(pt._h, 64, 64, p_view) or up(pt._h, 128, 16, p_pos):
                    raise RuntimeError("pt_upload_basic_data failed")
                show(i)
            for s_, b_ in enumerate(bufs):
                pt.BindPresentImage(s_, b_.data_ptr(), b_.numel())
            seen[0] = seen[1] = False
            for i in range(16):
                show_ref(i)
            pt.Synchronize()
            t_r = time.perf_counter()
            for i in range(n2):
                show_ref(i)
            for s_ in range(2):
                pt.PresentWait(s_)
            sync_local()
            el_r = time.perf_counter() - t_r
            for s_ in range(2):
                pt.BindPresentImage(s_, None)
            # ... and the same loop without a present (Render() only, uploads in between) under the library's automatic batching: the
            # uploads must not break the pipelining (launches_per_step stays 1 / frames_per_launch)
            pt.SetFrameBatch(args.frame_batch)
            for _ in range(128):
                up(pt._h, 64, 64, p_view); up(pt._h, 128, 16, p_pos); pt.Render()
            pt.Synchronize()
            n3, t_u = 960, time.perf_counter()
            for _ in range(n3):
                up(pt._h, 64, 64, p_view); up(pt._h, 128, 16, p_pos); pt.Render()
            sync_local()
            el_u = time.perf_counter() - t_u
            # (d) a MOVING camera (MainWindow.cs:127-132: ProcessInputs moved the camera -> ResetRenderer(); the two uploads now carry new
            # bytes): every frame is frame 0 of a new accumulation, the upload flushes, the cached masks never become valid (the tile pass
            # culls against its own rays)
            cam2 = pkg.camera.Camera(position=(-17.0, 3.6, -8.5), look_x=-31.0, look_y=0.5)
            blob2 = (C_.c_ubyte * 144).from_buffer_copy(bytes(pkg.camera.basic_data_ubo(cam2, W, H)))
            views = [(p_view, p_pos), (C_.c_void_p(C_.addressof(blob2) + 64), C_.c_void_p(C_.addressof(blob2) + 128))]
            for s_, b_ in enumerate(bufs):
                pt.BindPresentImage(s_, b_.data_ptr(), b_.numel())
            seen[0] = seen[1] = False

            def show_moving(i):
                v_, p_ = views[i & 1]
                up(pt._h, 64, 64, v_); up(pt._h, 128, 16, p_)
                pt.ResetRenderer()
                show(i)
            for i in range(16):
                show_moving(i)
            pt.Synchronize()
            n4, t_m = 200, time.perf_counter()
            for i in range(n4):
                show_moving(i)
            for s_ in range(2):
                pt.PresentWait(s_)
            sync_local()
            el_m = time.perf_counter() - t_m
            for s_ in range(2):
                pt.BindPresentImage(s_, None)
            up(pt._h, 64, 64, p_view); up(pt._h, 128, 16, p_pos)
            pt.ResetRenderer()
            reference_loop = {"frames": n2, "ms_per_displayed_frame": round(el_r * 1e3 / n2, 5), "value": round(W * H * args.spp * n2 / el_r / 1e6, 2),
                              "unit": "Msamples/s", "vs_displayed_frame": round(el_r / el, 4),
                              "render_only": {"steps": n3, "ms_per_step": round(el_u * 1e3 / n3, 5), "value": round(W * H * args.spp * n3 / el_u / 1e6, 2)},
                              "moving_camera": {"frames": n4, "ms_per_displayed_frame": round(el_m * 1e3 / n4, 5),
                                                "note": "every frame: NEW InvView / ViewPos bytes, ResetRenderer(), Render(), present (MainWindow.cs:127-132)"},
                              "note": "per frame: pt_upload_basic_data(64, 64) + pt_upload_basic_data(128, 16) with UNCHANGED bytes, Render(), "
                                      "pt_present_rgba8_async into a bound device image = MainWindow.cs:131-132 + :40-69; `render_only` = the same "
                                      "uploads between pipelined Render() calls (no present)"}
            displayed["reference_loop"] = reference_loop
            pt.SetFrameBatch(args.frame_batch)

        # ---- N > 1: the same frames through ONE in-process group handle over the N devices (pt_create_multi: what the reference's
        # single-process host would call; gather by hipMemcpyPeerAsync over xGMI), compared bit for bit with the RCCL gather
        group = None
        if group_check and world > 1 and rank == 0 and (have >= world or args.share_gpu):
            # (never fatal: the ranks' measurement above is the metric; a problem on this second path is REPORTED in the line)
            try:
                group = group_handle_check(W, H, basic, warmup, steps, full)
            except Exception as e:  # noqa: BLE001
                group = {"devices": list(range(world)), "error": f"{type(e).__name__}: {e}", "equals_rccl_gather_bit_for_bit": None}
        if world > 1:
            dist.barrier()
        res = None
        if rank == 0:
            assert full is not None and tuple(full.shape) == (H, W, 4)
            res = {"W": W, "H": H, "rows": rows, "steps": steps, "elapsed": elapsed_max, "kernel_s": kernel_s_max, "present_ms": present_ms,
                   "devices_used": devices_used, "steady": steady, "per_frame": per_frame, "displayed": displayed, "group": group,
                   "checks": {"finite": bool(torch.isfinite(full).all().item()), "alpha_one": bool((full[..., 3] == 1).all().item()),
                              "mean_radiance": round(float(full[..., :3].mean().item()), 5)},
                   "basic": basic, "env_cpu": (pt.ReadEnvironment() if env_name != "sky2048" else None),
                   # hand-over bound (csrc/pt_kernel_common.hpp): what the repair passes of THIS rank's handle did during the whole run —
                   # all zero unless a launch was abandoned (a contended device)
                   "handover": pkg.native.debug_handover_stats(pt._h)}
        del full
        pt.Dispose()
        return res

    if args.strong_4k:
        W, H, scaling = 3840, 2160, "strong"
    elif args.weak:
        (W, H), scaling = weak_image_size(world), "weak"
    else:
        W, H, scaling = 1920, 1080, "strong"  # the metric's image, split N ways (N = 1: the whole image on one GPU)
    m = measure(W, H, args.steps, args.warmup, args.clock_warmup_ms, steady_ms=args.steady_ms, group_check=not args.no_group_check)
    m4k = None
    if world > 1 and not (args.strong_4k or args.weak or args.no_4k):
        # BASELINE configs[3]: ONE 3840x2160 image over the N GPUs, same run, shorter (the clocks are warm)
        steps4k = max(64, (args.steps // 4) // 64 * 64)
        m4k = measure(3840, 2160, steps4k, max(64, (args.warmup // 4) // 64 * 64), 0.0)

    if rank == 0:
        W, H, rows = m["W"], m["H"], m["rows"]
        if args.variant == 0 and not args.frame_batch:  # the library's automatic choice (include/mi355pt.h: pt_set_frame_batch)
            frames_per_launch = 256 if (args.spp == 1 and ((W + 7) // 8) * ((rows + 7) // 8) < 12000) else 64
        samples = W * H * args.spp * args.steps
        ms_per_step = m["elapsed"] * 1e3 / args.steps
        kernel_ms = m["kernel_s"] * 1e3 / args.steps
        algo_bytes = ALGO_BYTES_PER_PIXEL_FRAME * W * rows  # per frame on one GPU (rank 0's rows)
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        wl_key = f"{scene_name}_{W}x{H}_d{depth}_spp{args.spp}_{env_name}_g{world}" + (f"_variant{args.variant}" if args.variant else "") + (f"_fb{args.frame_batch}" if args.frame_batch not in (0, 64) else "") + ("_strong4k" if args.strong_4k else "") + ("_weak" if args.weak and world > 1 else "") + ("_nogrid" if "no_sphere_grid=1" in args.tune else "") + ("_nocarry" if "carry_last=0" in args.tune else "") + ("_gridcarry" if "grid_carry=1" in args.tune else "") + ("_gridcarry2" if "grid_carry=2" in args.tune else "")
        if world == 1:
            where = "one GPU" + (f" (BASELINE configs[{baseline_index}])" if baseline_index is not None and (W, H) == (1920, 1080) else "")
        else:
            where = (f"ONE image row-tiled over {world} GPUs, {rows} rows per GPU in block-cyclic {BAND}-row bands"
                     + (" (BASELINE configs[3])" if (W, H) == (3840, 2160) and baseline_index == 1 else "")
                     + (" (the metric's 1080p image, strong scaling)" if (W, H) == (1920, 1080) and baseline_index == 1 else ""))
        traffic, traffic_note = load_profile_number("traffic.json", wl_key, csrc_hash)
        out = {
            "metric": "Msamples/sec + ms/frame @1080p 8-bounce default scene, 1/2/4/8 GPU",
            "value": round(samples / m["elapsed"] / 1e6, 2), "unit": "Msamples/s", "n_gpus": m["devices_used"], "ranks": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{scene_name} scene ({scene.num_spheres} spheres + {scene.num_cuboids} cuboids), {W}x{H}, "
                                   f"{depth} bounces, {args.spp} spp, progressive accumulate, env {env_name}, {where}",
                       "image": [W, H], "ray_depth": depth, "spp": args.spp, "parallelism": f"rowbands{world}",
                       "kernel_variant": args.variant, "csrc_hash": csrc_hash, **({"tuning": args.tune} if args.tune else {})},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         # measured HBM bytes per LAUNCH (calibrated PMC passes; profiles/traffic.json holds bytes per frame)
                         "traffic": (traffic * frames_per_launch if traffic else None),
                         "traffic_per_frame": traffic,
                         "traffic_ratio_to_algorithmic": (round(traffic / algo_bytes, 3) if traffic else None),
                         "kernel": "pt_integrate_persistent_kernel", "kernel_ms": round(kernel_ms, 5),
                         "launches_per_step": (1.0 / frames_per_launch if frames_per_launch > 1 else
                                               (2 if args.variant == 0 else (args.variant // 10 if 20 <= args.variant < 50 else 1))),
                         "frames_per_launch": frames_per_launch,
                         "algorithmic_bytes_per_launch": algo_bytes * frames_per_launch,
                         # average duration of one launch of the timed region (HIP events on the library's own streams; frac =
                         # algorithmic_bytes_per_launch / avg_launch_ns_timed_region / peak can be recomputed from this line alone)
                         "avg_launch_ns_timed_region": round(kernel_ms * 1e6 * frames_per_launch),
                         "launches_in_timed_region": round(args.steps / frames_per_launch, 3),
                         "provenance": {
                             "achieved": {"measured_in_this_run": True, "how": "HIP events around the timed region on the library's streams"},
                             "frac": {"measured_in_this_run": True},
                             "kernel_ms": {"measured_in_this_run": True},
                             "avg_launch_ns_timed_region": {"measured_in_this_run": True},
                             "traffic": {"measured_in_this_run": False, "source": "profiles/traffic.json" if traffic else None,
                                         "csrc_hash": csrc_hash if traffic else None,
                                         "how": "rocprofv3 PMC passes of tools/round_profiles.sh (separate runs of this command on the same kernel "
                                                "sources: used only when the entry's csrc_hash equals this library's)"},
                             "valu_issue": {"measured_in_this_run": False, "source": "profiles/valu_insts.json", "csrc_hash": csrc_hash,
                                            "how": "SQ_INSTS_VALU of a rocprofv3 PMC pass (same rule); the RATE divides it by this run's kernel_ms"}},
                         "note": "32 B/pixel/frame (float4 load + store of the accumulation image); kernel_ms = GPU time per "
                                 "step (= frame) from HIP events on the library's streams. The default kernel pipelines up to "
                                 "frames_per_launch consecutive frames inside ONE launch (achieved = algorithmic_bytes_per_launch "
                                 "/ (frames_per_launch x kernel_ms)). Consecutive launches OVERLAP pairwise on two streams (launch "
                                 "chaining: a launch is enqueued as soon as its predecessor is resident and starts in the wavefront slots that "
                                 "one's drain frees; rocprofv3 stamps a launch's start when the command processor begins the dispatch, so its "
                                 "duration includes the wait beside the predecessor), so rocprofv3's per-launch durations add up to ~2x the "
                                 "elapsed time: profiles/<round>/*_summary.json report their sum AND the union of the launch intervals per "
                                 "frame; the union is what kernel_ms measures. "
                                 "--frame-batch 1 launches every frame on its own (2 overlapping row-stripe launches). "
                                 "The path is fp32-VALU bound, see `valu_issue`"},
            "present_ms": round(m["present_ms"], 3),
            "steady": m["steady"],
            "per_frame_launch": m["per_frame"],
            "displayed_frame": m["displayed"],
            "reference_loop": (m["displayed"] or {}).pop("reference_loop", None),
            "clock_warmup_ms": args.clock_warmup_ms,
            "checks": m["checks"],
            "handover_bound": m["handover"],
        }
        if traffic_note:
            out["roofline"]["traffic_note"] = traffic_note
        vi, vi_note = load_profile_number("valu_insts.json", wl_key, csrc_hash)
        if vi:
            rate = vi / (kernel_ms * 1e-3) / 1e9
            out["roofline"]["valu_issue"] = {"wave_insts_per_step": vi, "achieved": round(rate, 1), "peak": round(VALU_ISSUE_PEAK_GINST, 1),
                                              "unit": "G wave-instructions/s", "frac": round(rate / VALU_ISSUE_PEAK_GINST, 4),
                                              "note": "the binding roof: SQ_INSTS_VALU per step (rocprofv3 PMC pass committed under "
                                                      "profiles/, same kernel sources: csrc_hash) / kernel time, against one wave64 VALU "
                                                      "instruction per 2 clocks per SIMD (the rate behind the 157.3 TFLOP/s FP32 vector "
                                                      "peak).  Measured on whole-machine launches (tools/ubench3.hip, profiles/r04/ubench3_*.log): "
                                                      "a VALU stream sustains one instruction per ~2.4 clocks at 6 wavefronts per SIMD, and a "
                                                      "scalar / branch instruction of the same wavefronts takes an issue slot of about the same "
                                                      "length - `all_issue` counts them too"}
            other = load_profile_entry("valu_insts.json", wl_key, csrc_hash).get("other")
            if other:
                tot = vi + sum(other.values())
                out["roofline"]["valu_issue"]["all_issue"] = {"wave_insts_per_step": tot, **other,
                                                               "achieved": round(tot / (kernel_ms * 1e-3) / 1e9, 1),
                                                               "frac_of_valu_peak": round(tot / (kernel_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK_GINST, 4)}
        elif vi_note:
            out["roofline"]["valu_issue"] = None
            out["roofline"]["valu_issue_note"] = vi_note
        if m["group"] is not None:
            out["in_process_group"] = m["group"]
        if args.share_gpu:
            out["n_gpus"] = 1  # debug mode: all ranks shared cuda:0
        if m4k is not None:
            k4 = m4k["kernel_s"] * 1e3 / m4k["steps"]
            a4 = ALGO_BYTES_PER_PIXEL_FRAME * 3840 * m4k["rows"] / (k4 * 1e-3) / 1e9
            out["configs3_4k"] = {
                "workload": f"{scene_name} scene, 3840x2160, {depth} bounces, {args.spp} spp, ONE image row-tiled over {world} GPUs "
                            f"({m4k['rows']} rows per GPU, block-cyclic {BAND}-row bands), RCCL gather at present (BASELINE configs[3])",
                "value": round(3840 * 2160 * args.spp * m4k["steps"] / m4k["elapsed"] / 1e6, 2), "unit": "Msamples/s",
                "ms_per_step": round(m4k["elapsed"] * 1e3 / m4k["steps"], 5), "steps": m4k["steps"], "scaling": "strong",
                "kernel_ms": round(k4, 5), "present_ms": round(m4k["present_ms"], 3), "checks": m4k["checks"],
                "roofline_frac_hbm": round(a4 / HBM_PEAK_GBS, 5)}
        mean_bounces = miss_fraction = None
        if world == 1 and not args.no_cpu_baseline:
            env_cpu = m["env_cpu"] if env_name != "sky2048" else pkg.envmap.synthetic_sky_srgb8(2048)
            wl_name = {("default", 8, "atmosphere256"): "C2_default_1080p_d8_atmo"}.get((scene_name, depth, env_name)) if (W, H) == (1920, 1080) and args.spp == 1 else None
            out["cpu_baseline"], mean_bounces, miss_fraction = cpu_baseline(pkg, scene, m["basic"], env_cpu, W, H, depth, args.spp, wl_name)
            # SURVEY section 8d's secondary byte figure: + 4 environment texels per path that ends in a miss (64 B for the RGBA32F cubes,
            # 16 B for sRGB8) — with a 100 MB cube (sky2048) those taps are not cache-resident, and the measured traffic shows them
            tap_bytes = (16 if env_name == "sky2048" else 64) * miss_fraction * args.spp
            sec = algo_bytes + tap_bytes * W * rows
            out["roofline"]["secondary"] = {
                "algorithmic_bytes_per_step_with_env_taps": round(sec), "paths_ending_in_a_miss_per_sample": round(miss_fraction, 4),
                "bytes_per_miss": 16 if env_name == "sky2048" else 64, "achieved": round(sec / (kernel_ms * 1e-3) / 1e9, 2), "unit": "GB/s",
                "frac": round(sec / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "traffic_ratio": (round(traffic / sec, 3) if traffic else None),
                "measured_in_this_run": True,
                "note": "SURVEY 8d's optional figure: 32 B per pixel and frame + 4 environment texels per path that ends in a miss (counted by "
                        "the oracle on one frame of this workload, this run)"}
        if mean_bounces is not None:
            fl = flops_per_sample(mean_bounces, scene.num_spheres, scene.num_cuboids)
            tf = fl * (W * rows * args.spp) / (kernel_ms * 1e-3) / 1e12
            out["roofline"]["valu"] = {"model_flops_per_sample": round(fl, 1), "mean_bounces": round(mean_bounces, 4),
                                       "achieved": round(tf, 3), "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s",
                                       "frac": round(tf / FP32_VALU_PEAK_TF, 4)}
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
