#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: Msamples/s of the path-tracing integrator on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one PathTracer.Render() — one pass of the integrator over the whole image with the inputs (scene
UBO, camera UBO, environment cube, accumulation image) already resident in HBM; frames accumulate progressively
exactly as in the reference (src/Render/PathTracer.cs:114-123).  The library launches consecutive Render() calls as one
kernel that pipelines up to --frame-batch frames; every frame is computed in full inside the
timed region, which ends with pt_timer_end + pt_synchronize.  Defaults: 960 steps after 320 warm-up steps; before the
warm-up steps the GPU is kept busy for --clock-warmup-ms (80 ms, reported in the JSON line) because it needs ~40 ms of
load to reach its steady clock — that time is neither a step nor timed, and the accumulation restarts at frame 0 after it.

N = 1 runs BASELINE.json configs[1]: default scene (48 spheres + 7 cuboids), 1920x1080, 8 bounces, 1 spp, the
reference's default environment (256^2 RGBA32F atmosphere cube, computed by the atmosphere kernel).
N > 1 is WEAK scaling: the 16:9 image grows to ~N x 2.07 Mpixel (N=4 is BASELINE configs[3]'s 3840x2160) and is
tiled across ranks in block-cyclic 16-row bands (balanced: floor rows cost ~2x sky rows), no data-path collective; the
RCCL gather happens only at present time and is timed separately (`present_ms`).  value = all pixels x spp x steps / max-over-ranks wall time.

The JSON line also carries
  roofline     : the integrator kernel against the HBM roof, from ALGORITHMIC bytes (32 B per pixel per frame: one
                 float4 load + one float4 store of the accumulation image) and the kernel's average duration measured
                 with HIP events on the stream it runs on;  `valu` gives the roof that actually binds (fp32 vector ALU);
  cpu_baseline : the C restatement (oracle/, "port") timed on this box's host cores on a bounded sample of the same
                 workload — rank 0, N = 1 only.  The oracle is used here ONLY as the thing timed, never by the GPU path.
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


def image_size(n_gpus: int) -> tuple[int, int]:
    if n_gpus in WEAK_SIZES:
        return WEAK_SIZES[n_gpus]
    w = int(round(1920 * n_gpus ** 0.5 / 16)) * 16
    return w, w * 9 // 16


def flops_per_sample(mean_bounces: float, ns: int, nc: int) -> float:
    """SURVEY.md section 8d model: 100 + sum over bounces of (17 Ns + 24 Nc + 120)."""
    return 100.0 + mean_bounces * (17.0 * ns + 24.0 * nc + 120.0)


def _load_profile_number(file_name: str, workload_key: str):
    p = os.path.join(ROOT, "profiles", file_name)
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(workload_key)
        except Exception:
            return None
    return None


def load_traffic(workload_key: str):
    """Measured HBM bytes per step from the rocprofv3 PMC passes committed under profiles/ (or None)."""
    return _load_profile_number("traffic.json", workload_key)


def load_valu_insts(workload_key: str):
    """Measured VALU wave-instructions per step (SQ_INSTS_VALU, same PMC passes) or None."""
    return _load_profile_number("valu_insts.json", workload_key)


# VALU issue roof: the 157.3 TFLOP/s FP32 vector peak = 1024 SIMDs x 32 lanes x 2 flops x 2.4 GHz, i.e. at best one
# wave64 instruction per 2 clocks per SIMD.  (A dependence-free v_fma_f32 stream sustains one per 2.63 clocks with
# 4 wavefronts per SIMD, tools/ubench2.hip; the integrator's mix — with instructions that retire early under an empty
# EXEC mask — runs at one per ~2.4.)
VALU_ISSUE_PEAK_GINST = 1024 * 2.4 / 2.0


def cpu_baseline(pkg, scene, basic, env, width, height, depth, spp):
    """Time the oracle ("port") on the host cores on a bounded sample: whole frames of the same workload until
    ~10 s have elapsed (at least one frame)."""
    oracle = graft.load_oracle().Oracle()
    cores = os.cpu_count() or 1
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
    _, st = oracle.render(width, height, basic, objs, env, y0=0, rows=height, threads=cores, want_stats=True, **kw)
    return {
        "value": round(width * height * spp * frames / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"{frames} full frame(s) of the same workload ({width}x{height}, {depth} bounces, {spp} spp) in {dt:.1f} s, "
                  f"oracle/pt_oracle.c row-parallel on {cores} threads",
    }, st["bounces"] / max(1, st["samples"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=960)
    ap.add_argument("--warmup", type=int, default=320)
    ap.add_argument("--scene", default="default", choices=["default", "stress256", "glass"])
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--env", default="atmosphere256", choices=["atmosphere256", "sky2048", "sky64"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--frame-batch", type=int, default=64,
                    help="frames one launch may pipeline (pt_set_frame_batch; 1 = one launch per Render())")
    ap.add_argument("--clock-warmup-ms", type=float, default=80.0,
                    help="wall time of untimed rendering before the W warm-up steps (GPU clock ramp); 0 = none")
    ap.add_argument("--strong-4k", action="store_true",
                    help="BASELINE configs[3]: ONE 3840x2160 image row-tiled over the N GPUs (strong scaling) instead of the "
                         "weak-scaling image that grows with N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--share-gpu", action="store_true",
                    help="debug: all ranks use cuda:0 and rendezvous over gloo (validates the N>1 logic on a 1-GPU box; "
                         "RCCL cannot put two ranks on one device)")
    args = ap.parse_args()

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

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus and world_env > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}")
    rank, world, local = (D.init_from_env(backend="gloo" if args.share_gpu else None) if args.gpus > 1
                          else (0, 1, int(os.environ.get("LOCAL_RANK", "0"))))
    if args.share_gpu:
        local = 0
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    import torch.distributed as dist

    W, H = (3840, 2160) if args.strong_4k else image_size(world)  # --strong-4k: BASELINE configs[3], one 4K image over N GPUs
    scene = {"default": pkg.scene.default_scene, "stress256": pkg.scene.stress_scene, "glass": pkg.scene.glass_scene}[args.scene]()
    cam = pkg.camera.Camera()
    basic = pkg.camera.basic_data_ubo(cam, W, H)

    pt = pkg.PathTracer(None, W, H, args.depth, args.spp, 20.0, 0.14, device=local)
    pt.SetVariant(args.variant)
    pt.SetFrameBatch(args.frame_batch)
    # frames one launch renders: consecutive Render() calls of the default spp=1 kernel are pipelined inside one launch
    frames_per_launch = args.frame_batch if args.variant == 0 else 1
    if args.env == "atmosphere256":   # MainWindow.cs:174-175,189
        pt.EnvironmentMap = pkg.AtmosphericScatterer(256, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)
    elif args.env == "sky2048":       # MainWindow.cs:177-187 shape, synthetic content
        pt.EnvironmentMap = pkg.envmap.synthetic_sky_srgb8(2048)
    else:
        pt.EnvironmentMap = pkg.envmap.synthetic_sky_rgba32f(64)
    pt.UploadScene(scene)
    pt.UploadBasicData(basic)
    BAND = 16  # block-cyclic 16-row bands across ranks: row cost varies ~2x between sky and floor rows
    tile = D.attach_tile(pt, H, rank, world, device=torch.device("cuda", local), band_rows=BAND)
    rows = pt.rows

    def gather_image():
        if args.share_gpu and world > 1:  # gloo cannot gather CUDA tensors: stage through the host
            out = D.present(tile.cpu(), H, rank, world, band_rows=BAND)
            return out.cuda() if out is not None else None
        return D.present(tile, H, rank, world, band_rows=BAND)

    def sync_all():
        pt.Synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # Clock warm-up (NOT counted as steps): the GPU needs ~40 ms of load to leave its idle clock; render for a fixed wall
    # time, then restart the accumulation so that the W warm-up steps and the K timed steps start from frame 0.
    if args.clock_warmup_ms > 0:
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < args.clock_warmup_ms:
            for _ in range(16):
                pt.Render()
            pt.Synchronize()
        pt.ResetRenderer()
    for _ in range(args.warmup):
        pt.Render()
    if world > 1:  # first RCCL call (communicator setup) outside the timed region; also validates the gather
        gather_image()
    sync_all()

    t0 = time.perf_counter()
    pt.TimerBegin()
    for _ in range(args.steps):
        pt.Render()
    kernel_ms_total = pt.TimerEnd()   # HIP events on the kernel's own stream; also drains it
    sync_all()
    elapsed = time.perf_counter() - t0

    t1 = time.perf_counter()
    full = gather_image()
    torch.cuda.synchronize()
    present_ms = (time.perf_counter() - t1) * 1e3

    times = torch.tensor([elapsed, kernel_ms_total / 1e3], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
    if world > 1:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    elapsed_max, kernel_s_max = (float(v) for v in times.cpu())

    if rank == 0:
        assert full is not None and tuple(full.shape) == (H, W, 4)
        checks = {"finite": bool(torch.isfinite(full).all().item()), "alpha_one": bool((full[..., 3] == 1).all().item()),
                  "mean_radiance": round(float(full[..., :3].mean().item()), 5)}
        samples = W * H * args.spp * args.steps
        ms_per_step = elapsed_max * 1e3 / args.steps
        kernel_ms = kernel_s_max * 1e3 / args.steps
        algo_bytes = ALGO_BYTES_PER_PIXEL_FRAME * W * rows  # per launch on one GPU (rank 0's row block)
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        wl_key = f"{args.scene}_{W}x{H}_d{args.depth}_spp{args.spp}_{args.env}_g{world}" + (f"_variant{args.variant}" if args.variant else "") + (f"_fb{args.frame_batch}" if args.frame_batch != 64 else "") + ("_strong4k" if args.strong_4k else "")
        out = {
            "metric": "Msamples/sec + ms/frame @1080p 8-bounce default scene, 1/2/4/8 GPU",
            "value": round(samples / elapsed_max / 1e6, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong" if args.strong_4k else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} scene ({scene.num_spheres} spheres + {scene.num_cuboids} cuboids), {W}x{H}, "
                                   f"{args.depth} bounces, {args.spp} spp, progressive accumulate, env {args.env}, "
                                   + (f"{rows} rows per GPU in block-cyclic {BAND}-row bands" if world > 1
                                      else "one GPU (BASELINE configs[1])"),
                       "image": [W, H], "ray_depth": args.depth, "spp": args.spp, "parallelism": f"rowbands{world}",
                       "kernel_variant": args.variant},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         # measured HBM bytes per LAUNCH (calibrated PMC passes, profiles/traffic.json holds bytes per frame)
                         "traffic": (load_traffic(wl_key) * frames_per_launch if load_traffic(wl_key) else None),
                         "traffic_per_frame": load_traffic(wl_key),
                         "kernel": "pt_integrate_persistent_kernel", "kernel_ms": round(kernel_ms, 5),
                         "launches_per_step": (1.0 / frames_per_launch if frames_per_launch > 1 else
                                               (2 if args.variant == 0 else (args.variant // 10 if 20 <= args.variant < 50 else 1))),
                         "frames_per_launch": frames_per_launch,
                         "algorithmic_bytes_per_launch": algo_bytes * frames_per_launch,
                         "note": "32 B/pixel/frame (float4 load + store of the accumulation image); kernel_ms = GPU time per "
                                 "step (= frame) from HIP events on the library's streams. The default kernel pipelines up to "
                                 "frames_per_launch consecutive frames inside ONE launch (rocprofv3's average launch duration "
                                 "= frames_per_launch x kernel_ms; achieved = algorithmic_bytes_per_launch / that duration). "
                                 "--frame-batch 1 launches every frame on its own (2 overlapping row-stripe launches). "
                                 "The path is fp32-VALU bound, see `valu_issue`"},
            "present_ms": round(present_ms, 3),
            "clock_warmup_ms": args.clock_warmup_ms,
            "checks": checks,
        }
        vi = load_valu_insts(wl_key)
        if vi:
            rate = vi / (kernel_ms * 1e-3) / 1e9
            out["roofline"]["valu_issue"] = {"wave_insts_per_step": vi, "achieved": round(rate, 1), "peak": round(VALU_ISSUE_PEAK_GINST, 1),
                                              "unit": "G wave-instructions/s", "frac": round(rate / VALU_ISSUE_PEAK_GINST, 4),
                                              "note": "the binding roof: SQ_INSTS_VALU per step (rocprofv3 PMC pass committed under "
                                                      "profiles/) / kernel time, against one wave64 VALU instruction per 2 clocks "
                                                      "per SIMD (the rate behind the 157.3 TFLOP/s FP32 vector peak); a pure "
                                                      "v_fma_f32 stream sustains 2.63 clocks (tools/ubench2.hip)"}
        mean_bounces = None
        if world == 1 and not args.no_cpu_baseline:
            env_cpu = pt.ReadEnvironment() if args.env != "sky2048" else pkg.envmap.synthetic_sky_srgb8(2048)
            out["cpu_baseline"], mean_bounces = cpu_baseline(pkg, scene, basic, env_cpu, W, H, args.depth, args.spp)
        if mean_bounces is not None:
            fl = flops_per_sample(mean_bounces, scene.num_spheres, scene.num_cuboids)
            tf = fl * (W * rows * args.spp) / (kernel_ms * 1e-3) / 1e12
            out["roofline"]["valu"] = {"model_flops_per_sample": round(fl, 1), "mean_bounces": round(mean_bounces, 4),
                                       "achieved": round(tf, 3), "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s",
                                       "frac": round(tf / FP32_VALU_PEAK_TF, 4)}
        print(json.dumps(out), flush=True)

    pt.Dispose()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
