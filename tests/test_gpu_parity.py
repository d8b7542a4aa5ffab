"""Parity tests proper: the HIP path, called through the C ABI (libmi355pt.so), against
  (1) the CPU oracle on the same inputs          -> BIT-EXACT (both implement the pt-f32 contract),
  (2) the committed fixtures of the reference GLSL run on llvmpipe -> tolerances of tests/tolerances.py,
  (3) size-independent properties at BASELINE.json's full sizes (tiled == untiled, resume, determinism).
Run with `pytest -m gpu` on an MI355X.  Nothing here reads /root/reference.
"""
import ctypes as C

import numpy as np
import pytest

import configs
import fixtures
import tolerances as tol

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_exact(got, want, what):
    same = (bits(got) == bits(want))
    if not same.all():
        bad = ~same.reshape(same.shape[0], -1).all(-1) if same.ndim == 2 else ~same.all(-1)
        idx = np.argwhere(bad)[:5]
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} pixels differ from the oracle; first at {idx.tolist()}")


def hip_render(pkg, w: configs.Workload, frames=None, variant=0, tile=None):
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, w.spp, w.focal_length, w.aperture)
    pt.SetVariant(variant)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    if tile:
        pt.SetTile(*tile)
    for _ in range(frames if frames is not None else w.frames):
        pt.Render()
    out = pt.Result
    pt.Dispose()
    return out


def oracle_render(oracle, w: configs.Workload, frames=None, **extra):
    sc, basic, objs, env, kw = configs.inputs(w)
    return oracle.render(w.width, w.height, basic, objs, env, num_frames=frames if frames is not None else w.frames,
                         **kw, **extra)


# ------------------------------------------------------------------------------------------------ (1) HIP == oracle
@pytest.mark.parametrize("w", configs.SMALL_FRAMES + configs.ENV_ONLY, ids=lambda w: w.name)
def test_hip_equals_oracle_bit_exact(pkg, native_lib, oracle, w):
    assert_bit_exact(hip_render(pkg, w), oracle_render(oracle, w), w.name)


@pytest.mark.parametrize("variant", [1, 2, 3, 6, 10, 12], ids=lambda v: f"variant{v}")
@pytest.mark.parametrize("w", [configs.SMALL_FRAMES[1], configs.SMALL_FRAMES[4], configs.SMALL_FRAMES[8],
                               configs.SMALL_FRAMES[9]], ids=lambda w: w.name)
def test_every_kernel_variant_is_bit_exact(pkg, native_lib, oracle, w, variant):
    """The tile-per-wave kernel, the wave-pool kernels and the persistent-queue kernel (pt_set_variant) all produce
    the oracle's bits: path regeneration re-orders work across lanes, never the arithmetic of a pixel."""
    assert_bit_exact(hip_render(pkg, w, variant=variant), oracle_render(oracle, w), f"{w.name} variant {variant}")


@pytest.mark.parametrize("w", [
    configs.Workload("one_pixel", "default", 1, 1, 8, "sky_f32_32"),
    configs.Workload("one_row", "default", 77, 1, 8, "sky_f32_32"),
    configs.Workload("one_column", "default", 1, 53, 8, "sky_f32_32"),
    configs.Workload("depth0", "default", 40, 24, 0, "sky_f32_32"),
    configs.Workload("depth1", "default", 40, 24, 1, "sky_f32_32"),
    configs.Workload("depth50_spp10", "default", 40, 24, 50, "sky_f32_32", spp=10),
    configs.Workload("no_aperture", "default", 64, 36, 8, "sky_f32_32", aperture=0.0),
    configs.Workload("wide_aperture", "default", 64, 36, 8, "sky_f32_32", aperture=3.0, focal_length=5.0),
    configs.Workload("look_up", "default", 64, 36, 8, "sky_f32_32", look=(90.0, 89.0)),
    configs.Workload("inside_glass", "edge", 64, 36, 24, "sky_srgb_32", frames=2),
    configs.Workload("stress_full_ubo", "stress256", 96, 54, 8, "tiny_4"),
], ids=lambda w: w.name)
def test_edge_shapes_and_parameters(pkg, native_lib, oracle, w):
    """ragged / degenerate sizes (the reference relies on GL's out-of-bounds image semantics, compute.glsl:104,129;
    the kernel guards explicitly), GUI parameter extremes (Gui.cs:39-73: SPP 1-10, depth 1-50)."""
    assert_bit_exact(hip_render(pkg, w), oracle_render(oracle, w), w.name)


def test_full_ubo_max_objects(pkg, native_lib, oracle):
    """256 spheres AND 64 cuboids: the whole GameObjectsUBO (MainWindow.cs:17)."""
    s = pkg.scene
    sc = s.stress_scene(256)
    rng = np.random.RandomState(5)
    sc.cuboids = []
    for i in range(64):
        c = rng.uniform([-18, -11, -21], [18, 11, 1])
        sc.cuboids.append(s.Cuboid(c.astype(np.float32), rng.uniform(0.2, 1.5, 3).astype(np.float32), i,
                                   s.Material(albedo=rng.rand(3), specular_chance=rng.rand() * 0.5,
                                              specular_roughness=rng.rand(), ior=1 + rng.rand(),
                                              refraction_chance=rng.rand() * 0.5)))
    cam = pkg.camera.Camera()
    W, H = 96, 54
    basic = pkg.camera.basic_data_ubo(cam, W, H)
    env = configs.load_env("sky_f32_32")
    pt = pkg.PathTracer(env, W, H, 8, 1, 20.0, 0.14)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    pt.Render()
    want = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_spheres=256, num_cuboids=64, ray_depth=8)
    assert_bit_exact(pt.Result, want, "256 spheres + 64 cuboids")


def _random_cameras(n, seed):
    rng = np.random.RandomState(seed)
    cams = []
    for i in range(n):
        pos = tuple(float(v) for v in rng.uniform([-17.5, -10.5, -20.5], [17.5, 10.5, 0.5]))
        look = (float(rng.uniform(-180, 180)), float(rng.uniform(-80, 80)))
        cams.append((pos, look))
    return cams


@pytest.mark.parametrize("scene,size,aperture,focal", [
    ("stress256", (128, 72), 0.14, 20.0),   # 256 spheres: every tile culls against all four 64-sphere mask words
    ("stress256", (24, 16), 0.14, 20.0),    # a tile spans a large part of the field of view: wide ray bundles
    ("default", (8, 8), 0.14, 20.0),        # one tile = the whole image
    ("default", (96, 54), 8.0, 2.0),        # lens far larger than the scene features: ray origins spread over metres
    ("edge", (96, 54), 1.0, 6.0),           # cameras inside / next to glass
    ("randmat", (96, 54), 0.0, 20.0),       # pinhole: zero-radius apex
], ids=lambda v: str(v))
def test_tile_pass_culling_is_conservative(pkg, native_lib, oracle, scene, size, aperture, focal):
    """The spp = 1 kernels run each tile's first bounce with the spheres culled against the tile's ray bundle
    (cull_spheres, pt_device.hpp).  A sphere removed by mistake would change pixels, so random cameras all over
    (and inside) the scene must still match the brute-force oracle bit for bit."""
    for k, (pos, look) in enumerate(_random_cameras(6, sum(map(ord, scene)) + size[0])):
        w = configs.Workload(f"cull_{scene}_{k}", scene, size[0], size[1], 6, "sky_f32_32", aperture=aperture,
                             focal_length=focal, look=look, position=pos)
        assert_bit_exact(hip_render(pkg, w), oracle_render(oracle, w), f"{w.name} pos={pos} look={look}")


@pytest.mark.parametrize("scene,size,aperture,focal", [
    ("stress256", (128, 72), 0.14, 20.0), ("stress256", (24, 16), 0.14, 20.0), ("default", (8, 8), 0.14, 20.0),
    ("default", (96, 54), 8.0, 2.0), ("default", (96, 54), 0.5, 0.3), ("edge", (96, 54), 1.0, 6.0), ("randmat", (96, 54), 0.0, 20.0),
], ids=lambda v: str(v))
def test_cached_tile_masks_are_conservative(pkg, native_lib, oracle, scene, size, aperture, focal):
    """Round 4: once the camera has been left alone for two launches, the tile pass takes its sphere masks from a per-tile cache that
    pt_tile_masks_kernel fills from an ANALYTIC bound of every ray the tile can cast (any jitter, any lens sample: tile_cone,
    pt_device.hpp).  Random cameras all over (and inside) the scenes, lenses from pinhole to larger than the focal length: five launches
    of four frames (the last three with cached masks), the camera then moved (masks stale, per-tile culling again, rebuilt two launches
    later) — bit for bit the brute-force oracle."""
    cams = _random_cameras(5, 7 * sum(map(ord, scene)) + size[0])
    for k, (pos, look) in enumerate(cams):
        w = configs.Workload(f"masks_{scene}_{k}", scene, size[0], size[1], 6, "sky_f32_32", aperture=aperture, focal_length=focal,
                             look=look, position=pos)
        sc, basic, objs, env, kw = configs.inputs(w)
        pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
        pt.UploadScene(sc)
        pt.UploadBasicData(basic)
        for _ in range(5):
            for _ in range(4):
                pt.Render()
            pt.Synchronize()
        want = oracle.render(w.width, w.height, basic, objs, env, num_frames=20, **kw)
        assert_bit_exact(pt.Result, want, f"{w.name} pos={pos} look={look} (cached masks)")
        # move the camera: the accumulation restarts, the masks are stale until two launches later
        pos2, look2 = cams[(k + 1) % len(cams)]
        w2 = configs.Workload(f"masks_{scene}_{k}b", scene, size[0], size[1], 6, "sky_f32_32", aperture=aperture, focal_length=focal,
                              look=look2, position=pos2)
        _, basic2, _, _, _ = configs.inputs(w2)
        pt.UploadBasicData(basic2)
        pt.ResetRenderer()
        for _ in range(4):
            for _ in range(3):
                pt.Render()
            pt.Synchronize()
        want2 = oracle.render(w.width, w.height, basic2, objs, env, num_frames=12, **kw)
        assert_bit_exact(pt.Result, want2, f"{w2.name} pos={pos2} look={look2} (after a camera move)")
        pt.Dispose()


def test_cached_tile_masks_follow_cuboid_edits(pkg, native_lib, oracle):
    """The cached masks hold a CUBOID mask too: a cuboid that is moved, resized or added through partial uploads of the GameObjectsUBO
    (Cuboid.cs:21 buffer offsets, all beyond the Spheres[] array) must drop them.  (Found by review at the end of round 4: only uploads
    that touched the sphere array did; a cuboid moved into a tile's view stayed culled there.)"""
    import copy
    w = configs.Workload("masks_cuboid_edit", "default", 128, 72, 6, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    for _ in range(5):
        for _ in range(3):
            pt.Render()
        pt.Synchronize()
    before = oracle.render(w.width, w.height, basic, objs, env, num_frames=15, **kw)
    assert_bit_exact(pt.Result, before, "before the edit (cached masks)")
    # 1: the last cuboid becomes a slab right in front of the camera (one SubData at its own offset, like the reference's editor)
    cam = pkg.camera.Camera(position=w.position, look_x=w.look[0], look_y=w.look[1])
    sc2 = copy.deepcopy(sc)
    c = sc2.cuboids[-1]
    c.position = (np.asarray(w.position, dtype=np.float32) + np.asarray(cam.view_dir if hasattr(cam, "view_dir") else (0.0, 0.0, -1.0), dtype=np.float32) * np.float32(6.0)).astype(np.float32)
    c.dimensions = np.asarray((3.0, 2.0, 0.5), dtype=np.float32)
    d = c.gpu_data()
    pt.GameObjectsUBO.SubData(c.buffer_offset, d.nbytes, d)
    pt.ResetRenderer()
    for _ in range(4):
        for _ in range(3):
            pt.Render()
        pt.Synchronize()
    after = oracle.render(w.width, w.height, basic, sc2.ubo_bytes(), env, num_frames=12, **kw)
    assert np.mean(np.abs(after[..., :3] - oracle.render(w.width, w.height, basic, objs, env, num_frames=12, **kw)[..., :3]) > 1e-3) > 0.01, "the edit must be visible"
    assert_bit_exact(pt.Result, after, "after moving a cuboid (partial upload beyond the sphere array)")
    # 2: one more cuboid (count change only; its bytes were uploaded before the count was raised)
    sc3 = copy.deepcopy(sc2)
    extra = copy.deepcopy(sc2.cuboids[-1])
    extra.instance = len(sc3.cuboids)
    extra.position = (np.asarray(c.position, dtype=np.float32) + np.asarray((-4.0, 1.5, 0.0), dtype=np.float32)).astype(np.float32)
    extra.dimensions = np.asarray((1.5, 1.5, 1.5), dtype=np.float32)
    sc3.cuboids.append(extra)
    d = extra.gpu_data()
    pt.GameObjectsUBO.SubData(extra.buffer_offset, d.nbytes, d)
    for _ in range(3):  # (masks cached again for the old count)
        for _ in range(3):
            pt.Render()
        pt.Synchronize()
    pt._numCuboids = sc3.num_cuboids
    pt._push_params()
    pt.ResetRenderer()
    for _ in range(4):
        for _ in range(3):
            pt.Render()
        pt.Synchronize()
    kw3 = dict(kw, num_cuboids=sc3.num_cuboids)
    assert_bit_exact(pt.Result, oracle.render(w.width, w.height, basic, sc3.ubo_bytes(), env, num_frames=12, **kw3), "after adding a cuboid")
    pt.Dispose()


def test_cached_tile_masks_in_the_multisample_kernel(pkg, native_lib, oracle):
    """The spp > 1 batch-pass kernel takes the cached masks for its fresh-tile passes (sample 0); continuations keep the per-bundle
    culling.  Small images only reach that kernel with the tuning knob batch_pass_min_tiles = 0."""
    pkg.native.debug_set("batch_pass_min_tiles", 0)
    try:
        for k, (pos, look) in enumerate(_random_cameras(4, 4242)):
            for scene in ("default", "stress256"):
                w = configs.Workload(f"masks_ms_{scene}_{k}", scene, 96, 54, 6, "sky_f32_32", spp=3, look=look, position=pos)
                sc, basic, objs, env, kw = configs.inputs(w)
                pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, w.spp, w.focal_length, w.aperture)
                pt.UploadScene(sc)
                pt.UploadBasicData(basic)
                for _ in range(5):
                    for _ in range(3):
                        pt.Render()
                    pt.Synchronize()
                want = oracle.render(w.width, w.height, basic, objs, env, num_frames=15, **kw)
                assert_bit_exact(pt.Result, want, f"{w.name} pos={pos} look={look}")
                pt.Dispose()
    finally:
        pkg.native.debug_set("batch_pass_min_tiles", 16384)


@pytest.mark.parametrize("size,frames,batch", [((8, 8), 64, 16), ((8, 8), 70, 32), ((16, 8), 150, 64), ((24, 16), 40, 16), ((128, 72), 23, 16), ((128, 72), 23, 5),
                                               ((128, 72), 23, 1), ((96, 54), 37, 2)], ids=lambda v: str(v))
def test_frame_pipelining_is_bit_exact(pkg, native_lib, oracle, size, frames, batch):
    """Consecutive Render() calls are launched as one kernel that pipelines the frames (pt_set_frame_batch); the running
    mean of frame f+1 must see frame f of the same pixel (carried by the alpha tag inside a batch).  Tiny images put
    the same tile of consecutive frames into flight at the same time, so the hand-over is exercised constantly."""
    w = configs.Workload("pipelined", "default", size[0], size[1], 8, "sky_f32_32", frames=frames)
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
    pt.SetFrameBatch(batch)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    for _ in range(frames):
        pt.Render()
    got = pt.Result
    assert pt.Samples == frames
    pt.Dispose()
    assert (got[..., 3] == 1.0).all(), "alpha tags must never be visible after a read"
    assert_bit_exact(got, oracle_render(oracle, w), f"pipelined {size} x{frames} batch {batch}")


@pytest.mark.parametrize("scene,depth,env,frames,spp", [("default", 0, "sky_f32_32", 9, 1), ("default", 1, "sky_f32_32", 9, 1),
                                                        ("empty", 4, "tiny_4", 20, 1), ("edge", 24, "sky_srgb_32", 12, 1),
                                                        ("stress256", 8, "sky_f32_32", 7, 1), ("default", 6, "sky_f32_32", 21, 3),
                                                        ("randmat", 13, "sky_f32_32", 9, 10), ("default", 0, "sky_f32_32", 5, 2)],
                         ids=lambda v: str(v))
def test_frame_pipelining_edge_parameters(pkg, native_lib, oracle, scene, depth, env, frames, spp):
    """Pipelined launches with degenerate depths (every path ends in the tile pass), an empty scene, NaN-producing glass
    (edge scene), materials read from device memory (256 spheres) and several samples per pixel (the kernel without
    the tile pass)."""
    w = configs.Workload("pipe_edge", scene, 72, 40, depth, env, frames=frames, spp=spp)
    assert_bit_exact(hip_render(pkg, w), oracle_render(oracle, w), f"{scene} depth {depth} x{frames}")


@pytest.mark.parametrize("spp", [1, 2])
def test_frame_pipelining_survives_oversubscription(pkg, native_lib, oracle, spp, monkeypatch):
    """More workgroups than the GPU can keep resident (8 per CU requested; registers allow 6 resp. 5): a workgroup that
    starts late must not own early (frame, tile) work that running workgroups wait for — every chunk of a pipelined
    launch is drawn from the global counter.  (With static first chunks this configuration deadlocks.)"""
    pkg.native.debug_set("batch_wg", 8)  # (a tuning knob of the library: csrc/pt_tuning.hpp)
    try:
        w = configs.Workload("oversub", "default", 640, 360, 5, "sky_f32_32", frames=24, spp=spp)
        got = hip_render(pkg, w)
    finally:
        pkg.native.debug_set("batch_wg", 0)
    xy = np.stack([np.arange(0, 640 * 360, 97) % 640, np.arange(0, 640 * 360, 97) // 640], 1)
    sc, basic, objs, env, kw = configs.inputs(w)
    want = None
    for f in range(w.frames):
        want = oracle.render_pixels(w.width, w.height, basic, objs, env, xy, frame=f, last=want, **kw)
    assert np.array_equal(bits(got[xy[:, 1], xy[:, 0]]), bits(want))


def test_frame_pipelining_applies_uploads_to_later_frames_only(pkg, native_lib, oracle):
    """Inputs changed between two Render() calls must not reach the frames already accepted: 5 frames with camera A,
    then 4 with camera B and a moved sphere, then a depth change — compared with the same sequence launched frame by
    frame."""
    def run(batch):
        w = configs.Workload("seq", "default", 96, 54, 6, "sky_f32_32")
        sc, basic, objs, env, kw = configs.inputs(w)
        pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
        pt.SetFrameBatch(batch)
        pt.UploadScene(sc)
        pt.UploadBasicData(basic)
        for _ in range(5):
            pt.Render()
        cam = pkg.camera.Camera(position=(-10.0, 2.0, -6.0), look_x=-60.0, look_y=-5.0)
        pt.UploadBasicData(pkg.camera.basic_data_ubo(cam, w.width, w.height))
        sc.spheres[3].position = pkg.scene.vec3(-9.0, 0.0, -9.0)
        pt.UploadScene(sc)
        for _ in range(4):
            pt.Render()
        pt.RayDepth = 3
        for _ in range(3):
            pt.Render()
        out = pt.Result
        pt.Dispose()
        return out
    assert_bit_exact(run(64), run(1), "batched vs frame-by-frame launch sequence")


def test_randomised_scenes_cameras_and_parameters(pkg, native_lib):
    """tools/fuzz_parity.py: random scenes (0-256 spheres from tiny to room-sized, nested / overlapping, 0-64 cuboids,
    random materials), random cameras, lens, depth, spp, image size, frame count and batch size; every image must equal the
    oracle bit for bit.  (22,000 cases were run while the tile pass and the frame pipelining were developed; 80 here.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "80", "7"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


# ------------------------------------------------------------------------------------------------ (2) HIP vs reference fixtures
@pytest.mark.parametrize("name", fixtures.names("frame_"))
def test_hip_matches_reference_fixtures(pkg, native_lib, name, parity_report):
    """Layer 2: the HIP path against the reference GLSL's own output (llvmpipe fixtures).  Each fixture has its OWN pass
    mark: the agreement measured for it (tests/golden/agreement.json) minus 0.3 percentage points (tests/tolerances.py)."""
    fx = fixtures.load(name)
    got = fixtures.hip_frames(pkg, fx)
    ref = fx["expected"]
    # sRGB8 environments: llvmpipe decodes texels with a cubic approximation (fixtures.llvmpipe_srgb_lut); the product uses
    # the exact GL 4.5 formula, so that fixture is compared with a band that covers the approximation (0.6 %)
    srgb = fx["env"].dtype == np.uint8
    for k in range(ref.shape[0]):
        st = tol.agreement(ref[k], got[k], srgb_band=srgb)
        need = tol.min_fraction(name, k)
        parity_report(f"HIP {name} #{k}", st, need)
        assert st["within"] >= need, f"{name} #{k}: {100 * st['within']:.3f}% within tolerance, need {100 * need:.2f}%"
        assert st["mean_rel_err"] <= (tol.SRGB_MEAN_REL_TOL if srgb else tol.MEAN_REL_TOL)


@pytest.mark.parametrize("name", [n for n in fixtures.names("envonly_") if "srgb" not in n])
def test_hip_environment_sampler_matches_reference(pkg, native_lib, name):
    fx = fixtures.load(name)
    got = fixtures.hip_frames(pkg, fx)[0]
    ref = fx["expected"]
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= tol.ENV_REL_TOL


@pytest.mark.parametrize("name", fixtures.names("sparse_"))
def test_full_size_frames_vs_reference_sparse_pixels(pkg, native_lib, oracle, name, parity_report):
    """BASELINE configs C1/C2/C3/C5 — and C2 with the reference's atmosphere environment, the exact workload bench.py
    times — at FULL resolution on the GPU: the 4096 reference pixels must agree within the fixture's own pass mark, and
    the same pixels must equal the oracle bit for bit."""
    fx = fixtures.load(name)
    pt = fixtures.hip_tracer(pkg, fx)
    pt.Render()
    img = pt.Result
    pt.Dispose()
    xy = fx["xy"]
    got = img[xy[:, 1], xy[:, 0], :3]
    st, need = tol.agreement(fx["expected"], got), tol.min_fraction(name)
    parity_report(f"HIP {name}", st, need)
    assert st["within"] >= need, f"{name}: {100 * st['within']:.3f}%, need {100 * need:.2f}%"
    full_mean = img[..., :3].mean(axis=(0, 1), dtype=np.float64)
    assert np.all(np.abs(full_mean - fx["frame_mean"]) <= tol.MEAN_REL_TOL * np.abs(fx["frame_mean"]))
    want = oracle.render_pixels(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], xy, frame=0,
                                **fixtures.kwargs(fx))
    assert_bit_exact(img[xy[:, 1], xy[:, 0]], want, name)


def test_large_srgb_cube_sparse_pixels_equal_oracle(pkg, native_lib, oracle):
    """The reference's OTHER environment shape: a 2048^2 x 6 SRGB8_A8 cube (MainWindow.cs:177-187; 100.7 MB, synthetic
    content).  Texel indices reach 6 * 2048^2 = 25.2 M: 1080p frame on the GPU, 8192 seeded pixels against the oracle, bit
    for bit (the 32-bit tap index math of the sampler is what a small cube never exercises)."""
    env = pkg.envmap.synthetic_sky_srgb8(2048)
    assert env.shape == (6, 2048, 2048, 4) and env.dtype == np.uint8
    w = configs.Workload("sky2048", "default", 1920, 1080, 8, "unused")
    sc = configs.make_scene("default")
    basic = pkg.camera.basic_data_ubo(pkg.camera.Camera(), w.width, w.height)
    kw = dict(num_spheres=sc.num_spheres, num_cuboids=sc.num_cuboids, ray_depth=8, spp=1, focal_length=20.0, aperture=0.14)
    pt = pkg.PathTracer(env, w.width, w.height, 8, 1, 20.0, 0.14)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    pt.Render()
    pt.Render()
    img = pt.Result
    pt.Dispose()
    rng = np.random.RandomState(77)
    xy = np.stack([rng.randint(0, w.width, 8192), rng.randint(0, w.height, 8192)], axis=1).astype(np.int32)
    f0 = oracle.render_pixels(w.width, w.height, basic, sc.ubo_bytes(), env, xy, frame=0, **kw)
    want = oracle.render_pixels(w.width, w.height, basic, sc.ubo_bytes(), env, xy, frame=1, last=f0, **kw)
    got = img[xy[:, 1], xy[:, 0]]
    assert_bit_exact(got, want, "2048^2 sRGB8 cube, two frames")
    # and the environment alone (empty scene, pinhole): every pixel is a cube lookup, taps all over the six faces
    for look in ((-32.2, 0.8), (135.0, -35.26), (0.0, -89.0)):
        cam = pkg.camera.Camera(look_x=look[0], look_y=look[1])
        b2 = pkg.camera.basic_data_ubo(cam, 640, 360)
        pt = pkg.PathTracer(env, 640, 360, 2, 1, 20.0, 0.0)
        pt.UploadBasicData(b2)
        pt.Render()
        got = pt.Result
        pt.Dispose()
        want = oracle.render(640, 360, b2, bytes(26624), env, num_spheres=0, num_cuboids=0, ray_depth=2, spp=1,
                             focal_length=20.0, aperture=0.0, num_frames=1)
        assert_bit_exact(got, want, f"2048^2 sRGB8 cube, environment only, look {look}")


# ------------------------------------------------------------------------------------------------ (3) properties at full size
def test_c2_full_frame_equals_oracle(pkg, native_lib, oracle):
    """The benchmark configuration itself (1920x1080, 8 bounces, default scene): every one of the 2,073,600 pixels,
    two accumulated frames, bit for bit."""
    w = configs.C2
    assert_bit_exact(hip_render(pkg, w, frames=2), oracle_render(oracle, w, frames=2), w.name)


@pytest.mark.parametrize("w,world", [(configs.C2, 8), (configs.C4, 8), (configs.Workload("odd", "default", 333, 211, 8, "sky_f32_32"), 3)],
                         ids=["1080p/8", "4k/8x270rows", "odd/3"])
def test_row_tiling_is_bit_identical(pkg, native_lib, w, world):
    """Multi-GPU emulated on one GPU: rendering the G row blocks separately (pt_set_tile) and stacking them equals
    the untiled render bit for bit — 270-row tiles of C4 are not a multiple of the 8-row tile height."""
    from opentk_pathtracer_amd import distributed as D
    full = hip_render(pkg, w, frames=2)
    parts = [hip_render(pkg, w, frames=2, tile=D.row_block(w.height, r, world)) for r in range(world)]
    assert [p.shape[0] for p in parts] == [D.row_block(w.height, r, world)[1] for r in range(world)]
    assert np.array_equal(bits(np.concatenate(parts)), bits(full))


def test_determinism_and_checksum(pkg, native_lib):
    w = configs.C2
    a, b = hip_render(pkg, w, frames=3), hip_render(pkg, w, frames=3)
    assert np.array_equal(bits(a), bits(b))
    assert (a[..., 3] == 1.0).all() and np.isfinite(a).all()


def test_resume_from_checkpoint(pkg, native_lib):
    """pt_read_result / pt_write_result: render 2 frames, save, restore into a new renderer, render 2 more ==
    4 frames straight."""
    w = configs.Workload("resume", "default", 320, 180, 8, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    straight = hip_render(pkg, w, frames=4)
    first = hip_render(pkg, w, frames=2)
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    pt.WriteResult(first, 2)
    assert pt.FrameIndex == 2 and pt.Samples == 2
    pt.Render()
    assert pt.Render() == 4
    assert np.array_equal(bits(pt.Result), bits(straight))


def test_checkpoint_file_resume_and_screenshot(pkg, native_lib, tmp_path):
    """checkpoint.py on a real renderer: 5 frames, save, load into a fresh renderer (also a banded multi-GPU tile), 4 more
    frames == 9 frames straight; the screenshot PNG holds pt_present_rgba8's bytes top-down."""
    w = configs.Workload("ckpt", "default", 160, 96, 8, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)

    def renderer(tile=None):
        pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
        pt.UploadScene(sc)
        pt.UploadBasicData(basic)
        if tile:
            pt.SetInterleavedTile(*tile)
        return pt

    for tile in (None, (1, 3, 16)):
        straight = renderer(tile)
        for _ in range(9):
            straight.Render()
        a = renderer(tile)
        for _ in range(5):
            a.Render()
        path = str(tmp_path / f"acc_{bool(tile)}.ptck")
        a.SaveCheckpoint(path)
        b = renderer(tile)
        assert b.LoadCheckpoint(path)["frame_index"] == 5 and b.FrameIndex == 5
        for _ in range(4):
            b.Render()
        assert np.array_equal(bits(b.Result), bits(straight.Result))
        if tile is None:
            with pytest.raises(pkg.checkpoint.CheckpointError):
                renderer((0, 2, 16)).LoadCheckpoint(path)
            png = str(tmp_path / "shot.png")
            b.SaveScreenshot(png)
            assert np.array_equal(pkg.checkpoint.decode_png_rgb8(open(png, "rb").read()), b.Present()[::-1, :, :3])


def test_reset_and_resize_semantics(pkg, native_lib, oracle):
    """ResetRenderer (PathTracer.cs:137-140) restarts at frame 0 and the stale image must not leak in;
    SetSize (PathTracer.cs:131-135) reallocates; partial scene updates (Gui.cs:212-216) take effect."""
    w = configs.Workload("reset", "default", 160, 90, 6, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    for _ in range(3):
        pt.Render()
    pt.ResetRenderer()
    assert pt.FrameIndex == 0
    pt.Render()
    want = oracle.render(w.width, w.height, basic, objs, env, num_frames=1, **kw)
    assert_bit_exact(pt.Result, want, "after ResetRenderer")
    # one object edited in place: 80-byte SubData at its BufferOffset
    sph = sc.spheres[5]
    sph.material = pkg.scene.Material(albedo=(1, 0.2, 0.2), emissiv=(2, 2, 2))
    d = sph.gpu_data()
    pt.GameObjectsUBO.SubData(sph.buffer_offset, d.nbytes, d)
    pt.ResetRenderer()
    pt.Render()
    want = oracle.render(w.width, w.height, basic, sc.ubo_bytes(), env, num_frames=1, **kw)
    assert_bit_exact(pt.Result, want, "after a partial scene update")
    # resize
    w2 = configs.Workload("resized", "default", 200, 120, 6, "sky_f32_32")
    basic2 = pkg.camera.basic_data_ubo(pkg.camera.Camera(), w2.width, w2.height)
    pt.SetSize(w2.width, w2.height)
    pt.UploadBasicData(basic2)
    pt.Render()
    want = oracle.render(w2.width, w2.height, basic2, sc.ubo_bytes(), env, num_frames=1, **kw)
    assert_bit_exact(pt.Result, want, "after SetSize")
    # RayDepth / SPP property setters
    pt.RayDepth, pt.SPP = 3, 2
    pt.ResetRenderer()
    assert pt.Render() == 2
    want = oracle.render(w2.width, w2.height, basic2, sc.ubo_bytes(), env, num_frames=1, **dict(kw, ray_depth=3, spp=2))
    assert_bit_exact(pt.Result, want, "after RayDepth/SPP change")


def test_bound_external_buffer_and_pitch(pkg, native_lib):
    """torch-owned accumulation buffer (the multi-GPU plumbing) + pitched read-back."""
    torch = pytest.importorskip("torch")
    w = configs.Workload("bind", "default", 128, 72, 6, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    ref_img = hip_render(pkg, w, frames=2)
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    buf = torch.zeros((w.height, w.width, 4), dtype=torch.float32, device="cuda")
    pt.BindResultBuffer(buf.data_ptr(), buf.numel() * 4)
    pt.Render()
    pt.Render()
    pt.Synchronize()
    assert np.array_equal(bits(buf.cpu().numpy()), bits(ref_img))
    pitched = np.zeros((w.height, w.width + 5, 4), np.float32)
    pkg.native.check(native_lib.pt_read_result(pt._h, pitched.ctypes.data_as(C.POINTER(C.c_float)), (w.width + 5) * 16), pt._h)
    assert np.array_equal(bits(pitched[:, :w.width]), bits(ref_img)) and (pitched[:, w.width:] == 0).all()


def test_two_renderers_interleaved(pkg, native_lib, oracle):
    """Two handles on one device, different scenes / sizes / parameters, frames interleaved call by call: handles share
    no state (each has its own streams, queue counters, pending-frame list)."""
    wa = configs.Workload("a", "default", 160, 90, 8, "sky_f32_32", frames=11)
    wb = configs.Workload("b", "stress256", 96, 64, 5, "sky_srgb_32", frames=7, spp=2)
    pts = []
    for w in (wa, wb):
        sc, basic, objs, env, kw = configs.inputs(w)
        pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, w.spp, w.focal_length, w.aperture)
        pt.UploadScene(sc)
        pt.UploadBasicData(basic)
        pts.append(pt)
    for i in range(11):
        pts[0].Render()
        if i < 7:
            pts[1].Render()
    got = [pt.Result for pt in pts]
    for pt in pts:
        pt.Dispose()
    assert_bit_exact(got[0], oracle_render(oracle, wa), "renderer A")
    assert_bit_exact(got[1], oracle_render(oracle, wb), "renderer B")


def test_error_codes(pkg, native_lib):
    N = pkg.native
    h = C.c_void_p()
    assert native_lib.pt_create(0, 64, 64, C.byref(h)) == N.PT_OK
    assert native_lib.pt_render(h, None) == N.PT_E_NO_ENVIRONMENT
    blob = (C.c_char * 200)()
    assert native_lib.pt_upload_basic_data(h, 100, 64, blob) == N.PT_E_OUT_OF_RANGE
    assert native_lib.pt_upload_basic_data(h, 128, 16, blob) == N.PT_OK
    assert native_lib.pt_upload_game_objects(h, 26600, 80, blob) == N.PT_E_OUT_OF_RANGE
    assert native_lib.pt_upload_game_objects(h, -4, 8, blob) == N.PT_E_OUT_OF_RANGE
    assert native_lib.pt_set_params(h, 257, 0, 1, 1, 1.0, 0.0) == N.PT_E_OUT_OF_RANGE
    assert native_lib.pt_set_params(h, 0, 65, 1, 1, 1.0, 0.0) == N.PT_E_OUT_OF_RANGE
    assert native_lib.pt_set_params(h, 1, 1, 1, 0, 1.0, 0.0) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_set_size(h, 0, 5) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_set_tile(h, 60, 10) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_set_environment(h, 4, 7, None) == N.PT_E_BAD_ARGUMENT
    assert b"GameObjectsUBO" in native_lib.pt_last_error(h) or len(native_lib.pt_last_error(h)) > 0
    assert native_lib.pt_create(99, 64, 64, C.byref(C.c_void_p())) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_destroy(h) == N.PT_OK


# ------------------------------------------------------------------------------------------------ atmosphere kernel
def test_atmosphere_equals_oracle_and_reference(pkg, native_lib, oracle):
    ubo = pkg.camera.atmospheric_data_ubo()
    for name in fixtures.names("atmo_"):
        fx = fixtures.load(name)
        size, isteps, jsteps = (int(v) for v in fx["params"])
        pt = pkg.PathTracer(None, 16, 16, 1, 1, 1.0, 0.0)
        at = pkg.AtmosphericScatterer(size, fx["ubo"].tobytes(), fx["light_pos"], pt)
        at.ISteps, at.JSteps, at.LightIntensity = isteps, jsteps, float(fx["intensity"])
        pt.EnvironmentMap = at
        got = at.Result
        want = oracle.atmosphere(size, fx["ubo"].tobytes(), fx["light_pos"], float(fx["intensity"]), isteps, jsteps)
        assert np.array_equal(bits(got), bits(want)), f"{name}: HIP atmosphere differs from the oracle"
        # ... and the GPU's own cube against the reference's, with the per-fixture marks frozen in round 6 (24^2 / 32^2 / 48^2 — one of them
        # not a power of two —, 3 and 15 sun-ray steps): the kernel's roots and its half-cube symmetry changed in round 5 together with
        # the oracle, so "equals the oracle" alone would not show a drift from the reference
        err = tol.atmo_error(fx["expected"], got[..., :3])
        worst, share = tol.ATMO_MARKS[name]
        assert err.max() <= worst and (err < 1e-4).mean() >= share, f"{name}: {err.max():.3g} / {100 * (err < 1e-4).mean():.2f} %"
        pt.Dispose()
    assert len(ubo) == 464


def test_atmosphere_service_resolutions_and_parameter_changes(pkg, native_lib, oracle):
    """the GUI re-renders the atmosphere whenever a knob moves and switches its resolution between 32 and
    2048 (Gui.cs:89-145).  Re-rendering on one renderer with changed parameters == a fresh computation (bit for bit, vs
    the oracle at 128); the largest size the GUI offers runs, is finite everywhere, and is timed through pt_timer_*."""
    pt = pkg.PathTracer(None, 16, 16, 1, 1, 1.0, 0.0)
    at = pkg.AtmosphericScatterer(128, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), pt)
    pt.EnvironmentMap = at
    first = at.Result
    at.LightPos = pkg.camera.atmosphere_light_pos(0.15)      # Gui.cs "Time" slider
    at.LightIntensity, at.ISteps, at.JSteps = 22.0, 30, 8
    at.Render()
    changed = at.Result
    want = oracle.atmosphere(128, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.15), 22.0, 30, 8)
    assert np.array_equal(bits(changed), bits(want)) and not np.array_equal(bits(changed), bits(first))
    at.Size, at.ISteps, at.JSteps = 2048, 50, 15            # Gui.cs:93 resolution switch
    pt.TimerBegin()
    at.Render()
    ms = pt.TimerEnd()
    big = at.Result
    assert big.shape == (6, 2048, 2048, 4) and np.isfinite(big).all() and (big[..., 3] == 1).all() and 0.0 < ms < 2000.0
    pt.Dispose()


def test_default_startup_sequence(pkg, native_lib, oracle):
    """MainWindow.OnLoad (MainWindow.cs:174-189,203): atmosphere cube at 256 -> PathTracer(env = atmosphere,
    rayDepth 13, spp 1, f 20, aperture 0.14) -> LoadScene -> frames.  End-to-end against the oracle, which renders
    with the atmosphere cube the oracle itself computed."""
    W, H = 208, 208
    sc, cam = pkg.scene.default_scene(), pkg.camera.Camera()
    basic = pkg.camera.basic_data_ubo(cam, W, H)
    ubo, lp = pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5)
    pt = pkg.PathTracer(None, W, H, 13, 1, 20.0, 0.14)
    pt.EnvironmentMap = pkg.AtmosphericScatterer(256, ubo, lp, pt)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    pt.Render()
    pt.Render()
    env = oracle.atmosphere(256, ubo, lp)
    want = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_spheres=48, num_cuboids=7, ray_depth=13, num_frames=2)
    assert_bit_exact(pt.Result, want, "startup sequence")


def test_cpp_host_startup_sequence(pkg, native_lib, oracle, tmp_path):
    """The C++ host mirror (host/pt_host_demo) replays MainWindow.OnLoad + 3 frames entirely in C++ over the C ABI;
    its image must equal the oracle fed with the blobs the C++ host produced."""
    import subprocess
    demo = pkg.native.build_host_demo()
    W, H, frames, depth, atmo = 200, 120, 3, 13, 64
    out, cam, scn, cube = tmp_path / "img.f32", tmp_path / "cam.bin", tmp_path / "scene.bin", tmp_path / "env.f32"
    subprocess.run([demo, "render", str(W), str(H), str(frames), str(out), str(depth), str(atmo), str(cube)], check=True)
    subprocess.run([demo, "dump-camera", str(W), str(H), str(cam)], check=True)
    subprocess.run([demo, "dump-scene", str(scn)], check=True)
    got = np.fromfile(out, np.float32).reshape(H, W, 4)
    # the C++ host builds its own atmosphere UBO / light position (float arithmetic of its own matrix code), so its cube is not the
    # Python harness's bit for bit: the cube it rendered with is read back through pt_read_environment, checked against the harness's
    # to float noise, and the oracle renders with exactly that cube -> the image must then be identical
    env_host = np.fromfile(cube, np.float32).reshape(6, atmo, atmo, 4)
    env_py = oracle.atmosphere(atmo, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5))
    assert np.abs(env_host - env_py).max() <= 2e-3 * max(1.0, float(np.abs(env_py).max()))
    want = oracle.render(W, H, cam.read_bytes(), scn.read_bytes(), env_host, num_spheres=48, num_cuboids=7, ray_depth=depth,
                         num_frames=frames)
    assert_bit_exact(got, want, "C++ host startup sequence")


def test_cpp_host_checkpoint_resume(pkg, native_lib, tmp_path):
    """host/pt_host.hpp SaveCheckpoint / LoadCheckpoint / SaveScreenshotPPM: 3 frames + checkpoint + NEW renderer + 2 frames
    in C++ == 5 frames straight (bit for bit); the file is the one checkpoint.py reads."""
    import subprocess
    demo = pkg.native.build_host_demo()
    W, H = 96, 64
    straight, resumed, ck, ppm = (tmp_path / n for n in ("straight.f32", "resumed.f32", "acc.ptck", "shot.ppm"))
    subprocess.run([demo, "render", str(W), str(H), "5", str(straight)], check=True)
    subprocess.run([demo, "resume", str(W), str(H), "3", "2", str(resumed), str(ck), str(ppm)], check=True)
    a, b = np.fromfile(straight, np.float32), np.fromfile(resumed, np.float32)
    assert a.size == W * H * 4 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    hdr, img = pkg.checkpoint.read_checkpoint_file(str(ck))
    assert (hdr["width"], hdr["height"], hdr["rows"], hdr["frame_index"], hdr["ray_depth"], hdr["spp"]) == (W, H, H, 3, 13, 1)
    assert img.shape == (H, W, 4) and (img[..., 3] == 1).all()
    head = ppm.read_bytes()[:15]
    assert head.startswith(b"P6\n96 64\n255\n") and ppm.stat().st_size == len(b"P6\n96 64\n255\n") + W * H * 3


def test_many_consecutive_frames_queue_accounting(pkg, native_lib, oracle):
    """The persistent kernel's global ticket counter is never reset (every launch must consume exactly the number of
    tickets the host accounts for): 48 consecutive frames, then bit-compare the accumulation with the oracle."""
    w = configs.Workload("queue", "default", 416, 234, 8, "sky_f32_32")
    assert_bit_exact(hip_render(pkg, w, frames=48), oracle_render(oracle, w, frames=48), "48 frames")


@pytest.mark.parametrize("w,world,band", [(configs.C2, 8, 16), (configs.Workload("odd", "default", 333, 211, 8, "sky_f32_32"), 3, 8)],
                         ids=["1080p/8x16", "odd/3x8"])
def test_interleaved_tiling_is_bit_identical(pkg, native_lib, w, world, band):
    """Block-cyclic ownership (pt_set_interleaved_tile), emulated on one GPU: each rank's compact band storage, scattered
    back to image rows, reproduces the untiled render bit for bit."""
    from opentk_pathtracer_amd import distributed as D
    sc, basic, objs, env, kw = configs.inputs(w)
    full = hip_render(pkg, w, frames=2)
    out = np.zeros_like(full)
    for r in range(world):
        pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, w.spp, w.focal_length, w.aperture)
        pt.UploadScene(sc)
        pt.UploadBasicData(basic)
        pt.SetInterleavedTile(r, world, band)
        pt.Render()
        pt.Render()
        rows = D.interleaved_rows(w.height, r, world, band)
        part = pt.Result
        assert part.shape[0] == len(rows)
        out[rows] = part
        pt.Dispose()
    assert np.array_equal(bits(out), bits(full))


def test_present_rgba8_equals_oracle_and_reference(pkg, native_lib, oracle):
    """pt_present_rgba8 (ScreenEffect.Render + PostProcessing/fragment.glsl fused with the read-back): the HIP pass on a
    rendered HDR image equals the oracle's RGBA8 byte for byte; and on the reference fixture's input image it is within
    1 LSB of the reference's own functions."""
    w = configs.Workload("present", "default", 320, 180, 8, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    for _ in range(6):
        pt.Render()
    hdr, ldr = pt.Result, pt.Present()
    _, want = oracle.postprocess(hdr)
    assert ldr.shape == (w.height, w.width, 4) and np.array_equal(ldr, want)
    # fixture: load its input image as the accumulation image, then present
    fx = fixtures.load("post_aces_gamma")
    img = fx["image"]
    pt2 = pkg.PathTracer(env, img.shape[1], img.shape[0], 1, 1, 1.0, 0.0)
    pt2.WriteResult(img, 1)
    got = pt2.Present()
    _, want2 = oracle.postprocess(img)
    assert np.array_equal(got, want2)
    ref_u8 = (np.clip(fx["expected"], 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    diff = np.abs(ref_u8.astype(int) - got[..., :3].astype(int))
    assert diff.max() <= 1 and (diff == 0).mean() >= 0.999
    # tiled present: row blocks concatenate to the full presented image
    parts = []
    for r in range(3):
        from opentk_pathtracer_amd import distributed as D
        p3 = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
        p3.UploadScene(sc)
        p3.UploadBasicData(basic)
        p3.SetTile(*D.row_block(w.height, r, 3))
        for _ in range(6):
            p3.Render()
        parts.append(p3.Present())
        p3.Dispose()
    assert np.array_equal(np.concatenate(parts), ldr)


def test_device_side_rgba8_tile_for_the_multi_gpu_present(pkg, native_lib):
    """distributed.present_rgba8: the tile every rank contributes is the library's own post-processed image, read
    zero-copy from device memory (world 1 here; the gather itself is covered over gloo on CPU)."""
    import torch
    from opentk_pathtracer_amd import distributed as D
    w = configs.Workload("p8", "default", 200, 120, 6, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    pt.SetInterleavedTile(1, 3, 16)  # rank 1 of 3, 16-row bands
    for _ in range(4):
        pt.Render()
    pad = D.max_interleaved_rows(w.height, 3, 16)
    tile = D.postprocessed_tile(pt, pad)
    assert tile.dtype == torch.uint8 and tuple(tile.shape) == (pad, w.width, 4) and tile.is_cuda
    assert np.array_equal(tile[:pt.rows].cpu().numpy(), pt.Present())
    assert not tile[pt.rows:].any()
    pt.Dispose()


def test_converged_accumulation_vs_reference(pkg, native_lib):
    """96 accumulated frames on the GPU against the reference's 96-frame accumulation (llvmpipe fixture): > 50 dB PSNR,
    99 % of pixels within 2 %, unbiased mean — the statistical half of the stated tolerance."""
    fx = fixtures.load("converged_default_96x54_d8_acc96")
    pt = fixtures.hip_tracer(pkg, fx)
    for _ in range(fx["frames"]):
        pt.Render()
    got, ref = pt.Result[..., :3], fx["expected"]
    a, b = ref / (1.0 + ref), got / (1.0 + got)
    psnr = 10.0 * np.log10(1.0 / max(float(np.mean((a.astype(np.float64) - b) ** 2)), 1e-20))
    assert psnr > 50.0
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 0.05)
    assert (rel.max(-1) < 0.02).mean() > 0.99 and abs(got.mean() - ref.mean()) < 1e-3 * ref.mean()


def test_edge_scene_differences_are_nan_direction_lookups(pkg, native_lib):
    """The quirk scene on the GPU (fixture from tests/golden/make_golden.py `edge`): 64 frames, read back after every
    frame; per sample, every gross difference from the reference GLSL ends in a NaN-direction environment lookup (flags from the
    oracle's diagnostic build, which the HIP path equals bit for bit) or belongs to the branch-flip budget."""
    fx = fixtures.load("edge_nanenv_edge_64x36_d16")
    pt = fixtures.hip_tracer(pkg, fx)
    acc = []
    for _ in range(fx["frames"]):
        pt.Render()
        acc.append(pt.Result[..., :3].copy())
    pt.Dispose()
    st = fixtures.edge_nanenv_check(fx, np.stack(acc))
    assert st["nan_env"] > 100, "the scene no longer exercises the quirk"
    assert st["gross_unflagged"] <= 3e-4 * st["samples"] and st["masked_mean_rel_err"] <= 4e-4, st


def test_external_stream_and_interleaved_uploads(pkg, native_lib, oracle):
    """pt_set_stream with a torch-owned HIP stream, and scene edits interleaved with frames in flight: every upload is
    ordered against the stripe kernels still running (the stripes live on the library's own streams)."""
    torch = pytest.importorskip("torch")
    w = configs.Workload("stream", "default", 256, 144, 8, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    stream = torch.cuda.Stream()
    pt = pkg.PathTracer(env, w.width, w.height, w.ray_depth, 1, w.focal_length, w.aperture)
    pt.SetStream(stream.cuda_stream)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    for _ in range(3):
        pt.Render()
    # edit one sphere while 3 frames may still be in flight, restart, render 2 frames, edit again, 1 more frame
    sph = sc.spheres[7]
    sph.material = pkg.scene.Material(albedo=(0.2, 0.9, 0.3), specular_chance=0.5, specular_roughness=0.2)
    d = sph.gpu_data()
    pt.GameObjectsUBO.SubData(sph.buffer_offset, d.nbytes, d)
    pt.ResetRenderer()
    pt.Render()
    pt.Render()
    stream.synchronize()
    want = oracle.render(w.width, w.height, basic, sc.ubo_bytes(), env, num_frames=2, **kw)
    assert_bit_exact(pt.Result, want, "after an upload ordered behind in-flight frames")
    pt.SetStream(None)
    pt.Render()
    want = oracle.render(w.width, w.height, basic, sc.ubo_bytes(), env, frame_start=2, num_frames=1, image=want, **kw)
    assert_bit_exact(pt.Result, want, "back on the library's own stream")
