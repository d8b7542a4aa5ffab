"""GPU tests of the sphere grid (large scenes, spp = 1): the generic bounce of scenes with >= 64 spheres walks a uniform grid
instead of testing every sphere (csrc/pt_sphere_grid.hpp, ray_trace_t<GRID>).  The grid only selects which spheres a ray
tests; the image must stay bit-identical to the oracle's in-order loop (compute.glsl:226-258) — including the cases the
reference's acceptance rule makes order-dependent: origins inside one or several spheres (refraction, overlapping spheres,
the camera inside a sphere), identical spheres (equal t1: the lower index wins), rays from far outside (in-order fallback).
Run with `pytest -m gpu` on an MI355X.  Nothing here reads /root/reference."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_parity import assert_bit_exact

pytestmark = pytest.mark.gpu

W, H = 112, 64


def grid_info(native_lib, pt):
    out = (C.c_int * 5)()
    native_lib.pt_debug_sphere_grid.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    assert native_lib.pt_debug_sphere_grid(pt._h, out) == 0
    return list(out)


def random_material(S, rng, glass):
    if glass:
        return S.Material(albedo=rng.rand(3), absorbance=rng.rand(3) * 0.5, specular_chance=0.03 * rng.rand(), ior=1.0 + rng.rand(),
                          refraction_chance=0.6 + 0.38 * rng.rand(), refraction_roughness=rng.rand() * 0.3)
    return S.Material(albedo=rng.rand(3), emissiv=(rng.rand(3) if rng.rand() < 0.1 else S.vec3(0.0)), specular_chance=rng.rand() * 0.6,
                      specular_roughness=rng.rand())


def big_scene(pkg, rng, n, *, radius=(0.2, 1.2), glass_fraction=0.0, duplicates=0, offset=(0.0, 0.0, 0.0), box=((-18, -11, -20), (18, 11, 0))):
    S = pkg.scene
    sc = S.Scene()
    lo, hi = np.array(box[0], np.float32), np.array(box[1], np.float32)
    off = np.array(offset, np.float32)
    for i in range(n):
        if duplicates and i >= n - duplicates:  # identical geometry at a higher index (and another material): equal t1
            src = sc.spheres[rng.randint(0, n - duplicates)]
            sc.spheres.append(S.Sphere(src.position.copy(), src.radius, i, random_material(S, rng, False)))
            continue
        pos = (lo + (hi - lo) * rng.rand(3).astype(np.float32) + off).astype(np.float32)
        sc.spheres.append(S.Sphere(pos, np.float32(rng.uniform(*radius)), i, random_material(S, rng, rng.rand() < glass_fraction)))
    for c in S.default_cuboids():
        c.position = (np.asarray(c.position, np.float32) + off).astype(np.float32)
        sc.cuboids.append(c)
    return sc


def render_both(pkg, oracle, sc, cam, *, depth=6, frames=3, w=W, h=H, edit=None):
    basic = pkg.camera.basic_data_ubo(cam, w, h)
    env = pkg.envmap.synthetic_sky_rgba32f(16)
    kw = dict(num_spheres=sc.num_spheres, num_cuboids=sc.num_cuboids, ray_depth=depth, spp=1, focal_length=12.0, aperture=0.05)
    pt = pkg.PathTracer(env, w, h, depth, 1, 12.0, 0.05)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    for _ in range(frames):
        pt.Render()
    got = pt.Result
    want = oracle.render(w, h, basic, sc.ubo_bytes(), env, num_frames=frames, **kw)
    return pt, got, want, basic, env, kw


def camera(pkg, pos, yaw=-90.0, pitch=0.0):
    return pkg.camera.Camera(position=np.asarray(pos, np.float32), look_x=yaw, look_y=pitch)


def test_grid_is_built_for_large_scenes_only(pkg, native_lib):
    S = pkg.scene
    env = pkg.envmap.synthetic_sky_rgba32f(16)
    pt = pkg.PathTracer(env, 64, 64, 4, 1, 20.0, 0.1)
    pt.UploadScene(S.default_scene())
    assert grid_info(native_lib, pt)[4] == 0, "48 spheres: in-order loop"
    pt.UploadScene(S.stress_scene())
    info = grid_info(native_lib, pt)
    assert info[4] == 1 and 64 <= info[0] * info[1] * info[2] <= 256 and 256 <= info[3] <= 1024, info
    pt.NumSpheres = 48
    assert grid_info(native_lib, pt)[4] == 0
    pt.Dispose()


@pytest.mark.parametrize("case", ["opaque256", "glass200", "overlap128", "duplicates", "tiny", "far_from_origin", "coplanar", "n64"])
def test_grid_traversal_equals_in_order_loop(pkg, native_lib, oracle, case):
    rng = np.random.RandomState({"opaque256": 1, "glass200": 2, "overlap128": 3, "duplicates": 4, "tiny": 5, "far_from_origin": 6,
                                 "coplanar": 7, "n64": 8}[case])
    cam = camera(pkg, (-17.0, 3.5, -8.6), yaw=-32.0, pitch=1.0)
    depth = 6
    if case == "opaque256":
        sc = big_scene(pkg, rng, 256)
    elif case == "glass200":  # refraction: most secondary rays start INSIDE a sphere
        sc = big_scene(pkg, rng, 200, glass_fraction=0.7, radius=(0.5, 1.6))
        depth = 12
    elif case == "overlap128":  # heavy overlap: origins inside several spheres at once, glass and opaque mixed
        sc = big_scene(pkg, rng, 128, glass_fraction=0.5, radius=(1.0, 2.5), box=((-8, -6, -14), (8, 6, -4)))
        depth = 10
    elif case == "duplicates":  # 40 spheres repeat the geometry of lower-index ones exactly
        sc = big_scene(pkg, rng, 160, glass_fraction=0.3, duplicates=40)
    elif case == "tiny":
        sc = big_scene(pkg, rng, 256, radius=(0.01, 0.08))
    elif case == "far_from_origin":  # coordinates ~1000: the grid's margins account for the coarser fp32 spacing there
        off = (1000.0, -700.0, 400.0)
        sc = big_scene(pkg, rng, 180, glass_fraction=0.2, offset=off)
        cam = camera(pkg, (-17.0 + off[0], 3.5 + off[1], -8.6 + off[2]), yaw=-32.0, pitch=1.0)
    elif case == "coplanar":  # all centres in one plane: one layer of cells
        sc = big_scene(pkg, rng, 144, radius=(0.3, 0.9), box=((-18, -11, -10), (18, 11, -10)))
    else:
        sc = big_scene(pkg, rng, 64, glass_fraction=0.3)
    pt, got, want, *_ = render_both(pkg, oracle, sc, cam, depth=depth)
    info = grid_info(native_lib, pt)
    assert info[4] == 1 or case == "overlap128", (case, info)  # (heavy overlap may exceed the reference budget: in-order loop)
    assert_bit_exact(got, want, f"sphere grid, {case}, grid {info}")
    pt.Dispose()


def test_grid_camera_inside_a_sphere_and_far_outside(pkg, native_lib, oracle):
    rng = np.random.RandomState(11)
    sc = big_scene(pkg, rng, 220, glass_fraction=0.4, radius=(0.4, 1.4))
    inside = sc.spheres[57]
    inside.radius = np.float32(2.5)  # the camera sits in this one (and in whatever overlaps it)
    for cam, what in ((camera(pkg, inside.position + np.float32(0.3)), "camera inside sphere 57"),
                      (camera(pkg, (0.0, 0.0, 400.0)), "camera 400 units away (origins out of the grid's reach: in-order loop)"),
                      (camera(pkg, (300.0, 200.0, 100.0), yaw=-160.0, pitch=-30.0), "camera far away, oblique")):
        pt, got, want, *_ = render_both(pkg, oracle, sc, cam, depth=8, frames=2)
        assert grid_info(native_lib, pt)[4] == 1
        assert_bit_exact(got, want, what)
        pt.Dispose()


def test_grid_follows_scene_edits(pkg, native_lib, oracle):
    """The grid is rebuilt before the first launch after a partial upload (one sphere moved / resized), a change of the sphere
    count, and across pipelined launches in flight (the edit lands between two 20-frame launches)."""
    rng = np.random.RandomState(12)
    S = pkg.scene
    sc = big_scene(pkg, rng, 256, glass_fraction=0.2)
    cam = camera(pkg, (-17.0, 3.5, -8.6), yaw=-32.0, pitch=1.0)
    basic = pkg.camera.basic_data_ubo(cam, W, H)
    env = pkg.envmap.synthetic_sky_rgba32f(16)
    kw = dict(num_cuboids=sc.num_cuboids, ray_depth=6, spp=1, focal_length=12.0, aperture=0.05)
    pt = pkg.PathTracer(env, W, H, 6, 1, 12.0, 0.05)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    acc, done = None, 0

    def advance(n, ns):
        nonlocal acc, done
        for _ in range(n):
            pt.Render()
        acc = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_spheres=ns, frame_start=done, num_frames=n, image=acc, **kw)
        done += n

    advance(20, 256)
    sc.spheres[100].position = np.array([-10.0, 2.0, -6.0], np.float32)  # out of its old cells
    sc.spheres[100].radius = np.float32(3.0)
    d = sc.spheres[100].gpu_data()
    pt.GameObjectsUBO.SubData(sc.spheres[100].buffer_offset, 16, d[:4])  # geometry only: a 16-byte partial update
    advance(20, 256)
    assert_bit_exact(pt.Result, acc, "after moving one sphere between two pipelined launches")
    pt.NumSpheres = 130  # fewer spheres: the rest must vanish from the grid
    advance(5, 130)
    assert_bit_exact(pt.Result, acc, "after lowering the sphere count")
    pt.NumSpheres = 40   # below the grid threshold
    assert grid_info(native_lib, pt)[4] == 0
    advance(3, 40)
    pt.NumSpheres = 256
    advance(3, 256)
    assert_bit_exact(pt.Result, acc, "after switching the grid off and on again")
    pt.Dispose()


@pytest.mark.parametrize("batch", [1, 64], ids=["batch_pass_kernel", "in_lane_chain"])
def test_grid_with_several_samples_per_pixel(pkg, native_lib, oracle, batch):
    """spp > 1: frame by frame the batch-pass kernel runs (pt_integrate_multisample_kernel), pipelined over a small image the
    in-lane sample chain — both walk the grid in their generic bounce."""
    rng = np.random.RandomState(14)
    sc = big_scene(pkg, rng, 240, glass_fraction=0.3)
    cam = camera(pkg, (-17.0, 3.5, -8.6), yaw=-32.0, pitch=1.0)
    basic = pkg.camera.basic_data_ubo(cam, W, H)
    env = pkg.envmap.synthetic_sky_rgba32f(16)
    want = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_spheres=240, num_cuboids=sc.num_cuboids, ray_depth=6, spp=3,
                         focal_length=12.0, aperture=0.05, num_frames=5)
    pt = pkg.PathTracer(env, W, H, 6, 3, 12.0, 0.05)
    pt.SetFrameBatch(batch)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    for _ in range(5):
        pt.Render()
    assert grid_info(native_lib, pt)[4] == 1
    assert_bit_exact(pt.Result, want, f"3 spp, frame batch {batch}, 240 spheres")
    pt.Dispose()


def test_grid_on_group_handle_and_caller_stream(pkg, native_lib, oracle):
    torch = pytest.importorskip("torch")
    rng = np.random.RandomState(13)
    sc = big_scene(pkg, rng, 256, glass_fraction=0.3)
    cam = camera(pkg, (-17.0, 3.5, -8.6), yaw=-32.0, pitch=1.0)
    basic = pkg.camera.basic_data_ubo(cam, W, H)
    env = pkg.envmap.synthetic_sky_rgba32f(16)
    want = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_spheres=256, num_cuboids=sc.num_cuboids, ray_depth=6, spp=1,
                         focal_length=12.0, aperture=0.05, num_frames=4)
    pt = pkg.PathTracer(env, W, H, 6, 1, 12.0, 0.05, devices=[0, 0, 0])
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    for _ in range(4):
        pt.Render()
    assert_bit_exact(pt.Result, want, "group handle over 3 parts, 256 spheres")
    pt.Dispose()
    pt = pkg.PathTracer(env, W, H, 6, 1, 12.0, 0.05)
    stream = torch.cuda.Stream()
    pt.SetStream(stream.cuda_stream)
    pt.UploadScene(sc)
    pt.UploadBasicData(basic)
    for _ in range(4):
        pt.Render()
    stream.synchronize()
    assert_bit_exact(pt.Result, want, "caller-owned stream (one launch per frame with drain compaction), 256 spheres")
    pt.SetStream(None)
    pt.Dispose()
