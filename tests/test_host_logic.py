"""CPU tests of the host-side logic: std140 packing, the default scene's data, camera blobs, env generators, and the
C-ABI library's export table / error behaviour without a GPU."""
import ctypes as C
import os

import numpy as np
import pytest


def test_std140_sizes_and_offsets(pkg):
    s = pkg.scene
    assert (s.MATERIAL_SIZE, s.SPHERE_SIZE, s.CUBOID_SIZE) == (64, 80, 96)       # Material.cs:9 Sphere.cs:8 Cuboid.cs:8
    assert s.CUBOIDS_OFFSET == 20480 and s.GAME_OBJECTS_UBO_SIZE == 26624        # MainWindow.cs:17,199-201
    m = s.Material(albedo=(1, 2, 3), emissiv=(4, 5, 6), absorbance=(7, 8, 9), specular_chance=0.25,
                   specular_roughness=0.5, ior=1.5, refraction_chance=0.5, refraction_roughness=0.75)
    assert m.gpu_data().tolist() == [1, 2, 3, 0.25, 4, 5, 6, 0.5, 7, 8, 9, 0.5, 0.75, 1.5, 0, 0]
    sp = s.Sphere(s.vec3(1, 2, 3), 4.0, 7, m)
    assert sp.buffer_offset == 7 * 80 and sp.gpu_data()[:4].tolist() == [1, 2, 3, 4]
    cb = s.Cuboid(s.vec3(0, 0, 0), s.vec3(2, 4, 6), 3, m)
    assert cb.buffer_offset == 20480 + 3 * 96
    d = cb.gpu_data()
    assert d[0:3].tolist() == [-1, -2, -3] and d[4:7].tolist() == [1, 2, 3] and d[8:11].tolist() == [1, 2, 3]


def test_material_ctor_clamps(pkg):
    m = pkg.scene.Material(specular_chance=1.5, refraction_chance=0.9, ior=0.5)  # Material.cs:24-29
    assert m.specular_chance == 1.0 and m.refraction_chance == 0.0 and m.ior == 1.0
    m = pkg.scene.Material(specular_chance=0.25, refraction_chance=0.9)
    assert m.refraction_chance == np.float32(0.75)


def test_default_scene_data(pkg):
    sc = pkg.scene.default_scene()
    assert sc.num_spheres == 48 and sc.num_cuboids == 7                            # MainWindow.cs:208-267
    s0, s35 = sc.spheres[0], sc.spheres[35]
    assert np.allclose(s0.position, [-12.0, -11.2, -5.0]) and s0.radius == np.float32(1.3)
    assert s0.material.specular_chance == 0.0 and s35.material.specular_chance == 1.0
    assert s35.material.specular_roughness == 1.0 and np.allclose(s35.position, [10.0, 9.633333, -5.0], atol=1e-5)
    g0, w0 = sc.spheres[36], sc.spheres[37]
    assert np.allclose(g0.position, [-10.7, 3.0, -20.0]) and g0.material.ior == np.float32(1.05)
    assert g0.material.refraction_chance == np.float32(0.98) and w0.material.ior == np.float32(1.1)
    assert np.allclose(sc.spheres[46].material.absorbance, np.array([1, 2, 3]) * (5 / 6), atol=1e-6)
    light = sc.cuboids[1]
    assert np.allclose(light.material.emissiv, [4.585, 4.725, 2.565]) and np.allclose(light.position, [0, 18.49, -4.0])
    assert np.allclose(light.dimensions, [12.0, 0.005, 7.5])
    front = sc.cuboids[3]
    assert front.material.refraction_chance == np.float32(0.954) and front.material.specular_chance == np.float32(0.04)
    blob = np.frombuffer(sc.ubo_bytes(), np.float32)
    assert blob.size == 6656 and blob[48 * 20:5120].max() == 0.0 and blob[5120 + 7 * 24:].max() == 0.0
    assert np.array_equal(blob[0:4], np.array([-12.0, -11.2, -5.0, 1.3], np.float32))


def test_stress_and_glass_scenes(pkg):
    st = pkg.scene.stress_scene(256)
    assert st.num_spheres == 256 and st.num_cuboids == 7
    pos = np.array([s.position for s in st.spheres])
    assert pos[:, 0].min() > -20 and pos[:, 0].max() < 20 and pos[:, 1].min() > -12.5 and pos[:, 2].max() < 2.5
    assert st.ubo_bytes() == pkg.scene.stress_scene(256).ubo_bytes()  # deterministic
    gl = pkg.scene.glass_scene()
    assert all(s.material.ior == np.float32(1.5) and s.material.refraction_chance == np.float32(0.98) for s in gl.spheres)


def test_camera_blobs(pkg):
    cam = pkg.camera.Camera()
    assert abs(np.linalg.norm(cam.view_dir) - 1) < 1e-6
    blob = np.frombuffer(pkg.camera.basic_data_ubo(cam, 1920, 1080), np.float32)
    assert blob.size == 36 and np.allclose(blob[32:35], [-17.14, 3.53, -8.62])
    inv_view = blob[16:32].reshape(4, 4).astype(np.float64)
    assert np.allclose(inv_view @ cam.view.astype(np.float64), np.eye(4), atol=1e-5)
    assert np.allclose(inv_view[3, :3], cam.position, atol=1e-5)          # row-vector convention: translation in row 3
    inv_proj = blob[0:16].reshape(4, 4).astype(np.float64)
    proj = pkg.camera.perspective_fov(pkg.camera.degrees_to_radians(103.0), 1920 / 1080, 0.005, 1000.0).astype(np.float64)
    assert np.allclose(inv_proj @ proj, np.eye(4), atol=1e-3)
    assert abs(proj[1, 1] - 1.0 / np.tan(np.radians(103.0) / 2)) < 1e-5   # vertical FOV
    atmo = np.frombuffer(pkg.camera.atmospheric_data_ubo(), np.float32)
    assert atmo.size == 116
    lp = pkg.camera.atmosphere_light_pos(0.5)
    assert lp[2] < -1.49e11 and abs(lp[1]) < 1e5                         # sun on the -Z horizon at time 0.5


def test_env_generators(pkg):
    e = pkg.envmap
    d = e.face_directions(4)
    assert d.shape == (6, 4, 4, 3) and np.allclose(np.linalg.norm(d, axis=-1), 1)
    assert d[0, ..., 0].min() > 0 and d[1, ..., 0].max() < 0 and d[2, ..., 1].min() > 0 and d[5, ..., 2].max() < 0
    f = e.synthetic_sky_rgba32f(8)
    assert f.dtype == np.float32 and f.shape == (6, 8, 8, 4) and (f[..., 3] == 1).all() and f.min() >= 0
    s = e.synthetic_sky_srgb8(8)
    assert s.dtype == np.uint8 and (s[..., 3] == 255).all()
    t = e.tiny_test_cube(2)
    assert len(np.unique(t[..., 0])) == 24


def test_row_blocks_cover_image(pkg):
    from opentk_pathtracer_amd import distributed as D
    for h, g in [(1080, 1), (1080, 2), (1080, 8), (2160, 8), (75, 4), (7, 8)]:
        blocks = [D.row_block(h, r, g) for r in range(g)]
        assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == h
        for a, b in zip(blocks, blocks[1:]):
            assert a[0] + a[1] == b[0]
        assert max(b[1] for b in blocks) - min(b[1] for b in blocks) <= 1
        assert D.max_rows(h, g) == max(b[1] for b in blocks)
    assert D.row_block(2160, 3, 8) == (810, 270)
    # block-cyclic ownership (pt_set_interleaved_tile): every row exactly once, shares differ by at most one band
    for h, g, band in [(1080, 8, 16), (3060, 8, 16), (1530, 2, 8), (75, 4, 8), (2160, 3, 24)]:
        shares = [D.interleaved_rows(h, r, g, band) for r in range(g)]
        allrows = sorted(y for sh in shares for y in sh)
        assert allrows == list(range(h))
        assert max(len(sh) for sh in shares) - min(len(sh) for sh in shares) <= band
        assert D.max_interleaved_rows(h, g, band) == max(len(sh) for sh in shares)
        assert shares[1][:band] == list(range(band, min(2 * band, h)))


# ------------------------------------------------------------------------------------------------ C ABI (no GPU here)
def test_c_abi_exports_every_declared_symbol(pkg, native_lib):
    declared = pkg.native.declared_symbols()
    assert len(declared) >= 35
    raw = C.CDLL(pkg.native.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/mi355pt.h but not exported"
    assert b"mi355pt" in native_lib.pt_version()


def test_c_abi_has_no_cpu_fallback(pkg, native_lib):
    """Without a HIP device the library must fail loudly, not compute on the CPU."""
    if native_lib.pt_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = native_lib.pt_create(0, 64, 64, C.byref(h))
    assert rc == pkg.native.PT_E_NO_DEVICE and not h.value
    assert b"no CPU fallback" in native_lib.pt_last_error(None)
    with pytest.raises(pkg.native.NativeError):
        pkg.PathTracer(None, 8, 8, 1, 1, 1.0, 0.0)


def test_c_abi_argument_checks_without_device(pkg, native_lib):
    assert native_lib.pt_create(0, 0, 64, C.byref(C.c_void_p())) == pkg.native.PT_E_BAD_ARGUMENT
    assert native_lib.pt_create(0, 64, 64, None) == pkg.native.PT_E_BAD_ARGUMENT
    assert native_lib.pt_destroy(None) == pkg.native.PT_E_BAD_HANDLE
    assert native_lib.pt_render(None, None) == pkg.native.PT_E_BAD_HANDLE
    # round 2: limits and the group constructor reject bad arguments before touching a device
    N = pkg.native
    assert native_lib.pt_create(0, 64, N.PT_MAX_IMAGE_DIM + 1, C.byref(C.c_void_p())) == N.PT_E_OUT_OF_RANGE
    assert native_lib.pt_create_multi(None, 2, 64, 64, C.byref(C.c_void_p())) == N.PT_E_BAD_ARGUMENT
    ids = (C.c_int * 3)(0, 1, 2)
    assert native_lib.pt_create_multi(ids, 0, 64, 64, C.byref(C.c_void_p())) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_create_multi(ids, 17, 64, 64, C.byref(C.c_void_p())) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_create_multi(ids, 3, 64, 2, C.byref(C.c_void_p())) == N.PT_E_BAD_ARGUMENT
    assert native_lib.pt_present_rgba8_async(None, 0) == N.PT_E_BAD_HANDLE
    assert native_lib.pt_present_wait(None, 0, None, None, None) == N.PT_E_BAD_HANDLE
    assert native_lib.pt_multi_set_partition(None, 16) == N.PT_E_BAD_HANDLE
    if native_lib.pt_device_count() == 0:  # the group constructor has no CPU fallback either
        assert native_lib.pt_create_multi(ids, 3, 64, 64, C.byref(C.c_void_p())) == N.PT_E_NO_DEVICE


def test_product_never_imports_the_oracle(pkg):
    """Rule: only tests/, smoke() and bench's cpu_baseline may import, link or execute anything under oracle/.
    (Comments may mention it; code may not reach it.)"""
    import re
    code_refs = re.compile(r"(^\s*(from|import)\s+.*oracle)|(#\s*include\s+[\"<][^\n]*oracle)|(CDLL\([^\n]*oracle)|"
                           r"(load_oracle)|(libpt_oracle)|(dlopen\([^\n]*oracle)", re.M)
    for dirpath, _, files in os.walk(pkg.native.HERE):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not code_refs.search(text), f"{f} reaches into oracle/"
    # and the built product library has no dependency on the oracle library
    import subprocess
    out = subprocess.run(["ldd", pkg.native.LIB_PATH], capture_output=True, text=True).stdout
    assert "pt_oracle" not in out


# ------------------------------------------------------------------------------------------------ C++ host mirror
def test_cpp_host_mirror_packs_the_same_bytes(pkg, native_lib, tmp_path):
    """opentk-pathtracer_amd/host/pt_host.hpp restates Material/Sphere/Cuboid/LoadScene/Camera in C++ (the reference
    host is compiled C#); its GameObjectsUBO image must equal the Python harness's byte for byte, its camera blob
    to float rounding."""
    import subprocess
    demo = pkg.native.build_host_demo()
    scene_bin, cam_bin = tmp_path / "scene.bin", tmp_path / "cam.bin"
    out = subprocess.run([demo, "dump-scene", str(scene_bin)], capture_output=True, text=True, check=True).stdout
    assert "48 spheres, 7 cuboids" in out
    assert scene_bin.read_bytes() == pkg.scene.default_scene().ubo_bytes()
    subprocess.run([demo, "dump-camera", "1920", "1080", str(cam_bin)], check=True)
    cpp = np.frombuffer(cam_bin.read_bytes(), np.float32)
    py = np.frombuffer(pkg.camera.basic_data_ubo(pkg.camera.Camera(), 1920, 1080), np.float32)
    assert cpp.size == 36 and np.allclose(cpp, py, rtol=0, atol=2e-6)
    # without a GPU the demo's render mode must fail loudly with the library's message (Program.cs:15-25 catch-all)
    if native_lib.pt_device_count() == 0:
        r = subprocess.run([demo, "render", "64", "64", "1", str(tmp_path / "x.f32")], capture_output=True, text=True)
        assert r.returncode == 3 and "no CPU fallback" in r.stderr


# ------------------------------------------------------------------------------------------------ checkpoints / screenshots
class _FakeTracer:
    """Just the attributes checkpoint.py touches — the file logic is host code, testable without a GPU."""

    def __init__(self, w=7, h=5, rows=5, y0=0, frame=3):
        self.Width, self.Height, self.y0, self.rows = w, h, y0, rows
        self.band_rows, self.band_world, self.band_rank = 0, 1, 0
        self.RayDepth, self.SPP, self.FocalLength, self.ApertureDiameter = 8, 1, 20.0, 0.14
        self.FrameIndex = frame
        rng = np.random.RandomState(1)
        self.Result = rng.rand(rows, w, 4).astype(np.float32)
        self.Result[..., 3] = 1.0
        self.written = None

    def WriteResult(self, img, frame):
        self.written = (img.copy(), frame)

    def Present(self):
        return (np.clip(self.Result, 0, 1) * 255).astype(np.uint8)


def test_checkpoint_file_round_trip_and_validation(pkg, tmp_path):
    ck = pkg.checkpoint
    a = _FakeTracer()
    path = str(tmp_path / "a.ptck")
    ck.save_checkpoint(path, a)
    assert os.path.getsize(path) == 56 + 5 * 7 * 16
    b = _FakeTracer(frame=0)
    hdr = ck.load_checkpoint(path, b)
    assert hdr["frame_index"] == 3 and b.written[1] == 3
    assert np.array_equal(b.written[0].view(np.uint32), a.Result.view(np.uint32))
    with pytest.raises(ck.CheckpointError, match="width"):
        ck.load_checkpoint(path, _FakeTracer(w=8))
    other = _FakeTracer()
    other.RayDepth = 13
    with pytest.raises(ck.CheckpointError, match="ray_depth"):
        ck.load_checkpoint(path, other)
    assert ck.load_checkpoint(path, other, strict=False)["ray_depth"] == 8
    open(str(tmp_path / "bad.ptck"), "wb").write(b"NOTACKPT" + bytes(48))
    with pytest.raises(ck.CheckpointError, match="magic"):
        ck.read_checkpoint_file(str(tmp_path / "bad.ptck"))
    # block-cyclic ownership: every rank has y0 = 0 and (often) the same row count — rank, world size and band height
    # are what identify the rows
    banded = _FakeTracer()
    banded.band_rows, banded.band_world, banded.band_rank = 8, 2, 1
    bpath = str(tmp_path / "banded.ptck")
    ck.save_checkpoint(bpath, banded)
    same = _FakeTracer()
    same.band_rows, same.band_world, same.band_rank = 8, 2, 1
    assert ck.load_checkpoint(bpath, same)["band_rank"] == 1
    for attr, val, what in (("band_rank", 0, "band_rank"), ("band_world", 4, "band_world"), ("band_rows", 16, "band_rows"),
                            ("band_rows", 0, "band_rows")):
        wrong = _FakeTracer()
        wrong.band_rows, wrong.band_world, wrong.band_rank = 8, 2, 1
        setattr(wrong, attr, val)
        with pytest.raises(ck.CheckpointError, match=what):
            ck.load_checkpoint(bpath, wrong)
    with pytest.raises(ck.CheckpointError, match="band_rows"):
        ck.load_checkpoint(path, same)  # a whole-image checkpoint into a banded renderer
    data = open(path, "rb").read()
    open(str(tmp_path / "short.ptck"), "wb").write(data[:-16])
    with pytest.raises(ck.CheckpointError, match="payload"):
        ck.read_checkpoint_file(str(tmp_path / "short.ptck"))


def test_screenshot_png_is_flipped_like_the_reference(pkg, tmp_path):
    """Framebuffer.cs:79 flips the GL image (row 0 = bottom) when saving; the PNG must start with the TOP row."""
    ck = pkg.checkpoint
    t = _FakeTracer(w=9, h=4, rows=4)
    path = str(tmp_path / "s.png")
    ck.save_screenshot(path, t)
    rgb = ck.decode_png_rgb8(open(path, "rb").read())
    assert rgb.shape == (4, 9, 3)
    assert np.array_equal(rgb, t.Present()[::-1, :, :3])
    assert np.array_equal(ck.decode_png_rgb8(ck.encode_png(t.Present(), flip_vertically=False)), t.Present()[..., :3])


def test_bench_refuses_to_report_fewer_gpus_than_asked(tmp_path):
    """`python bench.py --gpus N` with no torchrun environment must never degrade to a smaller run silently (round 2 did): without N HIP
    devices it exits non-zero with a message and prints no JSON line.  (This container has no GPU at all: N = 2 and N = 1 both refuse.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for n in ("2", "1"):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", n, "--steps", "4", "--warmup", "1"], capture_output=True, text=True,
                           timeout=300, env=env)
        import ctypes
        have = ctypes.CDLL(os.path.join(root, "opentk-pathtracer_amd", "libmi355pt.so")).pt_device_count()
        if have >= int(n):
            continue  # (a GPU box: nothing to refuse)
        assert p.returncode != 0 and "refusing" in (p.stdout + p.stderr), p.stderr[-500:]
        assert not any(line.startswith("{") for line in p.stdout.splitlines())
