"""Build-container-only checks (skipped wherever /root/reference or Mesa llvmpipe is absent, e.g. on the GPU box):
re-run the reference's own GLSL live through oracle/_ref/glsl_runner and confirm that
  * the committed fixtures are reproducible bit-for-bit from the reference (they are data derived from it, not from us),
  * the oracle agrees with a FRESH reference run on inputs that are in no fixture (random camera, random materials).
"""
import importlib.util
import os

import numpy as np
import pytest

import configs
import fixtures
import tolerances as tol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("glsl_run", os.path.join(ROOT, "oracle", "glsl_ref", "run.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

pytestmark = pytest.mark.skipif(not ref.available(), reason="needs /root/reference + Mesa llvmpipe + oracle/_ref/glsl_runner")


def test_committed_fixture_is_reproducible_from_the_reference():
    fx = fixtures.load("frame_default_128x72_d8")
    out = ref.run_pathtracer(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=fx["frames"],
                             dump_each=True, **fixtures.kwargs(fx))
    assert np.array_equal(out[fx["frame_indices"]][..., :3].view(np.uint32), fx["expected"].view(np.uint32))


def test_reference_is_deterministic_across_thread_counts():
    w = configs.SMALL_FRAMES[0]
    sc, basic, objs, env, kw = configs.inputs(w)
    a = ref.run_pathtracer(w.width, w.height, basic, objs, env, threads=1, **kw)
    b = ref.run_pathtracer(w.width, w.height, basic, objs, env, threads=4, **kw)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_vs_fresh_reference_run_on_unseen_inputs(oracle, pkg, seed):
    """Random camera pose inside the room + random materials on every object, random depth/spp/aperture."""
    rng = np.random.RandomState(seed)
    s = pkg.scene
    sc = s.default_scene()
    for o in sc.objects():
        if rng.rand() < 0.6:
            o.material = s.Material(albedo=rng.rand(3), emissiv=rng.rand(3) * (rng.rand() < 0.15), absorbance=rng.rand(3),
                                    specular_chance=rng.rand() * 0.6, specular_roughness=rng.rand(), ior=1 + rng.rand() * 0.6,
                                    refraction_chance=rng.rand() * 0.4, refraction_roughness=rng.rand() * 0.5)
    cam = pkg.camera.Camera(position=rng.uniform([-18, -10, -20], [18, 10, 0]), look_x=rng.uniform(-180, 180),
                            look_y=rng.uniform(-60, 60))
    W, H = 112, 64
    basic = pkg.camera.basic_data_ubo(cam, W, H)
    env = configs.load_env("sky_f32_32")
    kw = dict(num_spheres=48, num_cuboids=7, ray_depth=int(rng.randint(2, 20)), spp=int(rng.randint(1, 4)),
              focal_length=float(rng.uniform(5, 30)), aperture=float(rng.uniform(0, 0.5)))
    r = ref.run_pathtracer(W, H, basic, sc.ubo_bytes(), env, num_frames=2, **kw)[0][..., :3]
    o = oracle.render(W, H, basic, sc.ubo_bytes(), env, num_frames=2, **kw)[..., :3]
    both_nan = np.isnan(r).any(-1) & np.isnan(o).any(-1)
    ok = tol.within(r, o) | both_nan
    assert ok.mean() >= tol.PIXEL_FRACTION, f"seed {seed}: {100 * ok.mean():.2f}% within tolerance"
    fin = np.isfinite(r).all(-1) & np.isfinite(o).all(-1)
    assert abs(r[fin].mean() - o[fin].mean()) <= 5 * tol.MEAN_REL_TOL * abs(r[fin].mean())


def test_edge_scene_differences_are_nan_direction_environment_lookups():
    """The one scene whose accumulation does NOT converge onto the reference's within its sampling error (max |z| 13, image mean off by
    6e-4 after 4,096 frames) is the quirk scene: a glass cuboid and a camera-containing glass sphere with zero roughness produce total
    internal reflection -> refract() = 0 -> normalize(vec3(0)) = NaN direction (compute.glsl:210-211), and the path ends in
    texture(SamplerEnvironment, NaN) (compute.glsl:177) — undefined in GL.  llvmpipe returns a deterministic texel average that depends
    on the NaN's sign bit (probed with a test main); the contract clamps the coordinate.  This test shows that this is ALL there is to
    it: per sample, every gross difference between the oracle and the reference either ends in such a lookup (flagged by the oracle's
    diagnostic build) or belongs to the usual branch-flip budget (< 0.03 % of the samples)."""
    import __graft_entry__ as graft
    O = graft.load_oracle()
    plain, marked = O.Oracle(), O.Oracle(mark_nan_env=True)
    w = configs.Workload("edge_64x36_d16", "edge", 64, 36, 16, "sky_f32_32")
    sc, basic, objs, env, kw = configs.inputs(w)
    n = 96
    acc = [x(w.width, w.height, basic, objs, env, num_frames=n, dump_each=True, **kw)[..., :3].astype(np.float64)
           for x in (ref.run_pathtracer, plain.render, marked.render)]

    def samples(a):  # per-frame samples from the running means
        s = np.empty_like(a)
        s[0] = a[0]
        for k in range(1, len(a)):
            s[k] = (k + 1) * a[k] - k * a[k - 1]
        return s
    sr, so, sm = (samples(a) for a in acc)
    gross = np.abs(so - sr).max(-1) > 1e-2 * np.maximum(1.0, np.abs(sr).max(-1))
    nan_env = np.abs(sm - so).max(-1) > 10.0
    assert nan_env.sum() > 100, "the scene no longer exercises the quirk"
    assert (gross & ~nan_env).sum() <= 3e-4 * gross.size, (int(gross.sum()), int(nan_env.sum()), int((gross & ~nan_env).sum()))
    # without those samples the two accumulations agree like every other scene's
    so_masked, sr_masked = np.where(nan_env[..., None], 0.0, so), np.where(nan_env[..., None], 0.0, sr)
    assert abs(so_masked.mean() - sr_masked.mean()) <= 2e-4 * sr_masked.mean()
