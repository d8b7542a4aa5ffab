"""Size-independent properties of the integrator, checked on the CPU oracle (the GPU versions of the same properties
live in test_gpu_parity.py)."""
import numpy as np

import configs


def _inputs(name="default", W=96, H=54, depth=6, env="sky_f32_32", **kw):
    w = configs.Workload("t", name, W, H, depth, env, **kw)
    return w, configs.inputs(w)


def test_deterministic_and_thread_count_independent(oracle):
    w, (sc, basic, objs, env, kw) = _inputs()
    a = oracle.render(w.width, w.height, basic, objs, env, threads=1, **kw)
    b = oracle.render(w.width, w.height, basic, objs, env, threads=5, **kw)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert (a[..., 3] == 1.0).all() and np.isfinite(a).all()


def test_tiled_equals_untiled_bit_exact(oracle):
    """seeds and NDC use global coordinates, so any row tiling reproduces the untiled image."""
    w, (sc, basic, objs, env, kw) = _inputs(H=50)
    full = oracle.render(w.width, w.height, basic, objs, env, num_frames=2, **kw)
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            y0, y1 = r * w.height // world, (r + 1) * w.height // world
            parts.append(oracle.render(w.width, w.height, basic, objs, env, num_frames=2, y0=y0, rows=y1 - y0, **kw))
        assert np.array_equal(np.concatenate(parts).view(np.uint32), full.view(np.uint32))


def test_render_pixels_equals_full_frame(oracle):
    w, (sc, basic, objs, env, kw) = _inputs()
    full = oracle.render(w.width, w.height, basic, objs, env, **kw)
    rng = np.random.RandomState(0)
    xy = np.stack([rng.randint(0, w.width, 200), rng.randint(0, w.height, 200)], 1)
    px = oracle.render_pixels(w.width, w.height, basic, objs, env, xy, **kw)
    assert np.array_equal(px.view(np.uint32), full[xy[:, 1], xy[:, 0]].view(np.uint32))


def test_running_mean_accumulation(oracle):
    """compute.glsl:126-128: frame f contributes with weight 1/(f+1) — the accumulated image equals the mix() chain of
    the single frames, and frame 0 ignores the previous contents (even garbage)."""
    w, (sc, basic, objs, env, kw) = _inputs(W=64, H=36)
    acc = oracle.render(w.width, w.height, basic, objs, env, num_frames=3, **kw)
    singles = [oracle.render(w.width, w.height, basic, objs, env, frame_start=f, num_frames=1,
                             image=np.zeros((w.height, w.width, 4), np.float32), **kw) for f in range(3)]
    # frames rendered on a zero image at index f are scaled by 1/(f+1): undo, then rebuild the chain in float32
    f32 = np.float32
    new = [singles[0][..., :3], singles[1][..., :3] * f32(2), singles[2][..., :3] * f32(3)]
    chain = new[0]
    for f in (1, 2):
        wgt = f32(1.0) / f32(f + 1)
        chain = chain * (f32(1.0) - wgt) + new[f] * wgt
    assert np.allclose(chain, acc[..., :3], rtol=2e-6, atol=1e-7)
    garbage = np.full((w.height, w.width, 4), 1e30, np.float32)
    again = oracle.render(w.width, w.height, basic, objs, env, num_frames=1, image=garbage, **kw)
    clean = oracle.render(w.width, w.height, basic, objs, env, num_frames=1, **kw)
    assert np.array_equal(again.view(np.uint32), clean.view(np.uint32))


def test_empty_scene_is_environment_only(oracle, pkg):
    w, (sc, basic, objs, env, kw) = _inputs("empty", depth=3, aperture=0.0)
    img, st = oracle.render(w.width, w.height, basic, objs, env, want_stats=True, **kw)
    assert st["bounces"] == st["samples"] == st["env_lookups"] == w.width * w.height
    assert img[..., :3].min() >= 0 and img[..., :3].max() <= float(env[..., :3].max()) * 1.0001


def test_zero_depth_is_black(oracle):
    w, (sc, basic, objs, env, kw) = _inputs(depth=0)
    img = oracle.render(w.width, w.height, basic, objs, env, **kw)
    assert (img[..., :3] == 0).all() and (img[..., 3] == 1).all()


def test_stats_and_flop_model_inputs(oracle):
    w, (sc, basic, objs, env, kw) = _inputs(depth=8)
    _, st = oracle.render(w.width, w.height, basic, objs, env, want_stats=True, **kw)
    assert st["samples"] == w.width * w.height
    assert st["sphere_tests"] == st["bounces"] * 48 and st["cuboid_tests"] == st["bounces"] * 7
    assert 1.0 <= st["bounces"] / st["samples"] <= 8.0


def test_entry_distance_quirk(oracle):
    """compute.glsl:234: `t1 < hitInfo.T` compares the ENTRY distance: a sphere containing the origin (t1 < 0)
    replaces a nearer hit found earlier.  Sphere 0 is near and in front; sphere 1 is huge and contains the origin."""
    hit0 = oracle.ray_sphere([0, 0, 0], [0, 0, -1], [0, 0, -3, 1])
    hit1 = oracle.ray_sphere([0, 0, 0], [0, 0, -1], [0, 0, 0, 50])
    assert hit0 == (True, 2.0, 4.0) and hit1[0] and hit1[1] == -50.0 and hit1[2] == 50.0
    # acceptance order: T after sphere 0 is 2.0; sphere 1 has t1 = -50 < 2.0 -> accepted with T = t2 = 50 (from inside)


def test_spp_continues_the_rng_stream(oracle):
    """compute.glsl:110: the stream is NOT re-seeded between samples of a frame -> spp=2 is not the mean of two frames."""
    w, (sc, basic, objs, env, kw) = _inputs(W=48, H=27)
    kw2 = dict(kw, spp=2)
    a = oracle.render(w.width, w.height, basic, objs, env, **kw2)
    one = oracle.render(w.width, w.height, basic, objs, env, **kw)
    assert np.isfinite(a).all() and not np.array_equal(a, one)
    assert abs(a[..., :3].mean() - one[..., :3].mean()) < 0.15 * one[..., :3].mean()


def _ulp_error(got, exact64):
    """|got - exact| in units of the float32 ulp at `exact`"""
    exact32 = exact64.astype(np.float32)
    ulp = np.spacing(np.abs(exact32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - exact64) / ulp


def test_contract_reciprocal_root_accuracy(oracle):
    """The pt-f32 software 1/x, 1/sqrt(x), sqrt(x) (oracle/pt_oracle.c header) stay inside the bounds the contract
    states — far inside what GLSL 4.50 section 4.7.1 allows (2.5 ulp for a/b, 2 ulp for inversesqrt)."""
    rng = np.random.RandomState(7)
    x = np.exp(rng.uniform(np.log(1e-30), np.log(1e30), 2_000_000)).astype(np.float32)
    x = np.concatenate([x, rng.uniform(0.0, 4.0, 1_000_000).astype(np.float32) + np.float32(1e-6)])
    x64 = x.astype(np.float64)
    assert _ulp_error(oracle.rcp(x), 1.0 / x64).max() <= 0.52
    assert _ulp_error(oracle.rcp(-x), -1.0 / x64).max() <= 0.52
    assert _ulp_error(oracle.rsqrt(x), 1.0 / np.sqrt(x64)).max() <= 1.75
    assert _ulp_error(oracle.sqrt(x), np.sqrt(x64)).max() <= 0.502
    # special values are part of the contract
    sp = oracle.sqrt(np.array([0.0, 1.0, 4.0, np.inf, -1.0, np.nan], np.float32))
    assert sp[0] == 0.0 and sp[1] == 1.0 and sp[2] == 2.0 and not np.isfinite(sp[3:]).any()
    assert np.isinf(oracle.rcp(np.array([0.0], np.float32)))[0] and np.isinf(oracle.rsqrt(np.array([0.0], np.float32)))[0]
    assert np.isnan(oracle.rsqrt(np.array([-1.0], np.float32)))[0]


def test_thread_count_never_changes_the_image(oracle):
    """The row-parallel driver (oracle/pt_oracle.c, pto_render_frame) must render every row exactly once whatever thread count is
    asked for — more threads than rows (256 on the GPU box for an 8-row fuzz image), one thread, an odd count — and a worker that
    cannot be started must not leave its rows unrendered (round 3: thread creation is checked, the caller renders those rows)."""
    import numpy as np
    import configs
    w = configs.Workload("threads", "default", 24, 8, 4, "sky_f32_32", frames=3)
    sc, basic, objs, env, kw = configs.inputs(w)
    ref = oracle.render(w.width, w.height, basic, objs, env, num_frames=w.frames, threads=1, **kw)
    for threads in (2, 3, 8, 64, 256, 1000):
        img = oracle.render(w.width, w.height, basic, objs, env, num_frames=w.frames, threads=threads, **kw)
        assert (img.view(np.uint32) == ref.view(np.uint32)).all(), threads
