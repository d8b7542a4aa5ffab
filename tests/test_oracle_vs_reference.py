"""Pin the CPU oracle (oracle/pt_oracle.c) against outputs of the REFERENCE ITSELF: the committed fixtures under
tests/golden/ were produced by running the reference's unmodified GLSL on Mesa llvmpipe
(tests/golden/make_golden.py).  CPU-only; the tolerances are stated in tests/tolerances.py.
"""
import numpy as np
import pytest

import fixtures
import tolerances as tol

pytestmark = pytest.mark.filterwarnings("ignore")


# ------------------------------------------------------------------------------------------------ function level
@pytest.fixture(scope="module")
def micro():
    return fixtures.load("micro")


def test_rng_stream_is_bit_exact(oracle, micro):
    """compute.glsl:106,334-344 — seed formula, PCG hash, uint->float RNE conversion: integers, so bit-exact."""
    ref = micro["rng"]  # (64, 64, 4): 4 successive draws per pixel, frame 5
    frame = int(micro["frame"])
    for (x, y) in [(0, 0), (1, 0), (0, 1), (3, 5), (63, 63), (17, 42), (31, 7)]:
        got = oracle.rand_stream(oracle.pixel_seed(x, y, frame), 4)
        assert got.view(np.uint32).tolist() == ref[y, x].view(np.uint32).tolist(), (x, y)
    # whole image, vectorised re-implementation of the oracle's call (every pixel)
    for y in range(0, 64, 7):
        for x in range(64):
            got = oracle.rand_stream(oracle.pixel_seed(x, y, frame), 4)
            assert np.array_equal(got.view(np.uint32), ref[y, x].view(np.uint32))
    assert ref.min() >= 0.0 and ref.max() <= 1.0


def test_pcg_hash_values_bit_exact(oracle, micro):
    ref = micro["hash"].view(np.uint32)  # h>>8 bit-cast through float (denormal-safe: < 2^24)
    frame = int(micro["frame"])
    for y in range(0, 64, 5):
        for x in range(0, 64, 3):
            seed = (x * 2654435761 + y * 40503 + frame) & 0xFFFFFFFF
            h = oracle.hash_stream(seed, 3)
            assert [int(v) >> 8 for v in h] == ref[y, x, :3].tolist()


def test_known_answer_seed_precedence(oracle):
    """3*1973 + 5*9277 + 7*2699 | 1 = 71197 ('+' binds tighter than '|')."""
    assert oracle.pixel_seed(3, 5, 7) == 71197
    assert oracle.pixel_seed(0, 0, 0) == 1
    assert oracle.pixel_seed(1919, 1079, 63) == ((1919 * 1973 + 1079 * 9277 + 63 * 2699) | 1) & 0xFFFFFFFF


def test_uint_to_float_rounding(oracle):
    """float(4294967295u)/2^32 == 1.0 on the reference (appendix B) -> rand in [0, 1] inclusive."""
    import ctypes
    # find a seed state whose hash is >= 0xFFFFFF80 is impractical; check the conversion rule the oracle uses instead
    assert np.float32(np.uint32(4294967295)) * np.float32(2.0 ** -32) == np.float32(1.0)
    assert np.float32(np.uint32(4294967167)) * np.float32(2.0 ** -32) == np.float32(0.99999994)
    assert np.float32(np.uint32(16777217)) == np.float32(16777216.0)


def test_sphere_intersection(oracle, micro):
    """compute.glsl:261-277 on sphere (0.25,0.5,-1) r=1.5: rays from outside, inside and near-tangent."""
    ref = micro["sphere"]
    bad = 0
    for y in range(64):
        o = np.array([0.25 * (y // 16), 0.5, 4.0 - 1.25 * (y % 16) * 0.25], np.float32)
        for x in range(64):
            d = oracle.normalize([(x - 31.5) / 24.0, ((y % 16) - 7.5) / 24.0, -1.0])
            hit, t1, t2 = oracle.ray_sphere(o, d, [0.25, 0.5, -1.0, 1.5])
            r = ref[y, x]
            if bool(r[2]) != hit:
                bad += 1  # grazing rays: the discriminant's sign may differ by rounding
                continue
            if hit:
                # near-tangent rays: sqrt amplifies the last-bit difference of the direction
                assert abs(t1 - r[0]) <= 2e-3 and abs(t2 - r[1]) <= 2e-3
            else:
                assert t1 == r[0] == np.float32(3.4028235e38) and t2 == r[1]
    assert bad <= 4


def test_cuboid_intersection_and_normal(oracle, micro):
    """compute.glsl:280-294,322-332 on cuboid centre (0,0.25,-2) dims (2,1.5,1), incl. axis-parallel rays (inf slabs)."""
    ref = micro["cuboid"]
    mn, mx = np.array([-1.0, -0.5, -2.5], np.float32), np.array([1.0, 1.0, -1.5], np.float32)
    mism = 0
    for y in range(64):
        o = np.array([0.3 * (y // 16) - 0.5, 0.25, 4.0 - (y % 16) * 0.4], np.float32)
        for x in range(64):
            d = oracle.normalize([(x - 31.5) / 20.0, ((y % 16) - 7.5) / 20.0, -1.0])
            if x == 0:
                d = np.array([0.0, 0.0, -1.0], np.float32)
            if x == 1:
                d = np.array([1.0, 0.0, 0.0], np.float32)
            hit, t1, t2 = oracle.ray_cuboid(o, d, mn, mx)
            r = ref[y, x]
            if bool(r[2]) != hit:
                mism += 1
                continue
            if np.isfinite(r[0]) and np.isfinite(r[1]) and abs(r[0]) < 1e30 and abs(r[1]) < 1e30:
                assert abs(t1 - r[0]) <= 1e-4 * max(1.0, abs(r[0])) and abs(t2 - r[1]) <= 1e-4 * max(1.0, abs(r[1]))
            if hit:
                T = t2 if t1 < 0 else t1
                n = oracle.cuboid_normal(mn, mx, o + d * np.float32(T))
                code = n[0] + 2.0 * n[1] + 4.0 * n[2]
                if np.isnan(r[3]):
                    assert np.isnan(code)  # normalize(0) = NaN on both (off-face point)
                elif not np.isnan(code):
                    mism += abs(code - r[3]) > 1e-3  # edge hits may pick a different face within EPSILON
    assert mism <= 12


def test_builtin_accuracy(oracle, micro):
    """sin/cos on [0,2pi], exp(-x), pow(x,5): the oracle's fixed polynomials vs llvmpipe's built-ins."""
    ref = micro["math"]
    idx = np.arange(4096, dtype=np.float32)
    u = (idx + np.float32(0.5)) / np.float32(4096.0)
    a = (u * np.float32(2.0) * np.float32(3.14159265)).astype(np.float32)
    r = ref.reshape(4096, 4)
    worst = np.zeros(4)
    for i in range(0, 4096, 3):
        s, c = oracle.sincos(a[i])
        e = oracle.exp(np.float32(-u[i]) * np.float32(12.0))
        p = oracle.pow5(u[i] * np.float32(2.0))
        got = np.array([s, c, e, p])
        err = np.abs(got - r[i]) / np.array([1.0, 1.0, max(abs(r[i, 2]), 1e-30), max(abs(r[i, 3]), 1e-30)])
        worst = np.maximum(worst, err)
        # and against double-precision truth: the contract's routines are ~1 ulp
        assert abs(s - np.sin(np.float64(a[i]))) < 2.5e-7 and abs(c - np.cos(np.float64(a[i]))) < 2.5e-7
        assert abs(e / np.exp(np.float64(np.float32(-u[i]) * np.float32(12.0))) - 1.0) < 3e-7
    assert worst[0] < 5e-7 and worst[1] < 5e-7      # abs (llvmpipe: ~6e-8)
    assert worst[2] < 5e-6 and worst[3] < 1e-5      # rel (llvmpipe exp/pow are ~20 ulp)


def test_cosine_sample_hemisphere(oracle, micro):
    ref = micro["cosine"]
    frame = int(micro["frame"])
    worst = 0.0
    for y in range(0, 64, 3):
        for x in range(0, 64, 3):
            n = oracle.normalize([x - 31.5, y - 31.5, 9.0])
            d, _ = oracle.cosine_sample_hemisphere(n, oracle.pixel_seed(x, y, frame))
            norm = np.linalg.norm(ref[y, x, :3] - d)
            # normalize(n + v) is ill-conditioned when n + v ~ 0: scale the tolerance by the conditioning
            cond = 1.0 / max(1e-3, float(ref[y, x, 3]))
            worst = max(worst, norm / cond)
            assert norm <= 2e-6 * cond + 1e-6, (x, y, norm)
            assert abs(np.linalg.norm(d) - 1.0) < 1e-6


def test_fresnel_and_refract(oracle, micro):
    ref = micro["fresnel"]
    for y in range(0, 64, 2):
        ior = np.float32(1.0) + np.float32(y) / np.float32(63.0)
        for x in range(0, 64, 3):
            cosT = np.float32(x) / np.float32(63.0)
            f = oracle.fresnel_schlick(cosT, 1.0, ior)
            assert abs(f - ref[y, x, 0]) <= 2e-6 + 1e-5 * abs(ref[y, x, 0])
            i = np.array([np.sqrt(np.float32(1.0) - cosT * cosT), -cosT, 0.0], np.float32)
            eta = ior if y % 2 == 0 else np.float32(1.0) / ior
            r = oracle.refract(i, [0.0, 1.0, 0.0], eta)
            rr = ref[y, x, 1:4]
            if np.all(rr == 0.0) or np.all(r == 0.0):
                # total internal reflection returns vec3(0); at the critical angle the sign of k may differ
                assert np.all(np.abs(r - rr) < 2e-3) or (np.all(rr == 0.0) and np.all(r == 0.0))
            else:
                assert np.all(np.abs(r - rr) < 1e-3 * (1.0 + 1.0 / max(1e-3, abs(float(rr[1])))))


# ------------------------------------------------------------------------------------------------ environment sampler
@pytest.mark.parametrize("name", fixtures.names("envonly_"))
def test_environment_sampler_frames(oracle, name):
    """Empty scene + pinhole camera: every pixel is texture(SamplerEnvironment, primaryDir) — pins face selection,
    (s,t) mapping, bilinear weights, seamless edges and cube corners (tiny 2^2 / 4^2 cubes with distinct texels),
    and the per-texel sRGB decode-before-filter order."""
    fx = fixtures.load(name)
    srgb = fx["env"].dtype == np.uint8
    if srgb:
        oracle.set_srgb_lut(fixtures.llvmpipe_srgb_lut())
    try:
        got = oracle.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], **fixtures.kwargs(fx))[..., :3]
    finally:
        oracle.set_srgb_lut(None)
    ref = fx["expected"]
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= tol.ENV_REL_TOL, f"{name}: max rel err {err.max():.3g}"


def test_exact_srgb_decode_differs_from_llvmpipe_only_slightly(oracle):
    """The product uses the exact GL 4.5 sRGB decode; llvmpipe's cubic approximation is within 0.6 % of it."""
    exact = np.array([oracle.lib.pto_srgb_to_linear(i) for i in range(256)], dtype=np.float64)
    approx = fixtures.llvmpipe_srgb_lut().astype(np.float64)
    assert np.abs(exact[16:] - approx[16:]).max() < 3e-3
    assert exact[0] == 0.0 and abs(exact[255] - 1.0) < 1e-7


# ------------------------------------------------------------------------------------------------ full small frames
@pytest.mark.parametrize("name", fixtures.names("frame_"))
def test_small_frames_match_reference(oracle, name, parity_report):
    fx = fixtures.load(name)
    srgb = fx["env"].dtype == np.uint8
    if srgb:
        oracle.set_srgb_lut(fixtures.llvmpipe_srgb_lut())
    try:
        got = fixtures.oracle_frames(oracle, fx)
    finally:
        oracle.set_srgb_lut(None)
    ref = fx["expected"]
    assert got.shape == ref.shape
    for k in range(ref.shape[0]):
        st = tol.agreement(ref[k], got[k])
        # (the sRGB fixture runs with llvmpipe's own decode table injected, so it is held to the narrow band here)
        need = 0.99 if srgb else tol.min_fraction(name, k)
        parity_report(f"oracle {name} #{k}", st, need)
        frac = st["within"]
        assert frac >= need, f"{name} frame#{k}: only {100 * frac:.3f}% of pixels within tolerance, need {100 * need:.2f}%"
        fin = np.isfinite(ref[k]).all(-1) & np.isfinite(got[k]).all(-1)
        m_ref, m_got = ref[k][fin].mean(), got[k][fin].mean()
        assert abs(m_ref - m_got) <= tol.MEAN_REL_TOL * abs(m_ref), f"{name}: mean {m_got} vs {m_ref}"
        # NaN pixels (normalize(0) quirks) must be rare and mostly coincide
        assert (np.isnan(ref[k]).any(-1) ^ np.isnan(got[k]).any(-1)).mean() < 2e-3


# ------------------------------------------------------------------------------------------------ BASELINE configs, sparse
@pytest.mark.parametrize("name", fixtures.names("sparse_"))
def test_full_resolution_sparse_pixels(oracle, name, parity_report):
    """C1/C2/C3/C5 at FULL size: 4096 seeded pixels of the reference's full frame (pixels are independent)."""
    fx = fixtures.load(name)
    got = oracle.render_pixels(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], fx["xy"], frame=0,
                               **fixtures.kwargs(fx))[..., :3]
    ref = fx["expected"]
    st, need = tol.agreement(ref, got), tol.min_fraction(name)
    parity_report(f"oracle {name}", st, need)
    assert st["within"] >= need, f"{name}: {100 * st['within']:.3f}% within tolerance, need {100 * need:.2f}%"
    assert abs(ref.mean() - got.mean()) <= tol.MEAN_REL_TOL * abs(ref.mean())


# ------------------------------------------------------------------------------------------------ atmosphere
@pytest.mark.parametrize("name", fixtures.names("atmo_"))
def test_atmosphere_matches_reference(oracle, name):
    """AtmosphericScattering/compute.glsl on llvmpipe vs the oracle's restatement: smooth function, no branching
    on random numbers -> every texel within 2e-4 relative (exp is ~20 ulp on llvmpipe, x 750 accumulation steps)."""
    fx = fixtures.load(name)
    size, isteps, jsteps = (int(v) for v in fx["params"])
    got = oracle.atmosphere(size, fx["ubo"].tobytes(), fx["light_pos"], float(fx["intensity"]), isteps, jsteps)[..., :3]
    err = tol.atmo_error(fx["expected"], got)
    worst, share = tol.ATMO_MARKS[name]  # frozen per fixture (round 6; a flat 2e-3 before)
    assert err.max() <= worst, f"{name}: largest per-texel error {err.max():.3g} > frozen mark {worst:.3g}"
    assert (err < 1e-4).mean() >= share, f"{name}: {100 * (err < 1e-4).mean():.2f} % of the texels within 1e-4 < frozen mark {100 * share:.2f} %"
    assert np.median(err) < 2e-5


# ------------------------------------------------------------------------------------------------ post-process ("next" row)
def test_postprocess_matches_reference(oracle):
    """ACESFilm + LinearToInverseGamma(2.4) of PostProcessing/fragment.glsl (the reference's own functions executed on
    llvmpipe through a compute-stage test main) vs the oracle: float colour within 1e-6, RGBA8 within 1 LSB and >= 99.9 %
    identical (llvmpipe's pow is ~20 ulp; the quantisation step amplifies that only at rounding boundaries)."""
    fx = fixtures.load("post_aces_gamma")
    f, u8 = oracle.postprocess(fx["image"])
    ref = fx["expected"]
    assert np.abs(f - ref).max() <= 1e-6
    ref_u8 = (np.clip(ref, 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    diff = np.abs(ref_u8.astype(int) - u8[..., :3].astype(int))
    assert diff.max() <= 1 and (diff == 0).mean() >= 0.999
    assert (u8[..., 3] == 255).all()
    # exact anchors of the curve
    g, q = oracle.postprocess(np.array([[0.0, 1.0, 0.0031307, 1.0]], np.float32))
    assert q[0, 0] == 0 and abs(g[0, 1] - 0.908230) < 2e-6 and q[0, 1] == 232


def test_log_accuracy(oracle):
    xs = np.exp(np.random.RandomState(0).uniform(np.log(1e-6), np.log(1e4), 3000)).astype(np.float32)
    for x in xs:
        t = np.log(np.float64(x))
        assert abs(oracle.log(x) - t) <= 2e-7 * max(1.0, abs(t))


def _psnr(ref, got):
    """PSNR of tone-compressed values x/(1+x) (peak 1) — robust to the few very bright light-source pixels."""
    a, b = ref / (1.0 + ref), got / (1.0 + got)
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 10.0 * np.log10(1.0 / max(mse, 1e-20))


def test_converged_accumulation_statistical_parity(oracle):
    """Second half of the stated tolerance (tests/tolerances.py): after 96 accumulated frames the pixels whose paths
    diverged in single frames (last-bit branch flips) have averaged out — the accumulated images agree to > 50 dB PSNR
    and 99 % of the pixels to 2 % relative, with no bias in the mean."""
    fx = fixtures.load("converged_default_96x54_d8_acc96")
    got = oracle.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=fx["frames"],
                        **fixtures.kwargs(fx))[..., :3]
    ref = fx["expected"]
    assert _psnr(ref, got) > 50.0
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 0.05)
    assert (rel.max(-1) < 0.02).mean() > 0.99
    assert abs(got.mean() - ref.mean()) < 1e-3 * ref.mean()


def test_per_pixel_convergence_against_the_reference(oracle):
    """4,096 frames of the default scene at 64x36: every pixel of the oracle's accumulation must sit within a small fraction of the
    REFERENCE'S OWN sampling error of the reference's 4,096-frame mean (fixture: mean + standard error per pixel from the
    reference GLSL on llvmpipe, tests/golden/make_golden.py convergence).  This is the test that separates benign branch flips
    (the 0.3-1.7 % of pixels per frame that leave the 1e-4 band) from a rare-path bug: flips are sampling-equivalent and
    average out, a bug biases the pixel.  (The glass and 256-sphere scenes run in the GPU suite, where 4,096 frames cost nothing.)"""
    fx = fixtures.load("convergence_default_64x36_d8")
    img = oracle.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=fx["frames"], **fixtures.kwargs(fx))[..., :3]
    st = tol.convergence_stats(fx["mean"], fx["stderr"], img)
    print(st)
    assert st["nan_mismatch"] == 0 and st["pixels"] == fx["width"] * fx["height"]
    assert st["max_abs_z"] <= tol.CONV_MAX_ABS_Z and st["rms_z"] <= tol.CONV_RMS_Z and st["mean_rel_err"] <= tol.CONV_MEAN_REL_TOL, st


def test_edge_scene_differences_are_nan_direction_lookups_fixture(oracle):
    """The committed form of tests/test_live_reference.py's edge-scene test: per sample, the only gross differences
    between this arithmetic and the reference on the quirk scene end in texture(SamplerEnvironment, NaN) — undefined in GL."""
    fx = fixtures.load("edge_nanenv_edge_64x36_d16")
    acc = oracle.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=fx["frames"], dump_each=True,
                        **fixtures.kwargs(fx))[..., :3]
    st = fixtures.edge_nanenv_check(fx, acc)
    print(st)
    assert st["nan_env"] > 100, "the scene no longer exercises the quirk"
    assert st["gross_unflagged"] <= 3e-4 * st["samples"] and st["masked_mean_rel_err"] <= 4e-4, st
