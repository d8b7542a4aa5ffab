"""Layer-2 parity as a PER-PIXEL statement (round 5).

Layer 2 (oracle ~ reference) used to be a percentage: "x % of the pixels lie inside 1e-4 * max(1, |ref|), the rest are assumed to be
branch flips" (tests/tolerances.py, thresholds.json).  The oracle's -DPT_ORACLE_MARGINS build (oracle/pt_oracle.c, "decision margins")
makes the assumption checkable: next to every path it carries a first-order bound of the path's own error per unit of relative error eps
of the arithmetic's primitives, and records per pixel
    margin = the smallest eps at which one of the pixel's data-dependent comparisons (compute.glsl:169,201,208,234,247,269,293,347-350,
             322-332, refract's k < 0) would come out the other way,
    cont   = the colour error per unit eps the pixel suffers without any flip.
M = min(margin, band / cont) is the relative error of the primitives (1 ulp = 6e-8) a conforming implementation needs to move the pixel
out of the band.  The reference's GLSL on llvmpipe and the pt-f32 contract differ by fractions of an ulp to a few ulp per primitive
(GLSL leaves /, sqrt, inversesqrt, sin, cos, exp implementation-defined), so:

    EVERY pixel of EVERY reference fixture that lies outside the band has M < TAU          (it sits on a knife edge), equivalently
    EVERY pixel with M >= TAU lies inside the band                                           (100 %, not 98-99.9 %).

TAU = 2e-8 (a third of an ulp; the largest M of an out-of-band pixel is 1.82e-8).  The frozen per-fixture percentages of thresholds.json stay as a report.  The margin build renders the same
bits as the plain oracle (checked here), and the HIP path equals the oracle bit for bit (tests/test_gpu_*.py), so the statement is the
HIP path's too; `-m gpu`: test_gpu_decision_margins below makes it on the GPU's own images.
"""
import numpy as np
import pytest

import fixtures
import tolerances as tol

TAU = 2e-8           # relative error of the arithmetic's primitives (1 ulp of binary32 = 6e-8)
_REPORT = []


@pytest.fixture(scope="module")
def oracle_margins():
    import __graft_entry__ as graft
    return graft.load_oracle().Oracle(margins=True)


def sensitivity(margin, cont, ref, band):
    """M per pixel: the eps that flips a comparison, or that moves the colour out of the band without a flip, whichever is smaller."""
    band_abs = band * np.maximum(1.0, np.abs(np.nan_to_num(ref, nan=1.0, posinf=1.0, neginf=1.0)).max(-1))
    return np.minimum(margin, band_abs / np.maximum(cont, 1e-30))


def check(name, ref, got, margin, cont, srgb):
    band = tol.SRGB_REL_TOL if srgb else tol.REL_TOL
    both_nan = np.isnan(ref).any(-1) & np.isnan(got).any(-1)
    inside = tol.within(ref, got, band) | both_nan
    M = sensitivity(margin, cont, ref, band)
    out = ~inside
    worst = float(M[out].max()) if out.any() else 0.0
    _REPORT.append((name, int(out.sum()), int(out.size), worst, float((M >= TAU).mean())))
    assert worst < TAU, f"{name}: a pixel outside the band needs eps = {worst:.2e} to flip (TAU = {TAU:.0e}): not a knife-edge decision"
    assert inside[M >= TAU].all()
    return M


@pytest.mark.parametrize("name", fixtures.names("frame_"))
def test_every_out_of_band_pixel_of_a_frame_fixture_sits_on_a_knife_edge(oracle, oracle_margins, name):
    fx = fixtures.load(name)
    imgs, margins, conts = oracle_margins.render_with_margins(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"],
                                                              num_frames=fx["frames"], dump_each=True, **fixtures.kwargs(fx))
    plain = oracle.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=fx["frames"], **fixtures.kwargs(fx))
    assert np.array_equal(imgs[-1].view(np.uint32), plain.view(np.uint32)), "the margin build must render the plain oracle's bits"
    for k, fi in enumerate(fx["frame_indices"]):  # accumulated image after frame fi: a flip in ANY frame so far moves the mean
        check(f"{name} #{k}", fx["expected"][k], imgs[fi][..., :3], margins[fi], conts[fi], fx["env"].dtype == np.uint8)


@pytest.mark.parametrize("name", fixtures.names("sparse_"))
def test_every_out_of_band_pixel_of_the_full_size_configs_sits_on_a_knife_edge(oracle_margins, name):
    """4,096 seeded pixels of the full-size BASELINE configs C1-C5 (1080p, 4K, 256 spheres, 32-bounce glass)."""
    fx = fixtures.load(name)
    got, margin, cont = oracle_margins.render_pixels_margins(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], fx["xy"],
                                                             **fixtures.kwargs(fx))
    check(name, fx["expected"], got[:, :3], margin, cont, fx["env"].dtype == np.uint8)


# ---- constructive witnesses (round 6).  "Sits on a knife edge" is an upper bound from a first-order analysis; the converse is SHOWN per
# pixel: a pixel of the reference that the contract misses is HIT — inside the band — by a neighbouring conforming implementation of the
# same GLSL, found by oracle/pt_oracle.c pto_witness_search (witness build, -DPT_ORACLE_PERTURB) and replayed here through
# pto_render_pixel_variant:
#   1  ONE data-dependent comparison of the pixel's path inverted (operands closer than 1e-6 of their scale — ~17 ulps; until the ensemble test showed two
#      spurious ones the search allowed 1e-2, and tightening it 10,000-fold lost no witness: 378 / 36 / 43 / 3 of 460 either way), everything else the contract;
#   2  ONE call of ONE primitive (1/x, inversesqrt, sqrt, sin, cos, exp, pow) 1-2 ulps off (sin / cos / exp up to 4, pow up to 16: GLSL
#      leaves them to the implementation), ONE a * b + c evaluated with two roundings instead of fused (llvmpipe never fuses), ONE a / b
#      as a true division where the contract multiplies by a reciprocal, or ONE mix(x, y, a) as x + a (y - x);
#   3  two comparisons inverted (the second on the changed path);
#   4, 5  several such calls at once (paths that amplify: a few bounces on curved surfaces turn one ulp into 1e-3 of the colour, and
#      the reference's pixel is one of the values the neighbours scatter over);
#   9  pow(x, 5) of a base that is negative or within four ulps of zero returns NaN (undefined in GLSL; llvmpipe's exp2(5 log2 x) does);
# or its path — the contract's (8), or the contract's with one comparison inverted / that pow (7) — ends in texture(env, NaN direction):
# undefined in GL, llvmpipe returns one texel average, the contract another (docs/parity.md; total internal reflection -> refract() = 0 ->
# normalize(0)).  Accumulated frames are taken one frame at a time from the REFERENCE's own previous accumulation (dumps of consecutive
# frames), so every dump is a single-frame statement.  Measured over all fixtures (553 pixel-frames outside the band, all searched): 93 %
# hit by a neighbour (78 % around the contract, the rest around llvmpipe's own choices, BASES), 6.5 % end in the undefined lookup (and imply the same value of it as at least two other pixel-frames of the
# environment), none is merely moved out of the band by a single call one ulp off (demonstrably unstable) without a neighbour landing inside
# — their neighbours scatter over tens to thousands of bands and the search enumerates six sites at a time —, 0.4 % neither (until the search also ran around llvmpipe's SUMMATION ORDERS: 82 / 7 / 9 / 1.3 %).  Gated: the share reached, and that no pixel is without
# any of the three.  The global variants of the earlier rounds (one primitive off EVERYWHERE: 40-60 %) are subsumed.
BASES = (0, 951, 7)    # the implementations the search runs around: the contract, then llvmpipe's own choices — 951: never fused, correctly rounded
                       # 1/x, sqrt, 1/sqrt, literal divisions, its summation orders, its own sin / cos / exp / pow, mix and the cube filter as lerps
                       # (tests/test_arithmetic_choices.py: with these the restatement renders 94 % of the reference's pixels bit for bit); 7: the
                       # first three only — as conforming as the contract, and much nearer to the reference where paths amplify.  Share reached:
                       # 90.0 % around (0, 7), 98.9 % with the summation orders, 99.6 % with 951
MAX_SEARCHED = 160     # (all of them: the 256-sphere fixtures have up to 156 outside the band, ~0.3 s each there)
_WITNESS_REPORT = []


@pytest.fixture(scope="module")
def witness_oracle():
    import __graft_entry__ as graft
    return graft.load_oracle().Oracle(perturb=True)


def _single_frame_cases(fx, sparse):
    """(frame index, xy, reference after this frame, reference's accumulation before it or None) for every dump that can be taken alone"""
    if sparse:
        return [("", 0, fx["xy"], fx["expected"], None)]
    H, W = fx["height"], fx["width"]
    yy, xx = np.mgrid[0:H, 0:W]
    xy = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.int32)
    cases, fis = [], [int(f) for f in fx["frame_indices"]]
    for k, fi in enumerate(fis):
        ref = fx["expected"][k].reshape(-1, 3)
        if fi == 0:
            cases.append((f" #{k}", 0, xy, ref, None))
        elif k > 0 and fis[k - 1] == fi - 1:
            prev = fx["expected"][k - 1].reshape(-1, 3)
            cases.append((f" #{k}", fi, xy, ref, np.concatenate([prev, np.ones((prev.shape[0], 1), np.float32)], 1)))
    return cases


def _witnesses(name, fx, oracle, wit, sparse):
    srgb = fx["env"].dtype == np.uint8
    band = tol.SRGB_REL_TOL if srgb else tol.REL_TOL
    W, H, kw = fx["width"], fx["height"], fixtures.kwargs(fx)
    scene = (W, H, fx["basic"], fx["objects"], fx["env"])
    for tag, fi, xy, ref, last in _single_frame_cases(fx, sparse):
        got = oracle.render_pixels(*scene, xy, frame=fi, last=last, **kw)[:, :3]
        out = ~(tol.within(ref, got, band) | (np.isnan(ref).any(-1) & np.isnan(got).any(-1)))
        idx = np.nonzero(out)[0][:MAX_SEARCHED]
        hit = unstable = nothing = 0
        lookups = []  # pixels whose (possibly varied) path ends in texture(env, NaN): the value of that lookup the reference's pixel implies
        first = {}    # what the search around the CONTRACT said about the pixels no base reaches
        todo = list(idx)
        for base in BASES:
            if not todo:
                break
            wit.set_base_variant(base)
            sel = np.array(todo)
            res = wit.witness_search(*scene, xy[sel], ref[sel], band, frame=fi, last=None if last is None else last[sel], **kw)
            todo = []
            for i, r in zip(sel, res):
                if r["kind"] not in (1, 2, 3, 4, 5, 9):
                    first.setdefault(int(i), (base, r))
                    todo.append(i)
                    continue
                # replay the neighbour and check it with the suite's own band test
                v, _ = wit.render_pixel_variant(*scene, xy[i, 0], xy[i, 1], frame=fi, last=None if last is None else last[i],
                                                flips=r["flips"][:2] if r["kind"] != 9 else (), sites=r["sites"],
                                                pow_neg_nan=r["flips"][2] if r["kind"] == 9 else 0, **kw)
                assert np.array_equal(v.view(np.uint32), r["value"].view(np.uint32)), f"{name}{tag}: the replay of pixel {xy[i]} differs from the search"
                assert tol.within(ref[i], v[:3], band) or (np.isnan(ref[i]).any() and np.isnan(v[:3]).any()), f"{name}{tag}: witness of {xy[i]} is outside"
                hit += 1
        for i in todo:
            base, r = first[int(i)]
            wit.set_base_variant(base)
            last_i = None if last is None else last[i]
            implied = None
            if r["kind"] in (7, 8):  # the pixel is linear in the undefined lookup's value E: pixel = A + T E
                var = dict(flips=(r["flips"][0],) if r["kind"] == 7 and r["flips"][0] >= 0 else (),
                           pow_neg_nan=r["flips"][2] if r["kind"] == 7 and r["flips"][0] == -2 else 0)
                ab = []
                for e in (0.0, 1.0):
                    wit.set_nan_env([e, e, e])
                    ab.append(wit.render_pixel_variant(*scene, xy[i, 0], xy[i, 1], frame=fi, last=last_i, **var, **kw)[0][:3].astype(np.float64))
                wit.set_nan_env(None)
                T = ab[1] - ab[0]
                if np.isfinite(T).all() and (T > 1e-6).all():
                    implied = (ref[i].astype(np.float64) - ab[0]) / T
            if implied is not None and np.isfinite(implied).all() and (implied > 0).all():
                lookups.append((implied, r["unstable_calls"] > 0))
            elif r["unstable_calls"] > 0:
                unstable += 1
            else:
                nothing += 1
        wit.set_base_variant(0)
        # the unperturbed replay is the contract (the hooks change nothing when idle)
        if len(idx):
            v, _ = wit.render_pixel_variant(*scene, xy[idx[0], 0], xy[idx[0], 1], frame=fi, last=None if last is None else last[idx[0]], **kw)
            assert np.array_equal(v[:3].view(np.uint32), got[idx[0]].view(np.uint32))
        env_key = (fx["env"].shape, hash(fx["env"].tobytes()))
        _WITNESS_REPORT.append([f"{name}{tag}", int(out.sum()), len(idx), hit, 0, unstable, nothing, env_key, lookups])


def _settle_undefined_lookups():
    """texture(env, NaN) is ONE value per environment and sign of the NaN in llvmpipe: a pixel counts as explained by the undefined lookup
    only if the value it implies is implied by at least two other pixel-frames of the same environment as well (within 0.5 %); the others
    fall back to `unstable` / `nothing`."""
    by_env = {}
    for r in _WITNESS_REPORT:
        for implied, _ in r[8]:
            by_env.setdefault(r[7], []).append(implied)
    for r in _WITNESS_REPORT:
        pool = by_env.get(r[7], [])
        for implied, is_unstable in r[8]:
            agree = sum(1 for other in pool if np.all(np.abs(other - implied) <= 5e-3 * np.abs(implied)))
            if agree >= 3:  # itself + two others
                r[4] += 1
            elif is_unstable:
                r[5] += 1
            else:
                r[6] += 1
        r[8] = []


@pytest.mark.parametrize("name", fixtures.names("frame_"))
def test_out_of_band_pixels_of_a_frame_fixture_have_conforming_witnesses(oracle, witness_oracle, name):
    _witnesses(name, fixtures.load(name), oracle, witness_oracle, False)


@pytest.mark.parametrize("name", fixtures.names("sparse_"))
def test_out_of_band_pixels_of_the_full_size_configs_have_conforming_witnesses(oracle, witness_oracle, name):
    _witnesses(name, fixtures.load(name), oracle, witness_oracle, True)


def test_planted_errors_find_no_witness(oracle_margins, witness_oracle):
    """The discriminating power of the pair (margin, witness).  The margin test alone lets a wrong pixel through whenever the pixel's
    M < TAU (45 - 85 % of the pixels).  Plant errors of 10 bands into 120 seeded in-band pixels of a fixture: the ones with M >= TAU fail
    the margin test; for the others the witness search must come back empty — a neighbour of the contract reproduces what the REFERENCE
    computed, not an arbitrary value.  (Measured on four fixtures, errors of 3 / 10 / 100 bands, 1,500 planted pixels that pass the
    margin test: 0 explained.)"""
    name = "frame_default_128x72_d8"
    fx = fixtures.load(name)
    W, H, kw = fx["width"], fx["height"], fixtures.kwargs(fx)
    scene = (W, H, fx["basic"], fx["objects"], fx["env"])
    yy, xx = np.mgrid[0:H, 0:W]
    xy = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.int32)
    ref = fx["expected"][0].reshape(-1, 3)
    got, margin, cont = oracle_margins.render_pixels_margins(*scene, xy, **kw)
    inside = tol.within(ref, got[:, :3], tol.REL_TOL)
    M = sensitivity(margin, cont, ref, tol.REL_TOL)
    rng = np.random.default_rng(7)
    sel = rng.choice(np.nonzero(inside & np.isfinite(ref).all(-1))[0], 120, replace=False)
    planted = ref[sel] + np.sign(rng.standard_normal((120, 3))) * 10.0 * tol.REL_TOL * np.maximum(1.0, np.abs(ref[sel]))
    passes_margin = M[sel] < TAU
    assert 30 <= passes_margin.sum() <= 110  # (the margin test alone is not enough: that is the point)
    res = witness_oracle.witness_search(*scene, xy[sel[passes_margin]], planted[passes_margin], tol.REL_TOL, **kw)
    explained = sum(1 for r in res if r["kind"] in (1, 2, 3, 4, 5, 9))
    assert explained <= 1, f"{explained} of {passes_margin.sum()} planted errors have a 'witness'"


def test_witness_share():
    """(runs after the witness tests) over all fixtures: the share of the searched out-of-band pixels that a neighbouring conforming
    implementation reproduces inside the band or whose path ends in GL's undefined lookup; and how many have nothing to show."""
    if not _WITNESS_REPORT:
        pytest.skip("the witness tests did not run in this session")
    _settle_undefined_lookups()
    searched = sum(r[2] for r in _WITNESS_REPORT)
    reached = sum(r[3] + r[4] for r in _WITNESS_REPORT)
    nothing = sum(r[6] for r in _WITNESS_REPORT)
    assert searched > 500
    assert reached / searched >= 0.95, f"only {reached} of {searched} out-of-band pixels are reached by a conforming neighbour (measured: 99.6 %)"
    assert nothing / searched <= 0.02, f"{nothing} of {searched} out-of-band pixels have neither a witness nor a demonstrated instability (measured: 0.4 %)"


def test_report(capsys):
    """(not a check: prints what the tests above measured; runs last in this module)"""
    with capsys.disabled():
        print("\n  decision margins: fixture, pixels outside the band, largest eps among them, share of pixels with M >= TAU (proven inside)")
        for name, nout, n, worst, safe in _REPORT:
            print(f"    {name:44s} {nout:5d} / {n:7d}   {worst:9.2e}   {100 * safe:6.2f} %")
        print("  witnesses: fixture, pixels outside the band, searched | hit by a conforming neighbour, end in GL's undefined lookup, "
              "unstable (one call one ulp off moves them out of the band) but not reached, nothing")
        for name, nout, searched, hit, undefined, unstable, nothing, _, _ in _WITNESS_REPORT:
            print(f"    {name:44s} {nout:5d} {searched:5d} | {hit:5d} {undefined:5d} {unstable:5d} {nothing:5d}")
        tot = [sum(r[k] for r in _WITNESS_REPORT) for k in range(1, 7)]
        if tot[1]:
            print(f"    {'all':44s} {tot[0]:5d} {tot[1]:5d} | {tot[2]:5d} {tot[3]:5d} {tot[4]:5d} {tot[5]:5d}   "
                  f"(reached {100 * (tot[2] + tot[3]) / tot[1]:.1f} %, nothing {100 * tot[5] / tot[1]:.1f} %)")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["frame_default_128x72_d8", "frame_stress256_128x72_d8", "frame_glass_128x72_d32_atmo"])
def test_gpu_decision_margins(pkg, native_lib, oracle_margins, name):
    """The same statement on the GPU's own images (the HIP path renders the oracle's bits, so this is the same pixels: shown, not assumed)."""
    fx = fixtures.load(name)
    imgs, margins, conts = oracle_margins.render_with_margins(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"],
                                                              num_frames=fx["frames"], dump_each=True, **fixtures.kwargs(fx))
    hip = fixtures.hip_frames(pkg, fx)
    for k, fi in enumerate(fx["frame_indices"]):
        assert np.array_equal(hip[k].view(np.uint32), imgs[fi][..., :3].view(np.uint32))
        check(f"HIP {name} #{k}", fx["expected"][k], hip[k], margins[fi], conts[fi], fx["env"].dtype == np.uint8)
