"""Layer-2 parity as a per-pixel statement, second form (round 6): ENSEMBLE STABILITY.

tests/test_decision_margins.py bounds a pixel's sensitivity from one side with a first-order error analysis; because the bound pays every
bounce's worst-case amplification it certifies only 13 - 55 % of the pixels.  This test measures the sensitivity instead.  The oracle's
witness build (oracle/pt_oracle.c, -DPT_ORACLE_PERTURB, pto_set_ensemble) turns into ensemble MEMBERS: each member is ONE conforming
implementation of the reference's GLSL that differs from the pt-f32 contract everywhere at once, the way a real driver does —

  * every call of 1/x, inversesqrt, sqrt, sin, cos, exp, pow(x, 5) returns a result up to its allowance of ulps off (2, 2, 2, 4, 4, 4, 16:
    GLSL 4.60 section 4.7.1 for the first three, the others are left to the implementation; llvmpipe's own pow and exp are 22 and 16 ulps from
    the contract's), the shift a fixed pseudo-random function of the member and the result's bits (so the member's primitive IS a function);
  * every a * b + c fused or not, every division literal or by reciprocal, each a fixed function of the member and the operands;
  * the three products of a dot product and the four column terms of a matrix-vector product summed in an order of the member's own (GLSL
    fixes none; through the thin lens' `focalPoint - origin` one ulp of the origin is 16 - 50 ulps of the primary direction);
  * what GLSL / GL leave UNDEFINED the member chooses for itself: pow() of a negative base, comparisons and min / max on a NaN,
    texture(env, NaN direction).

A pixel is CERTIFIED when (a) no member moves its colour by more than THETA of the band around the contract's value, (b) every member
followed the contract's path through the scene — same object and side per bounce, same lobe, same ending, for every sample and every frame
accumulated so far (the "path signature" the witness build puts into the alpha channel: a pixel whose paths all die in the dark is black
under every member whatever they hit on the way, so colours alone do not see a path that forks), and (c) neither the contract nor a member
touched undefined behaviour on the way.  Statement, checked on every reference fixture (outputs of the reference's own GLSL on llvmpipe):

    EVERY certified pixel lies inside the band around the REFERENCE's value          (0 exceptions in 179,481 pixel-frames)
    i.e. every pixel the contract misses is one that conforming implementations do not agree on among themselves,

with 95.3 - 99.9 % of the pixels certified per dump (98.3 % overall) — the uncertified rest is where the frozen percentages of
thresholds.json and the witness search of test_decision_margins.py carry the claim.  A wrong pixel planted at random is caught with that
probability (test below).  The HIP path equals the contract bit for bit (tests/test_gpu_*.py), so the statement is the HIP path's too.

On FRESH data (tools/ensemble_fuzz.py, build container only: 1,200 random scenes rendered live by the reference's GLSL, 4.4 M pixels,
99.0 % certified, 34,000 outside the band): 7 certified pixels outside the band with these eight members (49 before the members
re-associated sums — that is how the re-association came in), 0 in 2.1 M pixels with sixteen.  Eight members sample the neighbourhood.

How this test found its own blind spot: the first version compared colours only and certified a black pixel of the 256-sphere fixture
whose reference value is sky-blue — five bounces of growing disagreement ended in the dark under the contract and all eight members, and
at the sky in the reference.  Hence (b).  The second exception was a camera at the centre of a glass sphere: 1 - cos(theta) = 1.2e-7 in
the Fresnel term, NaN in llvmpipe's pow() when the dot product comes out one ulp above 1.  Hence (c).
"""
import numpy as np
import pytest

import fixtures
import tolerances as tol

MEMBERS = tuple(0x1234567 * k + k for k in range(1, 9))   # eight conforming neighbours of the contract (2 / 4 / 8 / 16 members: 37 / 2 / 0 / 0
                                                          # certified pixels outside the band, 98.8 / 98.5 / 98.3 / 98.2 % certified)
AMPLITUDE = 16      # ulps, capped per primitive by its allowance (pt_oracle.c ens_allow)
THETA = 0.5         # a member may move a certified pixel by at most half the band
MIN_SHARE = 0.95    # measured 95.3 % (256 spheres) ... 99.9 %
_REPORT = []
_HULL = [0, 0]      # out-of-band pixel-frames with a finite reference value; those whose reference lies inside the implementations' range


@pytest.fixture(scope="module")
def members():
    import __graft_entry__ as graft
    o = graft.load_oracle().Oracle(perturb=True)
    yield o
    o.set_ensemble(0)
    o.set_signature_alpha(False)


def _band_distance(ref, got, band):
    """per pixel, in units of the band around `ref` (tests/tolerances.py within()); NaN == NaN agrees, NaN vs a number is infinitely far"""
    r = np.nan_to_num(ref, nan=1.0, posinf=1.0, neginf=1.0)
    scale = band * np.maximum(1.0, np.abs(r).max(-1))
    with np.errstate(invalid="ignore"):
        d = np.abs(ref.astype(np.float64) - got.astype(np.float64)).max(-1) / scale
    both = np.isnan(ref).any(-1) & np.isnan(got).any(-1)
    return np.where(both, 0.0, np.where(np.isnan(d), np.inf, d))


def _render(o, fx, sparse):
    kw = fixtures.kwargs(fx)
    if sparse:
        return o.render_pixels(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], fx["xy"], **kw)[None]
    dumps = o.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=fx["frames"], dump_each=True, **kw)
    return dumps[[int(f) for f in fx["frame_indices"]]]


def certify(o, fx, sparse):
    """-> (contract's dumps (k, ..., 4), certified mask (k, ...), colour spread in bands, members' dumps)"""
    o.set_signature_alpha(True)
    o.set_ensemble(0)
    base = _render(o, fx, sparse)
    band = tol.SRGB_REL_TOL if fx["env"].dtype == np.uint8 else tol.REL_TOL
    spread = np.zeros(base.shape[:-1])
    same_path = np.ones(base.shape[:-1], bool)
    mems = []
    for seed in MEMBERS:
        o.set_ensemble(seed, AMPLITUDE)
        m = _render(o, fx, sparse)
        mems.append(m)
        spread = np.maximum(spread, _band_distance(base[..., :3], m[..., :3], band))
        same_path &= base[..., 3].view(np.uint32) == m[..., 3].view(np.uint32)
    o.set_ensemble(0)
    o.set_signature_alpha(False)
    return base, (spread <= THETA) & same_path, spread, mems


@pytest.mark.parametrize("name", fixtures.names("frame_") + fixtures.names("sparse_"))
def test_every_certified_pixel_is_inside_the_band_of_the_reference(oracle, members, name):
    fx = fixtures.load(name)
    sparse = name.startswith("sparse_")
    base, certified, spread, mems = certify(members, fx, sparse)
    band = tol.SRGB_REL_TOL if fx["env"].dtype == np.uint8 else tol.REL_TOL
    expected = fx["expected"][None] if sparse else fx["expected"]
    # the witness build with no member selected IS the contract (colours; alpha carries the signature here)
    kw = fixtures.kwargs(fx)
    plain = (oracle.render_pixels(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], fx["xy"], **kw) if sparse else
             oracle.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=fx["frames"], **kw))
    assert np.array_equal(base[-1][..., :3].view(np.uint32), plain[..., :3].view(np.uint32)), "ensemble off must render the plain oracle's bits"
    for k in range(base.shape[0]):
        d_ref = _band_distance(expected[k], base[k][..., :3], band)
        outside = d_ref > 1.0
        share = float(certified[k].mean())
        # how the members themselves fare against the reference: they are neighbours, not strangers
        member_in = [float((_band_distance(expected[k], m[k][..., :3], band) <= 1.0).mean()) for m in mems]
        _REPORT.append((f"{name} #{k}", int(outside.sum()), int(outside.size), share, float(d_ref[certified[k]].max()), 1.0 - float(outside.mean()),
                        min(member_in), max(member_in)))
        # where the contract misses: is the reference's value inside the range the nine implementations span (+- one band, per channel)?
        stack = np.stack([base[k][..., :3]] + [m[k][..., :3] for m in mems])
        scale = band * np.maximum(1.0, np.abs(np.nan_to_num(expected[k], nan=1.0, posinf=1.0, neginf=1.0)).max(-1))[..., None]
        with np.errstate(invalid="ignore"):
            in_hull = ((expected[k] >= np.nanmin(stack, 0) - scale) & (expected[k] <= np.nanmax(stack, 0) + scale)).all(-1)
        finite = outside & np.isfinite(expected[k]).all(-1)
        _HULL[0] += int(finite.sum())
        _HULL[1] += int((finite & in_hull).sum())
        bad = np.argwhere(outside & certified[k])
        assert bad.size == 0, (f"{name} #{k}: {len(bad)} pixel(s) that no conforming neighbour moves differ from the reference by more than the band, "
                               f"first at index {bad[0].tolist()} ({d_ref[tuple(bad[0])]:.1f} bands): a discrepancy last-bit arithmetic does not explain")
        assert share >= MIN_SHARE, f"{name} #{k}: only {share:.1%} of the pixels certified"
        # a member agrees with the reference about as often as the contract does (it is as conforming as the contract)
        assert min(member_in) >= (1.0 - float(outside.mean())) - 0.02


def test_a_planted_error_is_caught_where_the_pixel_is_certified(members):
    """1,000 wrong pixels (3 - 100 bands off, random places) in the contract's image of two fixtures: flagged iff the pixel is certified."""
    rng = np.random.default_rng(7)
    caught = total = 0
    for name in ("frame_default_128x72_d8", "frame_stress256_128x72_d8"):
        fx = fixtures.load(name)
        base, certified, _, _ = certify(members, fx, False)
        img, cert = base[0][..., :3].copy(), certified[0]
        ys, xs = rng.integers(0, img.shape[0], 500), rng.integers(0, img.shape[1], 500)
        for y, x in zip(ys, xs):
            wrong = img[y, x] + rng.choice([-1.0, 1.0]) * rng.uniform(3, 100) * tol.REL_TOL * max(1.0, float(np.abs(fx["expected"][0][y, x]).max()))
            d = _band_distance(fx["expected"][0][y, x][None], wrong[None].astype(np.float32), tol.REL_TOL)[0]
            total += 1
            caught += bool(cert[y, x] and d > 1.0)
    _REPORT.append(("planted errors caught", caught, total, caught / total, 0.0, 0.0, 0.0, 0.0))
    assert caught / total >= 0.93   # = the certified share of these two fixtures (98.8 %, 95.3 %) minus pixels the error happens to leave inside


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["frame_default_128x72_d8", "frame_stress256_128x72_d8", "frame_edge_128x72_d16"])
def test_gpu_every_certified_pixel_is_inside_the_band_of_the_reference(pkg, native_lib, members, name):
    """The same statement on the GPU's own images (the HIP path renders the contract's bits: shown here, not assumed)."""
    fx = fixtures.load(name)
    base, certified, _, _ = certify(members, fx, False)
    hip = fixtures.hip_frames(pkg, fx)
    band = tol.SRGB_REL_TOL if fx["env"].dtype == np.uint8 else tol.REL_TOL
    for k in range(base.shape[0]):
        same = (hip[k].view(np.uint32) == base[k][..., :3].view(np.uint32)) | (np.isnan(hip[k]) & np.isnan(base[k][..., :3]))  # (NaN payloads may differ)
        assert same.all()
        d_ref = _band_distance(fx["expected"][k], hip[k], band)
        assert not ((d_ref > 1.0) & certified[k]).any()
        assert certified[k].mean() >= MIN_SHARE


def test_where_the_contract_misses_the_reference_is_one_of_the_neighbours():
    """(runs after the fixture tests)  In the pixels OUTSIDE the band the nine implementations (contract + eight members) disagree among
    themselves; if the reference's GLSL on llvmpipe is one more conforming implementation its value is exchangeable with theirs: it lies
    inside the range they span, per channel, about 8 times in 10 (a tenth draw is the smallest or the largest of ten with probability
    2 / 10), more often where outcomes are discrete.  Measured: 703 of 772 = 91 % (expected 80 % for a tenth draw, more where outcomes are discrete) — the reference is
    neither systematically brighter nor darker than its neighbours where they scatter (rank histogram of its luminance among 17 in docs/parity.md)."""
    if _HULL[0] == 0:
        pytest.skip("the fixture tests did not run in this session")
    assert _HULL[0] > 500
    assert _HULL[1] / _HULL[0] >= 0.80, f"only {_HULL[1]} of {_HULL[0]} out-of-band reference values lie inside the implementations' range"


def test_report():
    """(prints the table; run with -s)"""
    print("\nensemble stability: dump | outside the band | certified share | largest distance of a certified pixel (bands) | in-band share: contract, members min .. max")
    for name, n_out, n, share, worst, c_in, m_lo, m_hi in _REPORT:
        print(f"  {name:44s} {n_out:5d} / {n:6d}   {share:7.2%}   {worst:6.3f}   {c_in:7.2%}  {m_lo:7.2%} .. {m_hi:7.2%}")
    if _HULL[0]:
        print(f"  out-of-band pixel-frames whose reference value lies inside the range the nine implementations span: {_HULL[1]} of {_HULL[0]} = {_HULL[1] / _HULL[0]:.1%}")
