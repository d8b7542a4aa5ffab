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

TAU = 1e-7 (under 2 ulp).  The frozen per-fixture percentages of thresholds.json stay as a report.  The margin build renders the same
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


# ---- constructive witnesses (round 6).  "Sits on a knife edge" is an upper bound from a first-order analysis; the converse can be SHOWN
# for part of the pixels: a pixel of the reference that the contract misses and that IS HIT by a neighbouring conforming implementation —
# the contract with one of its primitives one or two ulps off EVERYWHERE (oracle build -DPT_ORACLE_PERTURB), with the literal slab
# division, or with correctly rounded 1/x, sqrt, 1/sqrt; GLSL allows every one of them, llvmpipe is yet another — is a demonstrated flip.
# Measured: this family of 26 GLOBAL variants hits 40 - 60 % of the out-of-band pixels (a wider one — +-4 ulps, +-16 on exp and pow, which
# is what llvmpipe's exp is off by — adds almost nothing).  The others need their ONE knife-edge comparison decided the other way while
# everything else stays as it is, which no global shift of a primitive does: the targeted single-decision flip (re-render the pixel with
# comparison #k inverted, k from the margins build) is the witness that can reach 100 %; it is not built.  What is gated here is the
# family's overall share, so that the statistic cannot silently rot.
WITNESS_VARIANTS = [("truediv", None), ("exact", None)] + [(f"{name}{ulps:+d}", (prim, ulps)) for prim, name in
                                                           enumerate(["rcp", "rsqrt", "sqrt", "sin", "cos", "exp"]) for ulps in (1, -1, 2, -2)]
_WITNESS_REPORT = []


@pytest.fixture(scope="module")
def witness_oracles():
    import __graft_entry__ as graft
    po = graft.load_oracle()
    return {"truediv": po.Oracle(true_division=True), "exact": po.Oracle(exact=True), "perturb": po.Oracle(perturb=True)}


def _variant_frames(orcs, variant, fx, sparse):
    name, pert = variant
    o = orcs["perturb"] if pert else orcs[name]
    if pert:
        o.set_perturbation(*pert)
    try:
        if sparse:
            return o.render_pixels(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], fx["xy"], **fixtures.kwargs(fx))[:, :3]
        imgs = o.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=fx["frames"], dump_each=True, **fixtures.kwargs(fx))
        return [imgs[fi][..., :3] for fi in fx["frame_indices"]]
    finally:
        if pert:
            o.set_perturbation(-1, 0)


def _witness(name, refs, gots, variants, srgb):
    band = tol.SRGB_REL_TOL if srgb else tol.REL_TOL
    for k, (ref, got) in enumerate(zip(refs, gots)):
        both_nan = np.isnan(ref).any(-1) & np.isnan(got).any(-1)
        out = ~(tol.within(ref, got, band) | both_nan)
        hit = np.zeros_like(out)
        for v in variants:
            vk = v[k]
            hit |= tol.within(ref, vk, band) | (np.isnan(ref).any(-1) & np.isnan(vk).any(-1))
        missing = int((out & ~hit).sum())
        _WITNESS_REPORT.append((f"{name} #{k}", int(out.sum()), missing))
    # (no per-fixture gate: see test_witness_share below)


@pytest.mark.parametrize("name", fixtures.names("frame_"))
def test_every_out_of_band_pixel_of_a_frame_fixture_has_a_conforming_witness(oracle, witness_oracles, name):
    fx = fixtures.load(name)
    imgs = oracle.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=fx["frames"], dump_each=True, **fixtures.kwargs(fx))
    gots = [imgs[fi][..., :3] for fi in fx["frame_indices"]]
    variants = [_variant_frames(witness_oracles, v, fx, False) for v in WITNESS_VARIANTS]
    _witness(name, fx["expected"], gots, variants, fx["env"].dtype == np.uint8)


@pytest.mark.parametrize("name", fixtures.names("sparse_"))
def test_every_out_of_band_pixel_of_the_full_size_configs_has_a_conforming_witness(oracle, witness_oracles, name):
    fx = fixtures.load(name)
    got = oracle.render_pixels(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], fx["xy"], **fixtures.kwargs(fx))[:, :3]
    variants = [[_variant_frames(witness_oracles, v, fx, True)] for v in WITNESS_VARIANTS]
    _witness(name, [fx["expected"]], [got], variants, fx["env"].dtype == np.uint8)


def test_witness_share():
    """(runs after the witness tests) of all out-of-band pixels of all fixtures' first frames, the share the 26 global variants hit."""
    firsts = [r for r in _WITNESS_REPORT if r[0].endswith("#0")]
    if not firsts:
        pytest.skip("the witness tests did not run in this session")
    nout, missing = sum(r[1] for r in firsts), sum(r[2] for r in firsts)
    assert nout > 0 and 1.0 - missing / nout >= 0.40, f"only {nout - missing} of {nout} out-of-band pixels have a witness in the global family"


def test_report(capsys):
    """(not a check: prints what the tests above measured; runs last in this module)"""
    with capsys.disabled():
        print("\n  decision margins: fixture, pixels outside the band, largest eps among them, share of pixels with M >= TAU (proven inside)")
        for name, nout, n, worst, safe in _REPORT:
            print(f"    {name:44s} {nout:5d} / {n:7d}   {worst:9.2e}   {100 * safe:6.2f} %")
        print("  witnesses: fixture, pixels outside the band, of which NO neighbouring conforming implementation lands inside")
        for name, nout, missing in _WITNESS_REPORT:
            print(f"    {name:44s} {nout:5d}   {missing:5d}")


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
