"""Layer-2 parity, explained by NAME (round 6): which arithmetic choices separate the pt-f32 contract from the reference's run on llvmpipe.

GLSL leaves open whether a * b + c is fused, how accurate 1/x, sqrt, inversesqrt, sin, cos, exp, pow are, whether a / b is a division or a
product with a reciprocal, which algebraic form mix() takes, and in which order a dot product's three and a matrix-vector product's four
terms are summed.  The contract (oracle/pt_oracle.c, csrc/pt_math.hpp) chose for the GPU: fused chains, Newton sequences (<= 0.5 / 1.7
ulp), reciprocals, x-y-z(-w) order, x (1 - a) + y a.  What llvmpipe chose was MEASURED here, in the build container, by running GLSL on
the live llvmpipe through oracle/_ref/glsl_runner (own test mains; intermediate values of an instrumented scratch copy of the reference's
shader): a * b + c in shader code never fused; a / b, 1 / x, sqrt correctly rounded; inversesqrt(x) = 1 / sqrt(x); mix = x + a (y - x); dot
products x x + (y y + z z); mat4 * vec4 = ((w + x) + z) + y by columns (a frame's primary rays then match bit for bit: 100.00 % of origins and
directions); the cube filter two nested fused lerps; and its sin, cos, exp, pow as Mesa's gallivm evaluates them — restated from the
published algorithm and BIT-IDENTICAL with the live llvmpipe on 65,536 arguments each (last test).

The oracle's witness build can BE that implementation (pto_set_base_variant, bits below).  Then the SAME C restatement

    misses 49 instead of 787 of the fixtures' 179,481 pixel-frames                          (outside the band 1e-4 max(1, |ref|))
    equals the reference BIT FOR BIT in 98.6 % of the first frames' pixels                   (contract 38.8 %; every dump incl. accumulated: 98.2 %)
    cumulatively, cheapest first:  contract 787 -> summation orders + mix form + filter lerps (free on the GPU) 740 -> literal divisions 732
                                   -> exact 1/x sqrt 1/sqrt 593 -> NEVER FUSED 61 -> llvmpipe's sin cos exp pow 49

which is the strongest available form of "the oracle restates the reference's algorithm": no pixel class is systematically off, and what
separates the two is, above all, the fused multiply-add.  (On 600 random scenes of tools/ensemble_fuzz.py: 11 scenes with undefined
behaviour in view hold 8,904 of the 12,995 out-of-band pixels and do not move; the other 589 go 4,091 -> 2,326.)  The contract keeps its
choices because the integrator is VALU-issue-bound: unfused multiply-adds and IEEE division / square root (43 / 52 issue
cycles on gfx950) would cost about a quarter of the speed for 0.4 points of agreement with ONE other conforming implementation.  These
tests pin the measurements."""
import ctypes as C
import importlib.util
import os
import tempfile

import numpy as np
import pytest

import fixtures
import tolerances as tol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("glsl_run", os.path.join(ROOT, "oracle", "glsl_ref", "run.py"))
ref = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref)

NEVER_FUSED, EXACT_DIV_SQRT, LITERAL_DIVISION, MATVEC_W_FIRST, MATVEC_LLVMPIPE, DOT_X_PLUS_YZ, LLVM_MATH, MIX_AS_LERP, SAMPLER_LERPS = 1, 2, 4, 1 << 3, 2 << 3, 1 << 5, 1 << 7, 1 << 8, 1 << 9
LLVMPIPE = NEVER_FUSED | EXACT_DIV_SQRT | LITERAL_DIVISION | MATVEC_LLVMPIPE | DOT_X_PLUS_YZ | LLVM_MATH | MIX_AS_LERP | SAMPLER_LERPS   # (LLVM_MATH: its sin, cos, exp, pow — exact, see the probe below)


@pytest.fixture(scope="module")
def variants():
    import __graft_entry__ as graft
    o = graft.load_oracle().Oracle(perturb=True)
    yield o
    o.set_base_variant(0)


def outside(o, bits):
    o.set_base_variant(bits)
    n_out = n = 0
    for name in fixtures.names("frame_") + fixtures.names("sparse_"):
        fx = fixtures.load(name)
        kw = fixtures.kwargs(fx)
        band = tol.SRGB_REL_TOL if fx["env"].dtype == np.uint8 else tol.REL_TOL
        if name.startswith("sparse_"):
            got = o.render_pixels(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], fx["xy"], **kw)[None][..., :3]
            exp = fx["expected"][None]
        else:
            dumps = o.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=fx["frames"], dump_each=True, **kw)
            got, exp = dumps[[int(f) for f in fx["frame_indices"]]][..., :3], fx["expected"]
        for k in range(got.shape[0]):
            ok = tol.within(exp[k], got[k], band) | (np.isnan(exp[k]).any(-1) & np.isnan(got[k]).any(-1))
            n_out += int((~ok).sum())
            n += ok.size
    o.set_base_variant(0)
    return n_out, n


def test_llvmpipes_arithmetic_choices_close_two_thirds_of_the_gap(variants):
    contract, n = outside(variants, 0)
    orders_only, _ = outside(variants, MATVEC_LLVMPIPE | DOT_X_PLUS_YZ)
    llvmpipe, _ = outside(variants, LLVMPIPE)
    print(f"\n  outside the band of {n} pixel-frames: contract {contract}, llvmpipe's summation orders only {orders_only}, all of llvmpipe's choices {llvmpipe}")
    assert n > 150000 and 600 <= contract <= 900          # (measured 787)
    assert orders_only <= contract                         # (743: the order alone helps a little under fused arithmetic)
    assert llvmpipe <= 0.1 * contract                      # (49 = 0.06 x)


@pytest.mark.parametrize("name", fixtures.names("atmo_"))
def test_the_atmosphere_cubes_band_is_the_same_named_choices(variants, name):
    """The atmosphere precompute (AtmosphericScattering/compute.glsl) has no branching on random numbers: its distance from
    the reference's cube is arithmetic only.  Contract: largest per-texel error 1.5e-4 ... 3.5e-4, median 4e-6, 97.8 - 99.9 % of the texels
    within 1e-4 (the frozen marks of tests/tolerances.py).  With llvmpipe's choices — never fused + correctly rounded 1/x, sqrt — the SAME
    restatement is within 9e-5 EVERYWHERE, median 1.4e-7: the band the atmosphere needs is those two choices (what remains is llvmpipe's
    exp, ~16 ulps, over 750 accumulation steps).  Either choice alone does not do it (never fused: 1.5e-4 ... 2.2e-4; exact roots: unchanged)."""
    fx = fixtures.load(name)
    size, isteps, jsteps = (int(v) for v in fx["params"])
    err = {}
    for bits in (0, LLVMPIPE):
        variants.set_base_variant(bits)
        got = variants.atmosphere(size, fx["ubo"].tobytes(), fx["light_pos"], float(fx["intensity"]), isteps, jsteps)[..., :3]
        err[bits] = tol.atmo_error(fx["expected"], got)
    variants.set_base_variant(0)
    assert err[LLVMPIPE].max() <= 1e-4, f"{name}: {err[LLVMPIPE].max():.3g}"
    assert err[LLVMPIPE].max() <= 0.6 * err[0].max()
    assert np.median(err[LLVMPIPE]) <= 1e-6 or np.median(err[0]) == 0.0


# ---- llvmpipe's built-ins, probed live (build container only).  The witness build restates sin, cos, exp, pow, exp2, log2 as Mesa's gallivm
# evaluates them (oracle/pt_oracle.c "base variant bit 128"; Mesa is a dependency of the reference's test rig, absent from /root/reference:
# restated from its published algorithm) — and the probe below shows the restatement is EXACT: bit-identical with the live llvmpipe on
# 65,536 arguments per function.  It also measures what the contract's allowances rest on: llvmpipe's a / b, 1 / x and sqrt are correctly
# rounded, inversesqrt(x) is 1 / sqrt(x) with two roundings, a * b + c in shader code is never fused (its built-ins' own polynomials are).
_TEST_MAIN = ("#version 450 core\nlayout(local_size_x = 8, local_size_y = 8, local_size_z = 1) in;\n"
              "layout(binding = 0, rgba32f) restrict uniform image2D ImgResult;\nvoid main() {\n  ivec2 c = ivec2(gl_GlobalInvocationID.xy);\n"
              "  if (c.x >= imageSize(ImgResult).x || c.y >= imageSize(ImgResult).y) return;\n  vec4 v = imageLoad(ImgResult, c);\n"
              "  imageStore(ImgResult, c, %s);\n}\n")


def _on_llvmpipe(expr, image):
    with tempfile.NamedTemporaryFile("w", suffix=".glsl", delete=False) as f:
        f.write(_TEST_MAIN % expr)
    try:
        return ref.run_image_transform(f.name, image)
    finally:
        os.unlink(f.name)


@pytest.mark.skipif(not ref.available(), reason="needs Mesa llvmpipe + oracle/_ref/glsl_runner (build container)")
def test_llvmpipes_builtins_as_restated_are_bit_identical_with_the_live_llvmpipe(variants):
    rng = np.random.default_rng(3)
    n = 256
    img = np.zeros((n, n, 4), np.float32)
    img[..., 0] = rng.uniform(0, 2 * np.pi, (n, n))     # the integrator's angles
    img[..., 1] = rng.uniform(-20, 0.5, (n, n))         # Beer's law arguments
    img[..., 2] = rng.uniform(0, 1, (n, n))             # 1 - cos(theta)
    img[..., 3] = rng.uniform(1e-3, 64, (n, n))
    a = _on_llvmpipe("vec4(sin(v.x), cos(v.x), exp(v.y), pow(v.z, 5.0))", img)
    b = _on_llvmpipe("vec4(exp2(v.y), log2(v.w), inversesqrt(v.w), v.z * v.w + v.y)", img)
    c = _on_llvmpipe("vec4(sqrt(v.w), 1.0 / v.w, v.z / v.w, 0.0)", img)
    lib = variants.lib
    lib.pto_llvmpipe_like.restype = C.c_int
    fp = C.POINTER(C.c_float)

    def restated(which, x, y=None):
        x = np.ascontiguousarray(x, np.float32).ravel()
        y = np.ascontiguousarray(x if y is None else y, np.float32).ravel()
        out = np.empty_like(x)
        assert lib.pto_llvmpipe_like(which, x.ctypes.data_as(fp), y.ctypes.data_as(fp), x.size, out.ctypes.data_as(fp)) == 0
        return out.reshape(n, n)

    def same(u, v):
        return float((np.ascontiguousarray(u, np.float32).view(np.uint32) == np.ascontiguousarray(v, np.float32).view(np.uint32)).mean())

    five = np.full((n, n), 5.0, np.float32)
    for name, got, want in (("sin", restated(0, img[..., 0]), a[..., 0]), ("cos", restated(1, img[..., 0]), a[..., 1]),
                            ("exp", restated(2, img[..., 1]), a[..., 2]), ("pow(x, 5)", restated(3, img[..., 2], five), a[..., 3]),
                            ("exp2", restated(4, img[..., 1]), b[..., 0]), ("log2", restated(5, img[..., 3]), b[..., 1])):
        assert same(got, want) == 1.0, f"{name}: {100 * same(got, want):.3f} % bit-identical"
    w = img[..., 3]
    assert same(np.float32(1.0) / np.sqrt(w), b[..., 2]) == 1.0                       # inversesqrt = 1 / sqrt, two roundings
    assert same(img[..., 2] * w + img[..., 1], b[..., 3]) == 1.0                      # a * b + c: never fused
    assert same(np.sqrt(w), c[..., 0]) == 1.0 and same(np.float32(1.0) / w, c[..., 1]) == 1.0 and same(img[..., 2] / w, c[..., 2]) == 1.0


def test_with_llvmpipes_choices_the_restatement_renders_the_references_bits(variants):
    """The strongest form of "the oracle restates the reference's algorithm": evaluated with llvmpipe's arithmetic choices the SAME C code
    reproduces the reference's first frames BIT FOR BIT in 98.6 % of the pixels (the contract, whose choices are the GPU's: 39 %; within 1e-6:
    95.0 -> 99.97 %), and the environment-only frames in 92 - 99 %.  What is left are expression forms not yet matched; no pixel class is
    systematically off."""
    def bit_exact(bits):
        variants.set_base_variant(bits)
        same = n = 0
        for name in fixtures.names("frame_"):
            fx = fixtures.load(name)
            if fx["env"].dtype == np.uint8:
                continue  # (llvmpipe decodes sRGB8 with a cubic approximation)
            got = variants.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=1, **fixtures.kwargs(fx))[..., :3]
            eq = (got.view(np.uint32) == fx["expected"][0].view(np.uint32)).all(-1)
            same += int(eq.sum())
            n += eq.size
        variants.set_base_variant(0)
        return same / n
    contract, llvmpipe = bit_exact(0), bit_exact(LLVMPIPE)
    print(f"\n  first frames bit for bit equal to the reference: contract {contract:.1%}, with llvmpipe's choices {llvmpipe:.1%}")
    assert llvmpipe >= 0.97 and contract <= 0.5
