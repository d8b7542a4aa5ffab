"""The stated tolerances of the parity claims (referenced from DESIGN.md).

1. HIP kernel vs CPU oracle: BIT-EXACT (both implement the same "pt-f32" arithmetic contract), every pixel.
2. Oracle (and therefore HIP) vs the reference GLSL executed by Mesa llvmpipe: GLSL leaves the precision of
   sin/cos/exp/pow/inversesqrt/normalize and fma contraction implementation-defined, and the integrator branches on
   RNG draws compared with computed floats (compute.glsl:169,201,208,234,247), so a last-bit difference flips a
   branch in a small fraction of pixels.  The claim is therefore two-part:
     (a) at least min_fraction(fixture, k) of the pixels agree within  REL_TOL * max(1, |reference|)  per channel;
     (b) the image means agree within MEAN_REL_TOL (no bias).
   The pass mark of (a) is PER FIXTURE and FROZEN (tests/golden/thresholds.json): 99.2-99.9 % for the default / glass /
   random-material scenes, 98.0 % for the 256-sphere scene (263 brute-force candidates per bounce: more near-ties).  The marks
   were set ONCE, at the end of round 3, from the agreement measured then minus 0.3 percentage points, and are no longer derived
   from the implementation under test: tests/test_thresholds_frozen.py pins the file's SHA-256 (THRESHOLDS_SHA256 below), so a
   change of the arithmetic that lowers the agreement fails instead of moving its own pass mark.  tests/golden/agreement.json
   and measure_agreement.py remain as a REPORT of what the oracle scores (not read by any test).  A fixture without a frozen
   mark falls back to PIXEL_FRACTION.  Every test reports the achieved numbers in the terminal summary (tests/conftest.py).
"""
import json
import os

import numpy as np

REL_TOL = 1e-4
PIXEL_FRACTION = 0.975   # fallback only (fixtures newer than agreement.json)
THRESHOLDS_SHA256 = "daf5fc5dbcbd08d2fbf19cf26f975c2192d45ac19af830116831854edb243bb6"  # of tests/golden/thresholds.json
MEAN_REL_TOL = 2e-3
# llvmpipe decodes sRGB8 texels with a cubic approximation (<= 0.6 % off the GL formula, fixtures.llvmpipe_srgb_lut); the
# product uses the exact GL 4.5 table, so sRGB-environment fixtures are compared inside this wider band
SRGB_REL_TOL = 8e-3
SRGB_MEAN_REL_TOL = 6e-3
# environment-only frames (no chaotic branching): every pixel must agree
ENV_REL_TOL = 5e-5
# function-level micro fixtures (absolute)
MICRO_ABS_TOL = 2e-5

_THRESHOLDS = None


def thresholds_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "thresholds.json")


def frozen(name: str):
    """The frozen pass marks of fixture `name` (one per accumulated frame), or None."""
    global _THRESHOLDS
    if _THRESHOLDS is None:
        _THRESHOLDS = json.load(open(thresholds_path()))["fixtures"]
    return _THRESHOLDS.get(name)


def min_fraction(name: str, k: int = 0) -> float:
    m = frozen(name)
    if not m or k >= len(m):
        return PIXEL_FRACTION
    return m[k]


def within(ref, got, rel_tol=REL_TOL):
    """per-pixel boolean: all channels inside rel_tol * max(1,|ref|)"""
    import numpy as np
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    tol = rel_tol * np.maximum(1.0, np.abs(ref))
    return (np.abs(ref - got) <= tol).all(axis=-1)


def agreement(ref, got, srgb_band=False) -> dict:
    """The numbers every layer-2 test reports: fraction of pixels inside the band (NaN == NaN counts as agreeing: the
    reference produces NaN pixels by design), fraction bit-identical, relative error of the image mean,
    mean absolute error over finite pixels."""
    import numpy as np
    ref = np.asarray(ref, dtype=np.float32)
    got = np.asarray(got, dtype=np.float32)
    both_nan = np.isnan(ref).any(-1) & np.isnan(got).any(-1)
    ok = within(ref, got, SRGB_REL_TOL if srgb_band else REL_TOL) | both_nan
    same = (ref.view(np.uint32) == got.view(np.uint32)).all(-1)
    fin = np.isfinite(ref).all(-1) & np.isfinite(got).all(-1)
    rm, gm = float(ref[fin].mean(dtype=np.float64)), float(got[fin].mean(dtype=np.float64))
    return {"within": float(ok.mean()), "bit_identical": float(same.mean()),
            "mean_rel_err": abs(rm - gm) / max(abs(rm), 1e-30),
            "mean_abs_err": float(np.abs(ref[fin].astype(np.float64) - got[fin]).mean())}


# ---- per-pixel convergence against the reference (tests/golden/convergence_*.npz: mean and standard error of the reference's own
# 4,096-frame accumulation per pixel and channel).  Our accumulation uses the SAME random streams, so it is strongly correlated
# with the reference's (measured: rms z 0.015 - 0.075, max |z| < 1): only the pixels whose branches flipped in some frame differ at
# all.  A rare-path bug with any systematic effect shows up as z-scores far beyond these marks long before it moves an image mean.
CONV_MAX_ABS_Z = 2.5      # every pixel, every channel
CONV_RMS_Z = 0.25         # over the image
CONV_MEAN_REL_TOL = 1e-4  # image mean


def convergence_stats(mean_ref, stderr_ref, got) -> dict:
    import numpy as np
    mean_ref = np.asarray(mean_ref, dtype=np.float64)
    se = np.maximum(np.asarray(stderr_ref, dtype=np.float64), 1e-12)
    got = np.asarray(got, dtype=np.float64)
    nan_ref, nan_got = np.isnan(mean_ref), np.isnan(got)
    z = (got - mean_ref) / se
    ok = ~(nan_ref | nan_got)
    az = np.abs(z[ok])
    return {"max_abs_z": float(az.max()), "rms_z": float(np.sqrt((az ** 2).mean())), "frac_abs_z_gt_1": float((az > 1).mean()),
            "mean_rel_err": float(abs(got[ok].mean() - mean_ref[ok].mean()) / abs(mean_ref[ok].mean())),
            "nan_mismatch": int((nan_ref != nan_got).sum()), "pixels": int(ok.all(-1).sum())}


# ---- atmosphere cubes (AtmosphericScattering/compute.glsl on llvmpipe vs the pt-f32 restatement), FROZEN in round 6 from the measured
# per-texel agreement: error = |got - ref| / max(|ref|, 1e-3 max|ref|); per fixture (largest error allowed = measured x 1.5, smallest share
# of texels within 1e-4 = measured - 0.3 points).  Measured: 1.49e-4 / 99.63 %, 3.45e-4 / 97.84 %, 1.48e-4 / 99.94 %.  Until round 5 the
# gate was a flat 2e-3 for every cube.  Moving a mark is a visible diff of this file.
ATMO_MARKS = {"atmo_24_few_steps": (2.3e-4, 0.9933), "atmo_32_default": (5.2e-4, 0.9754), "atmo_48_noon": (2.3e-4, 0.9964)}


def atmo_error(ref, got):
    scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max())
    return np.abs(got - ref) / scale
