"""The stated tolerances of the parity claims (referenced from DESIGN.md).

1. HIP kernel vs CPU oracle: BIT-EXACT (both implement the same "pt-f32" arithmetic contract), every pixel.
2. Oracle (and therefore HIP) vs the reference GLSL executed by Mesa llvmpipe: GLSL leaves the precision of
   sin/cos/exp/pow/inversesqrt/normalize and fma contraction implementation-defined, and the integrator branches on
   RNG draws compared with computed floats (compute.glsl:169,201,208,234,247), so a last-bit difference flips a
   branch in a small fraction of pixels.  The claim is therefore two-part:
     (a) at least PIXEL_FRACTION of the pixels agree within  REL_TOL * max(1, |reference|)  per channel;
     (b) the image means agree within MEAN_REL_TOL (no bias).
   Measured on the committed fixtures: 98.4 % .. 100 % of pixels inside the band (worst: 256-sphere scene).
"""
REL_TOL = 1e-4
PIXEL_FRACTION = 0.975
MEAN_REL_TOL = 2e-3
# environment-only frames (no chaotic branching): every pixel must agree
ENV_REL_TOL = 5e-5
# function-level micro fixtures (absolute)
MICRO_ABS_TOL = 2e-5


def within(ref, got):
    """per-pixel boolean: all channels inside REL_TOL * max(1,|ref|)"""
    import numpy as np
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    tol = REL_TOL * np.maximum(1.0, np.abs(ref))
    return (np.abs(ref - got) <= tol).all(axis=-1)
