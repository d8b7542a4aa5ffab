"""Measure how closely the pt-f32 arithmetic (oracle == HIP, bit for bit) agrees with the committed reference fixtures and
write tests/golden/agreement.json.  tests/tolerances.py derives every fixture's own pass mark from it: measured fraction
minus 0.3 percentage points, so a regression that merely doubles the diverged-pixel fraction fails.

    python tests/golden/measure_agreement.py        (CPU only; run after regenerating fixtures or changing the contract)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fixtures  # noqa: E402
import tolerances as tol  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

oracle = graft.load_oracle().Oracle()
out = {}
for name in fixtures.names("frame_"):
    fx = fixtures.load(name)
    got, ref = fixtures.oracle_frames(oracle, fx), fx["expected"]
    srgb = fx["env"].dtype == np.uint8
    out[name] = [tol.agreement(ref[k], got[k], srgb_band=srgb) for k in range(ref.shape[0])]
for name in fixtures.names("sparse_"):
    fx = fixtures.load(name)
    got = oracle.render_pixels(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], fx["xy"], frame=0,
                               **fixtures.kwargs(fx))[..., :3]
    out[name] = [tol.agreement(fx["expected"], got)]
json.dump(out, open(os.path.join(HERE, "agreement.json"), "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(f"{k:44s}", "  ".join(f"{100 * a['within']:.3f}%" for a in v))
