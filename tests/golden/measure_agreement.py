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
# Fidelity study: what the same fixtures would score with correctly rounded 1/x, sqrt, 1/sqrt and the
# literal slab division (oracle/_build/libpt_oracle_exact.so) — llvmpipe's `/` and sqrt are correctly rounded, the contract's
# Newton sequences are not.  Recorded next to the contract's own figure; the pass marks above never use it.
exact = graft.load_oracle().Oracle(exact=True)
study = {}
for name in fixtures.names("frame_"):
    fx = fixtures.load(name)
    got, ref = fixtures.oracle_frames(exact, fx), fx["expected"]
    srgb = fx["env"].dtype == np.uint8
    study[name] = {"contract_within": float(np.mean([a["within"] for a in out[name]])),
                   "exact_divsqrt_within": float(np.mean([tol.agreement(ref[k], got[k], srgb_band=srgb)["within"] for k in range(ref.shape[0])]))}
for name in fixtures.names("sparse_"):
    fx = fixtures.load(name)
    got = exact.render_pixels(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], fx["xy"], frame=0,
                              **fixtures.kwargs(fx))[..., :3]
    study[name] = {"contract_within": out[name][0]["within"], "exact_divsqrt_within": tol.agreement(fx["expected"], got)["within"]}
for v in study.values():
    v["gain_points"] = round(100 * (v["exact_divsqrt_within"] - v["contract_within"]), 4)
out["_fidelity_study_exact_divsqrt"] = study
json.dump(out, open(os.path.join(HERE, "agreement.json"), "w"), indent=1, sort_keys=True)
for k, v in out.items():
    if k.startswith("_"):
        continue
    print(f"{k:44s}", "  ".join(f"{100 * a['within']:.3f}%" for a in v), f"   exact div/sqrt would score {100 * study[k]['exact_divsqrt_within']:.3f}% ({study[k]['gain_points']:+.3f} pt)")
