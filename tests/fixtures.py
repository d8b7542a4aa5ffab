"""Helpers to load the committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from
the reference GLSL on llvmpipe) and to run the oracle / the HIP path on the same inputs."""
from __future__ import annotations

import glob
import os

import numpy as np

import configs

GOLDEN = configs.GOLDEN


def names(prefix: str) -> list[str]:
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load(name: str) -> dict:
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if "iparams" in d:
        ip, fp = d["iparams"], d["fparams"]
        d.update(width=int(ip[0]), height=int(ip[1]), num_spheres=int(ip[2]), num_cuboids=int(ip[3]), ray_depth=int(ip[4]),
                 spp=int(ip[5]), frames=int(ip[6]), focal_length=float(fp[0]), aperture=float(fp[1]))
        d["env"] = configs.load_env(str(d["env_key"]))
        d["basic"] = d["basic"].tobytes()
        d["objects"] = d["objects"].tobytes() if "objects" in d else bytes(26624)
    return d


def kwargs(fx: dict) -> dict:
    return dict(num_spheres=fx["num_spheres"], num_cuboids=fx["num_cuboids"], ray_depth=fx["ray_depth"], spp=fx["spp"],
                focal_length=fx["focal_length"], aperture=fx["aperture"])


def llvmpipe_srgb_lut() -> np.ndarray:
    """Mesa llvmpipe does NOT use the exact GL sRGB decode when sampling SRGB8_A8 textures: its JIT evaluates the
    cubic 0.3012 x^3 + 0.6935 x^2 + 0.0030 x + 0.0023 (x = byte/255) above byte 15 and byte/(12.6*255) below
    (observed on Mesa 23.2.1; error up to ~0.5 % vs the GL 4.5 section 8.24 formula).  To pin everything ELSE of the
    sampling path against llvmpipe, the pinning tests load this table into the oracle (test-only knob)."""
    f = np.float32
    xi = np.arange(256, dtype=np.float32)
    xs = (xi * f(1.0 / 255.0)).astype(np.float32)
    poly = (((f(0.3012) * xs + f(0.6935)) * xs + f(0.0030)) * xs + f(0.0023)).astype(np.float32)
    lin = (xi * f(1.0 / (12.6 * 255.0))).astype(np.float32)
    return np.where(xi <= 15, lin, poly).astype(np.float32)


def oracle_frames(oracle, fx: dict, threads=None) -> np.ndarray:
    """(len(frame_indices), H, W, 3): accumulated image after each stored frame index"""
    out = oracle.render(fx["width"], fx["height"], fx["basic"], fx["objects"], fx["env"], num_frames=fx["frames"],
                        dump_each=True, threads=threads, **kwargs(fx))
    idx = fx["frame_indices"] if "frame_indices" in fx else np.array([fx["frames"] - 1])
    return out[idx][..., :3]


def hip_tracer(pkg, fx: dict, **extra):
    pt = pkg.PathTracer(fx["env"], fx["width"], fx["height"], fx["ray_depth"], fx["spp"], fx["focal_length"],
                        fx["aperture"], **extra)
    objs = np.frombuffer(fx["objects"], dtype=np.uint8)
    pt.GameObjectsUBO.SubData(0, objs.nbytes, objs)
    pt.NumSpheres = fx["num_spheres"]
    pt.NumCuboids = fx["num_cuboids"]
    pt.UploadBasicData(fx["basic"])
    return pt


def hip_frames(pkg, fx: dict) -> np.ndarray:
    pt = hip_tracer(pkg, fx)
    idx = set(int(i) for i in (fx["frame_indices"] if "frame_indices" in fx else [fx["frames"] - 1]))
    outs = []
    for f in range(fx["frames"]):
        pt.Render()
        if f in idx:
            outs.append(pt.Result[..., :3].copy())
    pt.Dispose()
    return np.stack(outs)


def edge_nanenv_check(fx: dict, acc_each) -> dict:
    """The quirk scene (tests/golden/edge_nanenv_*.npz, make_golden.py `edge`): `acc_each` = this implementation's running means after
    every frame, (frames, H, W, 3).  Per SAMPLE, every gross difference from the reference either ends in a NaN-direction
    environment lookup (undefined in GL; flagged in the fixture by the oracle's diagnostic build) or belongs to the usual branch-flip
    budget; without the flagged samples the two accumulations agree like every other scene's."""
    def samples(a):
        a = np.asarray(a, dtype=np.float64)
        s_ = np.empty_like(a)
        s_[0] = a[0]
        for k in range(1, len(a)):
            s_[k] = (k + 1) * a[k] - k * a[k - 1]
        return s_
    sr, so = samples(fx["expected_each"]), samples(acc_each)
    nan_env = np.unpackbits(fx["nan_env"])[:sr[..., 0].size].reshape(sr.shape[:-1]).astype(bool)
    gross = np.abs(so - sr).max(-1) > 1e-2 * np.maximum(1.0, np.abs(sr).max(-1))
    so_m, sr_m = np.where(nan_env[..., None], 0.0, so), np.where(nan_env[..., None], 0.0, sr)
    return {"samples": int(gross.size), "nan_env": int(nan_env.sum()), "gross": int(gross.sum()), "gross_unflagged": int((gross & ~nan_env).sum()),
            "masked_mean_rel_err": float(abs(so_m.mean() - sr_m.mean()) / sr_m.mean())}
