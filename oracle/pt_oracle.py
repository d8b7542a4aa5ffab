"""ctypes binding for oracle/_build/libpt_oracle.so — the CPU restatement of the reference integrator.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libpt_oracle.so")
LIB_TRUEDIV_PATH = os.path.join(HERE, "_build", "libpt_oracle_truediv.so")
LIB_EXACT_PATH = os.path.join(HERE, "_build", "libpt_oracle_exact.so")  # fidelity study: IEEE 1/x, sqrt, 1/sqrt, a/b
LIB_NANMARK_PATH = os.path.join(HERE, "_build", "libpt_oracle_nanmark.so")  # diagnostic: env lookups with a NaN direction return 1000
LIB_MARGINS_PATH = os.path.join(HERE, "_build", "libpt_oracle_margins.so")  # decision margins per pixel (tests/test_decision_margins.py)
LIB_PERTURB_PATH = os.path.join(HERE, "_build", "libpt_oracle_perturb.so")  # witness build: one primitive a chosen number of ulps off


def build(force: bool = False) -> None:
    src = os.path.join(HERE, "pt_oracle.c")
    libs = (LIB_PATH, LIB_TRUEDIV_PATH, LIB_EXACT_PATH, LIB_NANMARK_PATH, LIB_MARGINS_PATH, LIB_PERTURB_PATH)
    stale = not all(os.path.exists(p) for p in libs) or min(os.path.getmtime(p) for p in libs) < os.path.getmtime(src)
    if force or stale:
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s", "all"], check=True, capture_output=True)


class PtoParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("numSpheres", C.c_int), ("numCuboids", C.c_int),
                ("rayDepth", C.c_int), ("spp", C.c_int), ("focalLength", C.c_float), ("apertureDiameter", C.c_float),
                ("envSize", C.c_int), ("envFormat", C.c_int)]


_fp = C.POINTER(C.c_float)


def _ptr(a, t=_fp):
    return a.ctypes.data_as(t)


class Oracle:
    def __init__(self, true_division: bool = False, exact: bool = False, mark_nan_env: bool = False, margins: bool = False, perturb: bool = False):
        build()
        self.lib = C.CDLL(LIB_PERTURB_PATH if perturb else LIB_MARGINS_PATH if margins else LIB_NANMARK_PATH if mark_nan_env else LIB_EXACT_PATH if exact else (LIB_TRUEDIV_PATH if true_division else LIB_PATH))
        L = self.lib
        L.pto_render_frame_margins.restype = C.c_int
        L.pto_render_frame_margins.argtypes = [C.POINTER(PtoParams), _fp, _fp, C.c_void_p, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
        L.pto_render_pixels_margins.restype = C.c_int
        L.pto_render_pixels_margins.argtypes = [C.POINTER(PtoParams), _fp, _fp, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, _fp, _fp, _fp]
        L.pto_render_frame.restype = C.c_int
        L.pto_render_frame.argtypes = [C.POINTER(PtoParams), _fp, _fp, C.c_void_p, _fp, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.POINTER(C.c_uint64)]
        L.pto_render_pixels.restype = C.c_int
        L.pto_render_pixels.argtypes = [C.POINTER(PtoParams), _fp, _fp, C.c_void_p, C.POINTER(C.c_int), C.c_int,
                                        C.c_int, _fp, _fp]
        L.pto_pcg_hash.restype = C.c_uint32
        L.pto_pcg_hash.argtypes = [C.POINTER(C.c_uint32)]
        L.pto_rand01.restype = C.c_float
        L.pto_rand01.argtypes = [C.POINTER(C.c_uint32)]
        L.pto_pixel_seed.restype = C.c_uint32
        L.pto_pixel_seed.argtypes = [C.c_int, C.c_int, C.c_int]
        L.pto_sincos.argtypes = [C.c_float, _fp, _fp]
        L.pto_exp.restype = C.c_float
        L.pto_exp.argtypes = [C.c_float]
        L.pto_ray_sphere.restype = C.c_int
        L.pto_ray_sphere.argtypes = [_fp, _fp, _fp, _fp]
        L.pto_ray_cuboid.restype = C.c_int
        L.pto_ray_cuboid.argtypes = [_fp, _fp, _fp, _fp, _fp]
        L.pto_cuboid_normal.argtypes = [_fp, _fp, _fp, _fp]
        L.pto_sample_env.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp, _fp]
        L.pto_srgb_to_linear.restype = C.c_float
        L.pto_srgb_to_linear.argtypes = [C.c_int]
        L.pto_pow5.restype = C.c_float
        L.pto_pow5.argtypes = [C.c_float]
        L.pto_fresnel_schlick.restype = C.c_float
        L.pto_fresnel_schlick.argtypes = [C.c_float, C.c_float, C.c_float]
        L.pto_refract.argtypes = [_fp, _fp, C.c_float, _fp]
        L.pto_reflect.argtypes = [_fp, _fp, _fp]
        L.pto_cosine_sample_hemisphere.argtypes = [_fp, C.POINTER(C.c_uint32), _fp]
        L.pto_normalize.argtypes = [_fp, _fp]
        L.pto_set_srgb_lut.argtypes = [_fp]
        L.pto_postprocess.restype = C.c_int
        L.pto_postprocess.argtypes = [_fp, C.c_int, _fp, C.POINTER(C.c_uint8)]
        L.pto_log.restype = C.c_float
        L.pto_log.argtypes = [C.c_float]
        L.pto_set_perturbation.restype = C.c_int
        L.pto_set_perturbation.argtypes = [C.c_int, C.c_int]
        _ip = C.POINTER(C.c_int)
        L.pto_render_pixel_variant.restype = C.c_int
        L.pto_render_pixel_variant.argtypes = [C.c_void_p, _fp, _fp, C.c_void_p, C.c_int, C.c_int, C.c_int, _fp, _ip, C.c_int, _ip, C.c_int, _fp]
        L.pto_set_nan_env.restype = C.c_int
        L.pto_set_nan_env.argtypes = [_fp]
        L.pto_witness_search.restype = C.c_int
        L.pto_witness_search.argtypes = [C.c_void_p, _fp, _fp, C.c_void_p, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_double, C.c_float, C.c_int,
                                         _ip, _ip, _ip, _ip, _fp]
        L.pto_atmosphere.restype = C.c_int
        L.pto_atmosphere.argtypes = [_fp, _fp, C.c_float, C.c_int, C.c_int, C.c_int, _fp, C.c_int]

    def set_perturbation(self, prim: int, ulps: int) -> None:
        """Oracle(perturb=True) only: primitive `prim` (0 rcp, 1 rsqrt, 2 sqrt, 3 sin, 4 cos, 5 exp) returns results `ulps` units in the
        last place further from zero from now on (negative: nearer; 0 = the contract again)."""
        assert self.lib.pto_set_perturbation(prim, ulps) == 0, "this oracle build has no perturbation hooks (Oracle(perturb=True))"

    def set_unfused(self, on: bool) -> None:
        """Oracle(perturb=True) only: every a * b + c outside the primitives with two roundings (llvmpipe never fuses) / the contract again."""
        assert self.lib.pto_set_unfused(int(on)) == 0, "this oracle build has no witness hooks (Oracle(perturb=True))"

    def set_base_variant(self, bits: int) -> None:
        """Oracle(perturb=True) only: the conforming implementation the witness searches and replays run around — 1 never fused, 2 correctly
        rounded 1/x, sqrt, 1/sqrt, 4 literal divisions; 7 = llvmpipe's choices, 0 = the contract."""
        assert self.lib.pto_set_base_variant(int(bits)) == 0, "this oracle build has no witness hooks (Oracle(perturb=True))"

    def set_ensemble(self, seed: int, amplitude: int = 16) -> None:
        """Oracle(perturb=True) only: the library becomes ensemble member `seed` (0: off) — ONE conforming implementation that differs from the
        contract everywhere at once: every primitive call up to min(amplitude, its allowance) ulps off, every multiply-add fused or not, every
        division literal or by reciprocal, each a fixed function of the member and the operands (pt_oracle.c, ens_hash)."""
        if not hasattr(self.lib.pto_set_ensemble, "_typed"):
            self.lib.pto_set_ensemble.restype = C.c_int
            self.lib.pto_set_ensemble.argtypes = [C.c_uint, C.c_int]
        assert self.lib.pto_set_ensemble(int(seed) & 0xFFFFFFFF, int(amplitude)) == 0, "this oracle build has no witness hooks (Oracle(perturb=True))"

    def set_signature_alpha(self, on: bool) -> None:
        """Oracle(perturb=True) only: alpha = 23 bits of the pixel's path signature (objects hit, lobes taken, how each path ended; chained
        over samples and accumulated frames) as a float in [1, 2) instead of 1.0."""
        assert self.lib.pto_set_signature_alpha(int(on)) == 0, "this oracle build has no witness hooks (Oracle(perturb=True))"

    def set_nan_env(self, rgb) -> None:
        """Oracle(perturb=True) only: what texture(env, NaN direction) returns (None: the contract's clamped lookup)."""
        v = None if rgb is None else np.ascontiguousarray(rgb, np.float32)
        assert self.lib.pto_set_nan_env(None if v is None else _ptr(v)) == 0, "this oracle build has no witness hooks (Oracle(perturb=True))"

    def witness_search(self, width, height, basic_ubo, objects_ubo, env_faces, xy, ref, band, *, num_spheres, num_cuboids, ray_depth, spp=1,
                       focal_length=20.0, aperture=0.14, frame=0, last=None, close_gap=1e-6, max_flips=256):
        """Oracle(perturb=True) only.  For every listed pixel: a conforming neighbour of the contract — one comparison inverted, one call of
        one primitive a few ulps off, a pair of inverted comparisons, or several calls a few ulps off — that lands inside
        band * max(1, |ref|) of `ref` (n x 3); see pto_witness_search.  Returns a list of dicts: kind (0 = none, 1 .. 5, 9 = a neighbour
        inside; 7, 8 = the path ends in texture(env, NaN), undefined in GL), flips, sites
        [(primitive, call, ulps)], value (4,), evaluated, unstable_calls, largest_move (bands), distance (bands, of the nearest variant)."""
        basic, objs, env = self._inputs(basic_ubo, objects_ubo, env_faces)
        p = self._params(width, height, num_spheres, num_cuboids, ray_depth, spp, focal_length, aperture, env)
        xy = np.ascontiguousarray(xy, dtype=np.int32)
        n = xy.shape[0]
        last = np.zeros((n, 4), np.float32) if last is None else np.ascontiguousarray(last, np.float32)
        ref = np.ascontiguousarray(ref, np.float32)
        ip = C.POINTER(C.c_int)
        res = []
        for i in range(n):
            flips, sites, ns, stats, out = np.zeros(3, np.int32), np.zeros(96, np.int32), C.c_int(0), np.zeros(4, np.int32), np.zeros(4, np.float32)
            kind = self.lib.pto_witness_search(C.byref(p), _ptr(basic), _ptr(objs), env.ctypes.data_as(C.c_void_p), int(xy[i, 0]), int(xy[i, 1]),
                                               frame, _ptr(last[i]), _ptr(ref[i]), float(band), close_gap, max_flips,
                                               flips.ctypes.data_as(ip), sites.ctypes.data_as(ip), C.byref(ns), stats.ctypes.data_as(ip), _ptr(out))
            assert kind >= 0, "this oracle build has no witness hooks (Oracle(perturb=True))"
            res.append(dict(kind=kind, flips=tuple(int(f) for f in flips), sites=[tuple(int(v) for v in sites[3 * t:3 * t + 3]) for t in range(ns.value)],
                            value=out, evaluated=int(stats[0]), unstable_calls=int(stats[1]), largest_move=stats[2] / 1000.0, distance=stats[3] / 1000.0))
        return res

    def render_pixel_variant(self, width, height, basic_ubo, objects_ubo, env_faces, x, y, *, num_spheres, num_cuboids, ray_depth, spp=1,
                             focal_length=20.0, aperture=0.14, frame=0, last=None, flips=(-1, -1, -1), sites=(), pow_neg_nan=0):
        """Oracle(perturb=True) only: one pixel with the listed comparisons inverted and / or the listed primitive calls shifted
        (sites = [(primitive, call, ulps)]; pow_neg_nan 1 / 2: pow() of a negative / nearly zero base is NaN; see witness_search).  Returns (pixel (4,), number of comparisons the evaluation passed)."""
        basic, objs, env = self._inputs(basic_ubo, objects_ubo, env_faces)
        p = self._params(width, height, num_spheres, num_cuboids, ray_depth, spp, focal_length, aperture, env)
        last = np.zeros(4, np.float32) if last is None else np.ascontiguousarray(last, np.float32)
        fl = np.array((list(flips) + [-1, -1, -1])[:3], np.int32)
        st = np.array([v for site in sites for v in site] or [0], np.int32)
        out = np.zeros(4, np.float32)
        ip = C.POINTER(C.c_int)
        nd = self.lib.pto_render_pixel_variant(C.byref(p), _ptr(basic), _ptr(objs), env.ctypes.data_as(C.c_void_p), int(x), int(y), frame, _ptr(last),
                                               fl.ctypes.data_as(ip), len(sites), st.ctypes.data_as(ip), pow_neg_nan, _ptr(out))
        assert nd >= 0, "this oracle build has no witness hooks (Oracle(perturb=True))"
        return out, nd

    # ---------------------------------------------------------------- frames
    @staticmethod
    def _inputs(basic_ubo: bytes, objects_ubo: bytes, env_faces: np.ndarray):
        basic = np.frombuffer(basic_ubo, dtype=np.float32).copy()
        objs = np.frombuffer(objects_ubo, dtype=np.float32).copy()
        env = np.ascontiguousarray(env_faces)
        assert basic.size == 36 and objs.size == 6656
        assert env.dtype in (np.float32, np.uint8) and env.shape[0] == 6 and env.shape[3] == 4
        return basic, objs, env

    @staticmethod
    def _params(width, height, num_spheres, num_cuboids, ray_depth, spp, focal_length, aperture, env):
        return PtoParams(width, height, int(num_spheres), int(num_cuboids), ray_depth, spp, focal_length, aperture,
                         env.shape[1], 0 if env.dtype == np.float32 else 1)

    def render(self, width, height, basic_ubo, objects_ubo, env_faces, *, num_spheres, num_cuboids, ray_depth, spp=1,
               focal_length=20.0, aperture=0.14, frame_start=0, num_frames=1, y0=0, rows=None, threads=None,
               image=None, dump_each=False, want_stats=False):
        """Accumulate frames [frame_start, frame_start+num_frames) onto `image` (zeros if None).
        Returns (rows, W, 4) float32 (or (frames, rows, W, 4) with dump_each); with want_stats also a dict."""
        basic, objs, env = self._inputs(basic_ubo, objects_ubo, env_faces)
        p = self._params(width, height, num_spheres, num_cuboids, ray_depth, spp, focal_length, aperture, env)
        rows = height - y0 if rows is None else rows
        threads = threads or os.cpu_count() or 1
        img = np.zeros((rows, width, 4), dtype=np.float32) if image is None else np.ascontiguousarray(image, np.float32)
        stats = (C.c_uint64 * 6)()
        total = np.zeros(6, dtype=np.uint64)
        dumps = []
        for f in range(frame_start, frame_start + num_frames):
            rc = self.lib.pto_render_frame(C.byref(p), _ptr(basic), _ptr(objs), env.ctypes.data_as(C.c_void_p),
                                           _ptr(img), y0, rows, f, threads, stats if want_stats else None)
            assert rc == 0
            total += np.array(list(stats), dtype=np.uint64)
            if dump_each:
                dumps.append(img.copy())
        out = np.stack(dumps) if dump_each else img
        if want_stats:
            keys = ["samples", "bounces", "sphere_tests", "cuboid_tests", "env_lookups", "reserved"]
            return out, {k: int(v) for k, v in zip(keys, total)}
        return out

    def render_with_margins(self, width, height, basic_ubo, objects_ubo, env_faces, *, num_spheres, num_cuboids, ray_depth, spp=1,
                            focal_length=20.0, aperture=0.14, num_frames=1, threads=None, dump_each=False):
        """Oracle(margins=True) only: frames [0, num_frames) accumulated from zero -> (image (H, W, 4), margin (H, W), cont (H, W)):
        per pixel the smallest relative error eps of the arithmetic's primitives that flips one of the data-dependent comparisons of
        any frame so far, and the largest flip-free absolute colour error per unit eps of any frame so far (pt_oracle.c,
        PT_ORACLE_MARGINS); with dump_each all three per frame."""
        basic, objs, env = self._inputs(basic_ubo, objects_ubo, env_faces)
        p = self._params(width, height, num_spheres, num_cuboids, ray_depth, spp, focal_length, aperture, env)
        threads = threads or os.cpu_count() or 1
        img = np.zeros((height, width, 4), dtype=np.float32)
        cum = np.full((height, width), np.inf, dtype=np.float32)
        cont = np.zeros((height, width), dtype=np.float32)
        m = np.empty((height, width, 2), dtype=np.float32)
        imgs, margins, conts = [], [], []
        for f in range(num_frames):
            rc = self.lib.pto_render_frame_margins(C.byref(p), _ptr(basic), _ptr(objs), env.ctypes.data_as(C.c_void_p), _ptr(img), 0, height, f,
                                                   threads, _ptr(m))
            assert rc == 0, "this oracle build records no margins (Oracle(margins=True))"
            cum = np.minimum(cum, m[..., 0])
            cont = np.maximum(cont, m[..., 1])
            if dump_each:
                imgs.append(img.copy())
                margins.append(cum.copy())
                conts.append(cont.copy())
        return (np.stack(imgs), np.stack(margins), np.stack(conts)) if dump_each else (img, cum, cont)

    def render_pixels_margins(self, width, height, basic_ubo, objects_ubo, env_faces, xy, *, num_spheres, num_cuboids,
                              ray_depth, spp=1, focal_length=20.0, aperture=0.14, frame=0, last=None):
        """render_pixels + the listed pixels' (margin, cont) of this frame (Oracle(margins=True)): (n, 4), (n,), (n,)."""
        basic, objs, env = self._inputs(basic_ubo, objects_ubo, env_faces)
        p = self._params(width, height, num_spheres, num_cuboids, ray_depth, spp, focal_length, aperture, env)
        xy = np.ascontiguousarray(xy, dtype=np.int32)
        n = xy.shape[0]
        last = np.zeros((n, 4), np.float32) if last is None else np.ascontiguousarray(last, np.float32)
        out = np.zeros((n, 4), dtype=np.float32)
        m = np.zeros((n, 2), dtype=np.float32)
        self.lib.pto_render_pixels_margins(C.byref(p), _ptr(basic), _ptr(objs), env.ctypes.data_as(C.c_void_p),
                                           xy.ctypes.data_as(C.POINTER(C.c_int)), n, frame, _ptr(last), _ptr(out), _ptr(m))
        return out, m[:, 0].copy(), m[:, 1].copy()

    def render_pixels(self, width, height, basic_ubo, objects_ubo, env_faces, xy, *, num_spheres, num_cuboids,
                      ray_depth, spp=1, focal_length=20.0, aperture=0.14, frame=0, last=None):
        basic, objs, env = self._inputs(basic_ubo, objects_ubo, env_faces)
        p = self._params(width, height, num_spheres, num_cuboids, ray_depth, spp, focal_length, aperture, env)
        xy = np.ascontiguousarray(xy, dtype=np.int32)
        n = xy.shape[0]
        last = np.zeros((n, 4), np.float32) if last is None else np.ascontiguousarray(last, np.float32)
        out = np.zeros((n, 4), dtype=np.float32)
        self.lib.pto_render_pixels(C.byref(p), _ptr(basic), _ptr(objs), env.ctypes.data_as(C.c_void_p),
                                   xy.ctypes.data_as(C.POINTER(C.c_int)), n, frame, _ptr(last), _ptr(out))
        return out

    def atmosphere(self, size, atmo_ubo: bytes, light_pos, light_intensity=15.0, i_steps=50, j_steps=15, threads=None):
        ubo = np.frombuffer(atmo_ubo, dtype=np.float32).copy()
        assert ubo.size == 116
        lp = np.ascontiguousarray(light_pos, dtype=np.float32)
        out = np.zeros((6, size, size, 4), dtype=np.float32)
        self.lib.pto_atmosphere(_ptr(ubo), _ptr(lp), light_intensity, size, i_steps, j_steps, _ptr(out),
                                threads or os.cpu_count() or 1)
        return out

    def postprocess(self, image):
        """ACES + gamma of PostProcessing/fragment.glsl on an (..., 4) float32 image -> (float (..., 3), uint8 (..., 4))."""
        img = np.ascontiguousarray(image, dtype=np.float32)
        n = img.size // 4
        of = np.zeros(img.shape[:-1] + (3,), np.float32)
        ou = np.zeros(img.shape[:-1] + (4,), np.uint8)
        self.lib.pto_postprocess(_ptr(img), n, _ptr(of), ou.ctypes.data_as(C.POINTER(C.c_uint8)))
        return of, ou

    def log(self, x):
        return self.lib.pto_log(float(x))

    def _array_fn(self, name, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty_like(x)
        fn = getattr(self.lib, name)
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        fn(x.ctypes.data, y.ctypes.data, x.size)
        return y

    def rcp(self, x):
        return self._array_fn("pto_rcp_array", x)

    def rsqrt(self, x):
        return self._array_fn("pto_rsqrt_array", x)

    def sqrt(self, x):
        return self._array_fn("pto_sqrt_array", x)

    # ---------------------------------------------------------------- micro helpers
    def rand_stream(self, seed: int, n: int):
        s = C.c_uint32(seed)
        return np.array([self.lib.pto_rand01(C.byref(s)) for _ in range(n)], dtype=np.float32)

    def hash_stream(self, seed: int, n: int):
        s = C.c_uint32(seed)
        return np.array([self.lib.pto_pcg_hash(C.byref(s)) for _ in range(n)], dtype=np.uint32)

    def pixel_seed(self, x, y, frame):
        return int(self.lib.pto_pixel_seed(x, y, frame))

    def sincos(self, a):
        s, c = C.c_float(), C.c_float()
        self.lib.pto_sincos(float(a), C.byref(s), C.byref(c))
        return s.value, c.value

    def exp(self, x):
        return self.lib.pto_exp(float(x))

    def ray_sphere(self, o, d, pos_r):
        o, d, pos_r = (np.ascontiguousarray(a, np.float32) for a in (o, d, pos_r))
        t = np.zeros(2, np.float32)
        hit = self.lib.pto_ray_sphere(_ptr(o), _ptr(d), _ptr(pos_r), _ptr(t))
        return bool(hit), float(t[0]), float(t[1])

    def ray_cuboid(self, o, d, mn, mx):
        o, d, mn, mx = (np.ascontiguousarray(a, np.float32) for a in (o, d, mn, mx))
        t = np.zeros(2, np.float32)
        hit = self.lib.pto_ray_cuboid(_ptr(o), _ptr(d), _ptr(mn), _ptr(mx), _ptr(t))
        return bool(hit), float(t[0]), float(t[1])

    def cuboid_normal(self, mn, mx, p):
        mn, mx, p = (np.ascontiguousarray(a, np.float32) for a in (mn, mx, p))
        n = np.zeros(3, np.float32)
        self.lib.pto_cuboid_normal(_ptr(mn), _ptr(mx), _ptr(p), _ptr(n))
        return n

    def pow5(self, x):
        return self.lib.pto_pow5(float(x))

    def fresnel_schlick(self, cos_theta, n1, n2):
        return self.lib.pto_fresnel_schlick(float(cos_theta), float(n1), float(n2))

    def refract(self, i, n, eta):
        i, n = (np.ascontiguousarray(a, np.float32) for a in (i, n))
        out = np.zeros(3, np.float32)
        self.lib.pto_refract(_ptr(i), _ptr(n), float(eta), _ptr(out))
        return out

    def reflect(self, i, n):
        i, n = (np.ascontiguousarray(a, np.float32) for a in (i, n))
        out = np.zeros(3, np.float32)
        self.lib.pto_reflect(_ptr(i), _ptr(n), _ptr(out))
        return out

    def normalize(self, v):
        v = np.ascontiguousarray(v, np.float32)
        out = np.zeros(3, np.float32)
        self.lib.pto_normalize(_ptr(v), _ptr(out))
        return out

    def cosine_sample_hemisphere(self, n, seed: int):
        n = np.ascontiguousarray(n, np.float32)
        out = np.zeros(3, np.float32)
        s = C.c_uint32(seed)
        self.lib.pto_cosine_sample_hemisphere(_ptr(n), C.byref(s), _ptr(out))
        return out, s.value

    def set_srgb_lut(self, lut256=None):
        """Test-only: override the exact sRGB8 decode table (None restores it)."""
        if lut256 is None:
            self.lib.pto_set_srgb_lut(None)
        else:
            self._lut = np.ascontiguousarray(lut256, np.float32)
            assert self._lut.size == 256
            self.lib.pto_set_srgb_lut(_ptr(self._lut))

    def sample_env(self, env_faces, direction):
        env = np.ascontiguousarray(env_faces)
        d = np.ascontiguousarray(direction, np.float32)
        out = np.zeros(3, np.float32)
        self.lib.pto_sample_env(env.ctypes.data_as(C.c_void_p), env.shape[1], 0 if env.dtype == np.float32 else 1,
                                _ptr(d), _ptr(out))
        return out
