"""Named workloads shared by the fixture generator (tests/golden/make_golden.py), the parity tests and bench.py.

C1..C5 are BASELINE.json's configs; the others are small parity cases.
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
GOLDEN = os.path.join(ROOT, "tests", "golden")


@dataclass
class Workload:
    name: str
    scene: str          # default | stress256 | glass | randmat | empty | edge
    width: int
    height: int
    ray_depth: int
    env: str            # key into envs.npz / make_env
    spp: int = 1
    frames: int = 1
    focal_length: float = 20.0
    aperture: float = 0.14
    look: tuple = (-32.2, 0.8)      # camera yaw, pitch in degrees (MainWindow.cs:36)
    position: tuple = (-17.14, 3.53, -8.62)


def make_scene(kind: str):
    s = pkg.scene
    if kind == "default":
        return s.default_scene()
    if kind == "stress256":
        return s.stress_scene(256)
    if kind == "glass":
        return s.glass_scene()
    if kind == "randmat":
        return s.random_material_scene()
    if kind == "empty":
        return s.Scene()
    if kind == "edge":
        return edge_case_scene()
    raise KeyError(kind)


def edge_case_scene():
    """Quirk coverage (hard part 4): a glass cuboid with IOR 1.5 and zero roughness (total
    internal reflection -> refract() returns 0 -> normalize(0) = NaN direction, compute.glsl:210-211), a big sphere
    that CONTAINS the camera (entry-distance acceptance quirk, compute.glsl:234), an emissive sphere, a mirror
    sphere touching a cuboid edge, on top of the default room."""
    s = pkg.scene
    sc = s.Scene()
    sc.cuboids = s.default_cuboids()
    sc.cuboids[6].material = s.Material(albedo=s.vec3(1.0), absorbance=s.vec3(0.05, 0.1, 0.2), specular_chance=0.05,
                                        ior=1.5, refraction_chance=0.9, refraction_roughness=0.0)
    mats = [
        s.Material(albedo=s.vec3(0.9), specular_chance=0.02, ior=1.3, refraction_chance=0.97),            # camera inside
        s.Material(albedo=s.vec3(0.1), emissiv=s.vec3(4.0, 2.0, 1.0)),                                       # emissive
        s.Material(albedo=s.vec3(0.95), specular_chance=1.0, specular_roughness=0.0),                       # mirror
        s.Material(albedo=s.vec3(0.8, 0.8, 0.2), specular_chance=0.3, specular_roughness=0.5, ior=1.4,
                   refraction_chance=0.5, refraction_roughness=0.3, absorbance=s.vec3(0.3, 0.1, 0.0)),
    ]
    geo = [((-17.14, 3.53, -8.62), 2.5), ((-5.0, -8.0, -12.0), 1.5), ((-13.5, -11.2, -13.5), 1.3), ((-8.0, 0.0, -14.0), 2.0)]
    for i, ((x, y, z), r) in enumerate(geo):
        sc.spheres.append(s.Sphere(s.vec3(x, y, z), r, i, mats[i]))
    return sc


def make_env(key: str) -> np.ndarray:
    """Environment cubes by key; fixture-sized ones are ALSO stored in tests/golden/envs.npz (see load_env)."""
    e = pkg.envmap
    if key.startswith("sky_f32_"):
        return e.synthetic_sky_rgba32f(int(key.split("_")[-1]))
    if key.startswith("sky_srgb_"):
        return e.synthetic_sky_srgb8(int(key.split("_")[-1]))
    if key.startswith("tiny_"):
        return e.tiny_test_cube(int(key.split("_")[-1]))
    raise KeyError(key)


_envs = None


def load_env(key: str) -> np.ndarray:
    """Committed environment data when the key is in envs.npz (bit-stable across numpy versions), else generated."""
    global _envs
    if _envs is None:
        p = os.path.join(GOLDEN, "envs.npz")
        _envs = dict(np.load(p)) if os.path.exists(p) else {}
    if key in _envs:
        return _envs[key]
    return make_env(key)


def inputs(w: Workload):
    """-> (scene, basic_ubo bytes, objects_ubo bytes, env faces, kwargs for oracle/reference runners)"""
    sc = make_scene(w.scene)
    cam = pkg.camera.Camera(position=w.position, look_x=w.look[0], look_y=w.look[1])
    basic = pkg.camera.basic_data_ubo(cam, w.width, w.height)
    env = load_env(w.env)
    kw = dict(num_spheres=sc.num_spheres, num_cuboids=sc.num_cuboids, ray_depth=w.ray_depth, spp=w.spp,
              focal_length=w.focal_length, aperture=w.aperture)
    return sc, basic, sc.ubo_bytes(), env, kw


# ---- BASELINE.json configs (full size) -------------------------------------------------------------------------
C1 = Workload("C1_default_512_d4", "default", 512, 512, 4, "sky_f32_32")
C2 = Workload("C2_default_1080p_d8", "default", 1920, 1080, 8, "sky_f32_32")
C3 = Workload("C3_stress256_1080p_d8", "stress256", 1920, 1080, 8, "sky_f32_32")
C4 = Workload("C4_default_4k_d8", "default", 3840, 2160, 8, "sky_f32_32")
C5 = Workload("C5_glass_1080p_d32", "glass", 1920, 1080, 32, "atmosphere_32")
# the workload bench.py times at N = 1 (default scene + the reference's default ATMOSPHERE environment, MainWindow.cs:174-175,189),
# with the cube computed by the reference's own AtmosphericScattering/compute.glsl (64^2 keeps the committed cube small)
C2_ATMO = Workload("C2_default_1080p_d8_atmo", "default", 1920, 1080, 8, "atmosphere_64")
FULL_SIZE = [C1, C2, C3, C5, C2_ATMO, C4]

# ---- per-pixel convergence statistics of the reference (4,096 frames of a small image; tests/golden/make_golden.py convergence)
CONVERGENCE = [
    Workload("default_64x36_d8", "default", 64, 36, 8, "sky_f32_32"),
    Workload("glass_64x36_d32_atmo", "glass", 64, 36, 32, "atmosphere_32"),
    Workload("stress256_64x36_d8", "stress256", 64, 36, 8, "sky_f32_32"),
    # the random-material scene (rough / absorbing dielectrics, emitters) at the reference's shipped depth (rayDepth 13, MainWindow.cs:189).
    # (An sRGB-environment variant is no use here: llvmpipe decodes sRGB8 texels with a cubic approximation, and pixels that see only
    # sky have a standard error of ~1e-9, so its deterministic 1e-7 difference reads as |z| in the hundreds.)
    Workload("randmat_64x36_d13", "randmat", 64, 36, 13, "sky_f32_32"),
    # several samples per pixel per frame (one RNG stream per pixel per frame continued across the samples, compute.glsl:106-124), the
    # reference's shipped depth, a wide lens and another camera
    Workload("default_64x36_d13_spp4", "default", 64, 36, 13, "sky_f32_32", spp=4, aperture=0.6, focal_length=12.0, look=(20.0, -10.0),
             position=(5.0, 2.0, -3.0)),
]

# ---- small full-frame parity cases -----------------------------------------------------------------------------
SMALL_FRAMES = [
    Workload("default_64_d4", "default", 64, 64, 4, "sky_f32_32", frames=3),
    Workload("default_128x72_d8", "default", 128, 72, 8, "sky_f32_32", frames=3),
    Workload("default_128x72_d32_acc16", "default", 128, 72, 32, "sky_f32_32", frames=16),
    Workload("default_128x72_d8_srgb", "default", 128, 72, 8, "sky_srgb_32"),
    Workload("default_96x54_d13_spp4", "default", 96, 54, 13, "sky_f32_32", spp=4, frames=2),
    Workload("stress256_128x72_d8", "stress256", 128, 72, 8, "sky_f32_32"),
    Workload("glass_128x72_d32_atmo", "glass", 128, 72, 32, "atmosphere_32"),
    Workload("randmat_128x72_d13", "randmat", 128, 72, 13, "sky_f32_32", frames=2),
    Workload("edge_128x72_d16", "edge", 128, 72, 16, "sky_f32_32", frames=2),
    Workload("default_75x43_d8_odd", "default", 75, 43, 8, "sky_f32_32"),
]

# ---- environment-sampler pinning: empty scene, pinhole camera, distinct texels ------------------------------------
ENV_ONLY = [
    Workload(f"env_{env}_{i}", "empty", 64, 36, 2, env, aperture=0.0, look=look)
    for env in ("tiny_2", "tiny_4", "sky_srgb_32")
    for i, look in enumerate([(-32.2, 0.8), (45.0, 35.26), (135.0, -35.26), (90.0, 89.0), (0.0, -89.0), (-135.0, 35.3)])
]
