"""Scene data model + std140 packing for the GameObjectsUBO blob the integrator consumes.

Host-side mirror (harness glue, not the product) of the reference's scene classes:
  Material   /root/reference/OpenTK-PathTracer/src/Material.cs:7-61
  Sphere     /root/reference/OpenTK-PathTracer/src/GameObjects/Sphere.cs:6-51
  Cuboid     /root/reference/OpenTK-PathTracer/src/GameObjects/Cuboid.cs:6-53
  LoadScene  /root/reference/OpenTK-PathTracer/src/MainWindow.cs:208-267
GLSL view of the same bytes: res/shaders/PathTracing/compute.glsl:13-42,66-70.

Layout (std140, bytes):  Material 64 = Albedo.xyz@0 SpecularChance@12 | Emissiv.xyz@16 SpecularRoughness@28 |
Absorbance.xyz@32 RefractionChance@44 | RefractionRoughness@48 IOR@52 pad@56..63.
Sphere 80 = Position.xyz@0 Radius@12 Material@16.  Cuboid 96 = Min.xyz@0 pad Max.xyz@16 pad Material@32.
GameObjectsUBO = Spheres[256]@0 (stride 80), Cuboids[64]@20480 (stride 96); 26,624 B total (MainWindow.cs:17,199-201).

All arithmetic that produces scene literals is done in float32, like the C# host (SSE float math on .NET 5).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

F = np.float32

MAX_GAMEOBJECTS_SPHERES = 256  # MainWindow.cs:17
MAX_GAMEOBJECTS_CUBOIDS = 64
MATERIAL_SIZE = 64
SPHERE_SIZE = 16 + MATERIAL_SIZE  # Sphere.cs:8
CUBOID_SIZE = 32 + MATERIAL_SIZE  # Cuboid.cs:8
CUBOIDS_OFFSET = SPHERE_SIZE * MAX_GAMEOBJECTS_SPHERES  # Cuboid.cs:21
GAME_OBJECTS_UBO_SIZE = CUBOIDS_OFFSET + CUBOID_SIZE * MAX_GAMEOBJECTS_CUBOIDS  # 26,624
HOST_EPSILON = F(0.005)  # MainWindow.cs:18


def vec3(x, y=None, z=None) -> np.ndarray:
    if y is None:
        return np.array([x, x, x], dtype=F)
    return np.array([x, y, z], dtype=F)


@dataclass
class Material:
    """Material.cs:19-31 — constructor clamps included."""

    albedo: np.ndarray = field(default_factory=lambda: vec3(1.0))
    emissiv: np.ndarray = field(default_factory=lambda: vec3(0.0))
    absorbance: np.ndarray = field(default_factory=lambda: vec3(0.0))  # "refractionColor" ctor arg
    specular_chance: float = 0.0
    specular_roughness: float = 0.0
    ior: float = 1.0
    refraction_chance: float = 0.0
    refraction_roughness: float = 0.0

    def __post_init__(self):
        self.albedo = np.asarray(self.albedo, dtype=F)
        self.emissiv = np.asarray(self.emissiv, dtype=F)
        self.absorbance = np.asarray(self.absorbance, dtype=F)
        self.specular_chance = F(min(max(F(self.specular_chance), F(0.0)), F(1.0)))
        self.specular_roughness = F(self.specular_roughness)
        self.ior = F(max(F(self.ior), F(1.0)))
        self.refraction_chance = F(min(max(F(self.refraction_chance), F(0.0)), F(1.0) - self.specular_chance))
        self.refraction_roughness = F(self.refraction_roughness)

    @staticmethod
    def zero() -> "Material":  # Material.cs:9
        return Material()

    def gpu_data(self) -> np.ndarray:
        """Material.cs:35-51 (GetGPUFriendlyData) -> 16 floats."""
        d = np.zeros(16, dtype=F)
        d[0:3] = self.albedo
        d[3] = self.specular_chance
        d[4:7] = self.emissiv
        d[7] = self.specular_roughness
        d[8:11] = self.absorbance
        d[11] = self.refraction_chance
        d[12] = self.refraction_roughness
        d[13] = self.ior
        return d


@dataclass
class Sphere:
    position: np.ndarray
    radius: float
    instance: int
    material: Material

    @property
    def buffer_offset(self) -> int:  # Sphere.cs:20
        return self.instance * SPHERE_SIZE

    def gpu_data(self) -> np.ndarray:  # Sphere.cs:23-31
        d = np.zeros(SPHERE_SIZE // 4, dtype=F)
        d[0:3] = np.asarray(self.position, dtype=F)
        d[3] = F(self.radius)
        d[4:] = self.material.gpu_data()
        return d


@dataclass
class Cuboid:
    position: np.ndarray
    dimensions: np.ndarray
    instance: int
    material: Material

    @property
    def buffer_offset(self) -> int:  # Cuboid.cs:21
        return CUBOIDS_OFFSET + self.instance * CUBOID_SIZE

    @property
    def min(self) -> np.ndarray:  # Cuboid.cs:23
        return (np.asarray(self.position, dtype=F) - np.asarray(self.dimensions, dtype=F) * F(0.5)).astype(F)

    @property
    def max(self) -> np.ndarray:  # Cuboid.cs:24
        return (np.asarray(self.position, dtype=F) + np.asarray(self.dimensions, dtype=F) * F(0.5)).astype(F)

    def gpu_data(self) -> np.ndarray:  # Cuboid.cs:27-35
        d = np.zeros(CUBOID_SIZE // 4, dtype=F)
        d[0:3] = self.min
        d[4:7] = self.max
        d[8:] = self.material.gpu_data()
        return d


@dataclass
class Scene:
    spheres: list = field(default_factory=list)
    cuboids: list = field(default_factory=list)

    @property
    def num_spheres(self) -> int:
        return len(self.spheres)

    @property
    def num_cuboids(self) -> int:
        return len(self.cuboids)

    def objects(self):
        return list(self.spheres) + list(self.cuboids)

    def ubo_bytes(self) -> bytes:
        """The full 26,624-byte GameObjectsUBO image after every object's Upload()
        (BaseSTD140Compatible.cs:12-16 -> BufferObject.SubData)."""
        blob = np.zeros(GAME_OBJECTS_UBO_SIZE // 4, dtype=F)
        for o in self.objects():
            off = o.buffer_offset // 4
            d = o.gpu_data()
            blob[off:off + d.size] = d
        return blob.tobytes()


def default_cuboids() -> list:
    """MainWindow.cs:249-262 — the 7 room cuboids (down, upLight, back, front, right, left, middle)."""
    width, height, depth = F(40.0), F(25.0), F(25.0)
    EPS = HOST_EPSILON
    cub = []

    def add(pos, dim, mat):
        cub.append(Cuboid(np.asarray(pos, dtype=F), np.asarray(dim, dtype=F), len(cub), mat))
        return cub[-1]

    down = add(vec3(0.0, -height / F(2.0), -10.0), vec3(width, EPS, depth),
               Material(albedo=vec3(0.2, 0.04, 0.04), specular_roughness=0.051))
    dpos, ddim = down.position, down.dimensions
    add(vec3(0.0, F(18.495) - EPS, -4.0), vec3(ddim[0] * F(0.3), EPS, ddim[2] * F(0.3)),
        Material(albedo=vec3(0.04), emissiv=(vec3(0.917, 0.945, 0.513) * F(5.0)).astype(F)))
    add(vec3(dpos[0], dpos[1] + height / F(2), dpos[2] + depth / F(2) - F(5.0)), vec3(width, height, EPS),
        Material(albedo=vec3(0.37109375, 0.67578125, 0.3359375)))
    add(vec3(dpos[0], dpos[1] + height / F(2) + EPS, dpos[2] - depth / F(2)),
        vec3(width, height - EPS * F(2), F(0.3)),
        Material(albedo=vec3(1.0), absorbance=vec3(0.01), specular_chance=0.04, ior=1.0, refraction_chance=0.954))
    add(vec3(dpos[0] + width / F(2), dpos[1] + height / F(2.0), dpos[2]), vec3(EPS, height, depth),
        Material(albedo=vec3(0.9453125, 0.75390625, 0.3046875), specular_chance=1.0, specular_roughness=0.19))
    add(vec3(dpos[0] - width / F(2), dpos[1] + height / F(2.0), dpos[2]), vec3(EPS, height, depth),
        Material(albedo=vec3(0.074219, 0.25, 0.453125)))
    add(vec3(-15.0, F(-10.5) + EPS, -15.0), vec3(3.0, 6.0, 3.0), Material(albedo=vec3(1.0)))
    return cub


def default_scene() -> Scene:
    """The reference's LoadScene(): 48 spheres + 7 cuboids (MainWindow.cs:208-267)."""
    width, height, depth = F(40.0), F(25.0), F(25.0)
    sc = Scene()
    balls = 6
    fb = F(balls)
    radius = F(1.3)
    dim = np.array([width * F(0.6), height, depth], dtype=F)  # MainWindow.cs:217
    for xi in range(balls):
        for yi in range(balls):
            x, y = F(xi), F(yi)
            pos = vec3(dim[0] / fb * x * F(1.1) - dim[0] / F(2), (dim[1] / fb) * y - dim[1] / F(2) + radius, -5.0)
            sc.spheres.append(Sphere(pos, radius, len(sc.spheres), Material(
                albedo=vec3(0.59, 0.59, 0.99), specular_chance=x / F(balls - 1), specular_roughness=y / F(balls - 1),
                ior=1.0, refraction_chance=0.0, refraction_roughness=0.1)))
    delta = (dim / fb).astype(F)
    for xi in range(balls):
        x = F(xi)
        m = Material.zero()
        m.albedo = vec3(0.9, 0.25, 0.25)
        m.specular_chance = F(0.02)
        m.ior = F(1.05)
        m.refraction_chance = F(0.98)
        m.absorbance = (vec3(1.0, 2.0, 3.0) * (x / fb)).astype(F)
        pos = vec3(-dim[0] / F(2) + radius + delta[0] * x, 3.0, -20.0)
        sc.spheres.append(Sphere(pos, radius, len(sc.spheres), m))
        m1 = Material.zero()
        m1.specular_chance = F(0.02)
        m1.specular_roughness = x / fb
        m1.ior = F(1.1)
        m1.refraction_chance = F(0.98)
        m1.refraction_roughness = x / fb
        m1.absorbance = vec3(0.0)
        pos = vec3(-dim[0] / F(2) + radius + delta[0] * x, -6.0, -20.0)
        sc.spheres.append(Sphere(pos, radius, len(sc.spheres), m1))
    sc.cuboids = default_cuboids()
    return sc


def _pcg(state: int):
    """The integrator's own hash (compute.glsl:334-339), reused here only to jitter stress scenes."""
    state = (state * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return state, ((word >> 22) ^ word) & 0xFFFFFFFF


def stress_scene(num_spheres: int = 256, seed: int = 3) -> Scene:
    """BASELINE config 3: `num_spheres` (fills the UBO at 256) small spheres on a jittered 16x16 lattice inside
    the default room, materials cycling through the 36 grid-sphere materials; same 7 cuboids.  Not a reference
    scene — the reference has only LoadScene(); this exists to exercise the 256-entry traversal."""
    base = default_scene()
    sc = Scene()
    st = seed
    n = int(np.ceil(np.sqrt(num_spheres)))
    for i in range(num_spheres):
        gx, gy = i % n, i // n
        st, r0 = _pcg(st)
        st, r1 = _pcg(st)
        st, r2 = _pcg(st)
        jx, jy, jz = (F(r) / F(4294967296.0) for r in (r0, r1, r2))
        pos = vec3(F(-18.0) + F(36.0) * (F(gx) + F(0.5) * jx) / F(n),
                   F(-11.5) + F(22.0) * (F(gy) + F(0.5) * jy) / F(n),
                   F(-19.0) + F(16.0) * jz)
        sc.spheres.append(Sphere(pos, F(0.6), i, base.spheres[i % 36].material))
    sc.cuboids = default_cuboids()
    return sc


def glass_scene() -> Scene:
    """BASELINE config 5: the default scene with all 48 spheres turned into clear/absorbing glass
    (closest reference precedent: MainWindow.cs:223-244)."""
    sc = default_scene()
    for i, s in enumerate(sc.spheres):
        s.material = Material(albedo=vec3(1.0), absorbance=(vec3(1.0, 2.0, 3.0) * (F(i % 6) / F(6))).astype(F),
                              specular_chance=0.02, ior=1.5, refraction_chance=0.98, refraction_roughness=0.0)
    return sc


def random_material_scene(seed: int = 7) -> Scene:
    """Default geometry with randomised materials on the 36 grid spheres — the analogue of the GUI's
    "SpheresRandomMaterial" button (Gui.cs:68-72 -> Material.GetRndMaterial, Material.cs:54-58); exercises
    emissive spheres, IOR>1 with roughness 0 (total internal reflection -> NaN direction) and absorbance."""
    rng = np.random.RandomState(seed)
    sc = default_scene()
    for s in sc.spheres[:36]:
        emissive = rng.rand() < 0.2
        s.material = Material(albedo=rng.rand(3), emissiv=rng.rand(3) if emissive else vec3(0.0),
                              absorbance=rng.rand(3) * 2.0, specular_chance=rng.rand() * 0.5,
                              specular_roughness=rng.rand(), ior=rng.rand() + 1.0,
                              refraction_chance=rng.rand() * 0.5, refraction_roughness=rng.rand())
    return sc
