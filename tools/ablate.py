"""Timing ablations by scene construction (development helper)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
S = pkg.scene
cam = pkg.camera.Camera()
W, H = 1920, 1080
sky = pkg.envmap.synthetic_sky_rgba32f(64)
basic = pkg.camera.basic_data_ubo(cam, W, H)
def timeit(name, sc, depth=8, variant=0):
    pt = pkg.PathTracer(sky, W, H, depth, 1, 20.0, 0.14)
    pt.SetVariant(variant); pt.UploadScene(sc); pt.UploadBasicData(basic)
    for _ in range(10): pt.Render()
    pt.Synchronize(); n = 100; ms = 1e9
    for rep in range(3):
        pt.TimerBegin()
        for _ in range(n): pt.Render()
        ms = min(ms, pt.TimerEnd() / n)
    print(f"{name:44s} v{variant} {ms:.4f} ms", flush=True)
    pt.Dispose()
    return ms
room = S.Scene(); room.cuboids = S.default_cuboids()
far = S.Scene(); far.cuboids = S.default_cuboids()
for i in range(48): far.spheres.append(S.Sphere(S.vec3(1000+3*i, 1000, 1000), 1.0, i, S.Material()))
behind = S.Scene(); behind.cuboids = S.default_cuboids()
for i in range(48): behind.spheres.append(S.Sphere(S.vec3(-17.14 + 0.01*i, 3.53, -8.62), 60.0, i, S.Material()))  # camera inside all: always candidates
empty = S.Scene()
diffuse = S.Scene(); diffuse.cuboids = S.default_cuboids()
for c in diffuse.cuboids: c.material = S.Material(albedo=c.material.albedo, emissiv=c.material.emissiv)
one = S.Scene(); one.cuboids = [S.Cuboid(S.vec3(0, 0, -10), S.vec3(40, 25, 25), 0, S.Material(albedo=S.vec3(0.7), emissiv=S.vec3(0.1)))]
for v in (0,):
    timeit("empty scene (env only)", empty, variant=v)
    timeit("one big diffuse box (1 cuboid, inside)", one, variant=v)
    timeit("room, all-diffuse materials", diffuse, variant=v)
    timeit("room only (7 cuboids)", room, variant=v)
    timeit("room + 48 far spheres (disc loop only)", far, variant=v)
    timeit("default scene", S.default_scene(), variant=v)
    timeit("default scene depth 1", S.default_scene(), depth=1, variant=v)
    timeit("default scene depth 2", S.default_scene(), depth=2, variant=v)
    timeit("default scene depth 4", S.default_scene(), depth=4, variant=v)
