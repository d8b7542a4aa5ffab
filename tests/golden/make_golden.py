"""Generate the committed golden fixtures by running the REFERENCE ITSELF in the build container.

Runs the reference's unmodified GLSL (/root/reference/OpenTK-PathTracer/res/shaders/...) on Mesa llvmpipe through
oracle/_ref/glsl_runner (see oracle/glsl_ref/glsl_runner.c) and stores inputs + outputs as small .npz files next to
this script.  Needs /root/reference and Mesa's swrast_dri.so, so it only works in the build container; the
fixtures it writes are data (input blobs, parameters, expected pixels) and are what travels to the GPU box.

    python tests/golden/make_golden.py            # regenerate everything
    python tests/golden/make_golden.py frames     # only one group: envs | micro | frames | envonly | sparse | atmo

The "micro" group drives individual functions of the reference shader: the reference text is read at run time, its
`void main()` is renamed in memory and a small test `main` (written here) that calls the reference's own functions
is appended; the derived text lives only in a temporary file.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import configs  # noqa: E402

spec = importlib.util.spec_from_file_location("glsl_run", os.path.join(ROOT, "oracle", "glsl_ref", "run.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
pkg = configs.pkg


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def params_array(w, kw):
    return np.array([w.width, w.height, kw["num_spheres"], kw["num_cuboids"], w.ray_depth, w.spp, w.frames], dtype=np.int32), \
        np.array([w.focal_length, w.aperture], dtype=np.float32)


# ------------------------------------------------------------------------------------------------ envs + atmosphere
def gen_envs():
    print("envs:")
    ubo = pkg.camera.atmospheric_data_ubo()
    lp = pkg.camera.atmosphere_light_pos(0.5)
    atmo32 = ref.run_atmosphere(32, ubo, lp, 15.0, 50, 15)
    envs = {k: configs.make_env(k) for k in ("sky_f32_32", "sky_srgb_32", "tiny_2", "tiny_4")}
    envs["atmosphere_32"] = atmo32
    envs["atmosphere_64"] = ref.run_atmosphere(64, ubo, lp, 15.0, 50, 15)  # env of the bench workload's pinned fixture
    save("envs", **envs)


def gen_atmo():
    print("atmosphere (reference AtmosphericScattering/compute.glsl on llvmpipe):")
    ubo = pkg.camera.atmospheric_data_ubo()
    for name, size, t, isteps, jsteps, inten in (("atmo_32_default", 32, 0.5, 50, 15, 15.0),
                                                   ("atmo_48_noon", 48, 0.25, 50, 15, 15.0),
                                                   ("atmo_24_few_steps", 24, 0.52, 7, 3, 22.0)):
        lp = pkg.camera.atmosphere_light_pos(t)
        out = ref.run_atmosphere(size, ubo, lp, inten, isteps, jsteps)
        save(name, ubo=np.frombuffer(ubo, np.uint8), light_pos=lp, params=np.array([size, isteps, jsteps], np.int32),
             intensity=np.float32(inten), expected=out[..., :3].copy())


# ------------------------------------------------------------------------------------------------ frames
def gen_frames():
    print("small full frames:")
    for w in configs.SMALL_FRAMES:
        sc, basic, objs, env, kw = configs.inputs(w)
        out = ref.run_pathtracer(w.width, w.height, basic, objs, env, num_frames=w.frames, dump_each=True, **kw)
        ip, fp = params_array(w, kw)
        keep = np.array(sorted({0, 1, 2, w.frames - 1} & set(range(w.frames))), dtype=np.int32)  # accumulated-after-frame k
        save("frame_" + w.name, basic=np.frombuffer(basic, np.uint8), objects=np.frombuffer(objs, np.uint8),
             env_key=np.array(w.env), iparams=ip, fparams=fp, frame_indices=keep, expected=out[keep][..., :3].copy())


def gen_envonly():
    print("environment-sampler frames (empty scene):")
    for w in configs.ENV_ONLY:
        sc, basic, objs, env, kw = configs.inputs(w)
        out = ref.run_pathtracer(w.width, w.height, basic, objs, env, num_frames=1, **kw)
        ip, fp = params_array(w, kw)
        save("envonly_" + w.name, basic=np.frombuffer(basic, np.uint8), env_key=np.array(w.env), iparams=ip, fparams=fp,
             expected=out[0, ..., :3].copy())


def gen_sparse(only=None):
    print("sparse full-resolution fixtures (4096 seeded pixels of the full frame):")
    for w in configs.FULL_SIZE:
        if only and w.name not in only:
            continue
        sc, basic, objs, env, kw = configs.inputs(w)
        out, log = ref.run_pathtracer(w.width, w.height, basic, objs, env, num_frames=1, return_log=True, **kw)
        print("   ", log.strip())
        rng = np.random.RandomState(1234)
        xy = np.stack([rng.randint(0, w.width, 4096), rng.randint(0, w.height, 4096)], axis=1).astype(np.int32)
        vals = out[0, xy[:, 1], xy[:, 0], :3].copy()
        ip, fp = params_array(w, kw)
        save("sparse_" + w.name, basic=np.frombuffer(basic, np.uint8), objects=np.frombuffer(objs, np.uint8),
             env_key=np.array(w.env), iparams=ip, fparams=fp, xy=xy, expected=vals,
             frame_mean=out[0, ..., :3].mean(axis=(0, 1), dtype=np.float64))


# ------------------------------------------------------------------------------------------------ micro (function-level)
MICRO_MAINS = {
    # 4 successive RNG draws of the per-pixel stream (compute.glsl:106,334-344)
    "rng": """
void main() {
    ivec2 c = ivec2(gl_GlobalInvocationID.xy);
    rndSeed = gl_GlobalInvocationID.x * 1973 + gl_GlobalInvocationID.y * 9277 + thisRendererFrame * 2699 | 1;
    float a = GetRandomFloat01(); float b = GetRandomFloat01(); float d = GetRandomFloat01(); float e = GetRandomFloat01();
    imageStore(ImgResult, c, vec4(a, b, d, e));
}""",
    # raw hash values, bit-cast through float so they survive the RGBA32F image exactly
    "hash": """
void main() {
    ivec2 c = ivec2(gl_GlobalInvocationID.xy);
    uint s = uint(c.x) * 2654435761u + uint(c.y) * 40503u + uint(thisRendererFrame);
    uint h0 = GetPCGHash(s); uint h1 = GetPCGHash(s); uint h2 = GetPCGHash(s);
    imageStore(ImgResult, c, vec4(uintBitsToFloat(h0 >> 8u), uintBitsToFloat(h1 >> 8u), uintBitsToFloat(h2 >> 8u), uintBitsToFloat(s >> 8u)));
}""",
    # sphere 0 of the UBO against a fan of rays from a point that is outside / inside / on the sphere
    "sphere": """
void main() {
    ivec2 c = ivec2(gl_GlobalInvocationID.xy);
    Ray r;
    r.Origin = vec3(0.25 * float(c.y / 16), 0.5, 4.0 - 1.25 * float(c.y % 16) * 0.25);
    r.Direction = normalize(vec3((float(c.x) - 31.5) / 24.0, (float(c.y % 16) - 7.5) / 24.0, -1.0));
    float t1, t2;
    bool hit = RaySphereIntersect(r, gameObjectsUBO.Spheres[0], t1, t2);
    imageStore(ImgResult, c, vec4(t1, t2, hit ? 1.0 : 0.0, GetSmallestPositive(t1, t2)));
}""",
    "cuboid": """
void main() {
    ivec2 c = ivec2(gl_GlobalInvocationID.xy);
    Ray r;
    r.Origin = vec3(0.3 * float(c.y / 16) - 0.5, 0.25, 4.0 - float(c.y % 16) * 0.4);
    r.Direction = normalize(vec3((float(c.x) - 31.5) / 20.0, (float(c.y % 16) - 7.5) / 20.0, -1.0));
    if (c.x == 0) r.Direction = vec3(0.0, 0.0, -1.0);
    if (c.x == 1) r.Direction = vec3(1.0, 0.0, 0.0);
    float t1, t2;
    bool hit = RayCuboidIntersect(r, gameObjectsUBO.Cuboids[0], t1, t2);
    vec3 n = GetNormal(gameObjectsUBO.Cuboids[0], r.Origin + r.Direction * GetSmallestPositive(t1, t2));
    imageStore(ImgResult, c, vec4(t1, t2, hit ? 1.0 : 0.0, n.x + 2.0 * n.y + 4.0 * n.z));
}""",
    # built-ins the integrator relies on, over the argument ranges it uses
    "math": """
void main() {
    ivec2 c = ivec2(gl_GlobalInvocationID.xy);
    float u = (float(c.y * 64 + c.x) + 0.5) / 4096.0;
    float a = u * 2.0 * PI;
    imageStore(ImgResult, c, vec4(sin(a), cos(a), exp(-u * 12.0), pow(u * 2.0, 5.0)));
}""",
    # cosine-weighted hemisphere sample around a pixel-dependent normal (compute.glsl:297-307)
    "cosine": """
void main() {
    ivec2 c = ivec2(gl_GlobalInvocationID.xy);
    rndSeed = gl_GlobalInvocationID.x * 1973 + gl_GlobalInvocationID.y * 9277 + thisRendererFrame * 2699 | 1;
    vec3 n = normalize(vec3(float(c.x) - 31.5, float(c.y) - 31.5, 9.0));
    vec3 d = CosineSampleHemisphere(n);
    imageStore(ImgResult, c, vec4(d, dot(d, n)));
}""",
    # Fresnel + refract incl. total internal reflection (compute.glsl:359-364, GLSL refract)
    "fresnel": """
void main() {
    ivec2 c = ivec2(gl_GlobalInvocationID.xy);
    float cosT = float(c.x) / 63.0;
    float ior = 1.0 + float(c.y) / 63.0;
    vec3 n = vec3(0.0, 1.0, 0.0);
    vec3 i = vec3(sqrt(1.0 - cosT * cosT), -cosT, 0.0);
    vec3 r = refract(i, n, (c.y % 2 == 0) ? ior : 1.0 / ior);
    imageStore(ImgResult, c, vec4(FresnelSchlick(cosT, 1.0, ior), r));
}""",
}


def run_micro(name, frame=5, size=64):
    src = open(ref.PT_SHADER, "rb").read().decode("utf-8-sig")
    assert "void main()" in src
    derived = src.replace("void main()", "void reference_main_unused()", 1) + "\n" + MICRO_MAINS[name] + "\n"
    sc = pkg.scene.Scene()
    sc.spheres.append(pkg.scene.Sphere(pkg.scene.vec3(0.25, 0.5, -1.0), 1.5, 0, pkg.scene.Material()))
    sc.cuboids.append(pkg.scene.Cuboid(pkg.scene.vec3(0.0, 0.25, -2.0), pkg.scene.vec3(2.0, 1.5, 1.0), 0, pkg.scene.Material()))
    basic = pkg.camera.basic_data_ubo(pkg.camera.Camera(), size, size)
    env = configs.make_env("tiny_2")
    with tempfile.NamedTemporaryFile("w", suffix=".glsl", delete=False) as f:
        f.write(derived)
        path = f.name
    old = ref.PT_SHADER
    try:
        ref.PT_SHADER = path
        out = ref.run_pathtracer(size, size, basic, sc.ubo_bytes(), env, num_spheres=1, num_cuboids=1, ray_depth=1,
                                 frame_start=frame, num_frames=1)
    finally:
        ref.PT_SHADER = old
        os.unlink(path)
    return out[0], sc


def gen_micro():
    print("micro (reference functions driven through a test main):")
    arrays = {}
    for name in MICRO_MAINS:
        out, sc = run_micro(name)
        arrays[name] = out
    arrays["frame"] = np.int32(5)
    arrays["objects"] = np.frombuffer(sc.ubo_bytes(), np.uint8)
    save("micro", **arrays)


def gen_post():
    print("post-process (reference PostProcessing/fragment.glsl functions on llvmpipe):")
    rng = np.random.RandomState(11)
    img = np.ones((96, 64, 4), dtype=np.float32)
    img[..., :3] = np.exp(rng.uniform(np.log(1e-6), np.log(80.0), (96, 64, 3))).astype(np.float32)
    ramp = np.linspace(0.0, 4.0, 64, dtype=np.float32)
    img[0, :, :3] = ramp[:, None]                       # grey ramp through the knee
    img[1, :, :3] = (ramp * np.float32(0.002))[:, None]  # around the 0.0031308 linear/power switch
    img[2, :8, :3] = [[0, 0, 0], [1, 1, 1], [0.0031308, 0.0031307, 0.0031309], [-0.5, -1e-3, 2.0], [1e-8, 1e-7, 1e-6],
                      [0.18, 0.18, 0.18], [100, 1000, 1e6], [0.5, 0.25, 0.75]]
    # plus a real HDR frame of the integrator (default scene) so the fixture covers the value distribution that matters
    w = configs.SMALL_FRAMES[1]
    sc, basic, objs, env, kw = configs.inputs(w)
    hdr = ref.run_pathtracer(w.width, w.height, basic, objs, env, num_frames=4, **kw)[0]
    img[24:96, :, :] = hdr[:, 32:96, :]
    out = ref.run_postprocess(img)
    save("post_aces_gamma", image=img, expected=out[..., :3].copy())


def gen_converged():
    print("converged accumulation (statistical parity: the branch-flip pixels average out):")
    w = configs.Workload("default_96x54_d8_acc96", "default", 96, 54, 8, "sky_f32_32", frames=96)
    sc, basic, objs, env, kw = configs.inputs(w)
    out = ref.run_pathtracer(w.width, w.height, basic, objs, env, num_frames=w.frames, **kw)
    ip, fp = params_array(w, kw)
    save("converged_" + w.name, basic=np.frombuffer(basic, np.uint8), objects=np.frombuffer(objs, np.uint8),
         env_key=np.array(w.env), iparams=ip, fparams=fp, expected=out[0, ..., :3].copy())


def gen_convergence():
    """Per-pixel convergence statistics of the REFERENCE (its own GLSL on llvmpipe) for the three scene families: 4,096 frames of
    a 64x36 image rendered as 64 independent blocks of 64 consecutive frame indices (block b = frames 64b .. 64b+63; the
    shader's running mean over a block that starts at frame index f0 on a zeroed image ends at sum / (f0 + 64), so the block
    mean is that value * (f0 + 64) / 64).  Stored per pixel and channel: the mean over all 4,096 frames and its standard error
    (std of the 64 block means / 8).  Pixels the reference itself turns NaN (normalize(0) paths) are kept as NaN."""
    print("per-pixel convergence statistics (reference GLSL, 64 blocks x 64 frames):")
    blocks, per = 64, 64
    for w in configs.CONVERGENCE:
        sc, basic, objs, env, kw = configs.inputs(w)
        means = np.empty((blocks, w.height, w.width, 3), np.float64)
        for b in range(blocks):
            f0 = b * per
            out = ref.run_pathtracer(w.width, w.height, basic, objs, env, frame_start=f0, num_frames=per, **kw)
            means[b] = out[0, ..., :3].astype(np.float64) * ((f0 + per) / per)
        mean = means.mean(axis=0)
        se = means.std(axis=0, ddof=1) / np.sqrt(blocks)
        ip, fp = params_array(w, kw)
        ip[6] = blocks * per
        print(f"   {w.name}: mean radiance {np.nanmean(mean):.5f}, median relative standard error {np.nanmedian(se / np.maximum(mean, 1e-6)):.4f}, "
              f"NaN pixels {int(np.isnan(mean).any(-1).sum())}")
        save("convergence_" + w.name, basic=np.frombuffer(basic, np.uint8), objects=np.frombuffer(objs, np.uint8),
             env_key=np.array(w.env), iparams=ip, fparams=fp, mean=mean.astype(np.float32), stderr=se.astype(np.float32))


def gen_edge_nanenv():
    """The quirk scene's per-SAMPLE record: for 64 frames of the `edge` scene (total internal reflection ->
    normalize(vec3(0)) = NaN direction -> texture(SamplerEnvironment, NaN), undefined in GL) the reference's running means after
    every frame, and — from the oracle's diagnostic build (oracle/Makefile: nanmark) — which (frame, pixel) samples END in such a
    NaN-direction environment lookup.  tests/test_live_reference.py makes the same comparison against the live llvmpipe run; this
    fixture lets the GPU box make it (the HIP path equals the oracle bit for bit, so the oracle's flags are the HIP path's)."""
    print("edge scene: per-frame reference accumulation + NaN-direction lookup flags:")
    import __graft_entry__ as graft
    O = graft.load_oracle()
    plain, marked = O.Oracle(), O.Oracle(mark_nan_env=True)
    w = configs.Workload("edge_64x36_d16", "edge", 64, 36, 16, "sky_f32_32", frames=64)
    sc, basic, objs, env, kw = configs.inputs(w)
    acc_ref = ref.run_pathtracer(w.width, w.height, basic, objs, env, num_frames=w.frames, dump_each=True, **kw)[..., :3]
    acc = [x.render(w.width, w.height, basic, objs, env, num_frames=w.frames, dump_each=True, **kw)[..., :3].astype(np.float64)
           for x in (plain, marked)]

    def samples(a):
        s_ = np.empty_like(a)
        s_[0] = a[0]
        for k in range(1, len(a)):
            s_[k] = (k + 1) * a[k] - k * a[k - 1]
        return s_
    nan_env = np.abs(samples(acc[1]) - samples(acc[0])).max(-1) > 10.0
    ip, fp = params_array(w, kw)
    print(f"   {int(nan_env.sum())} of {nan_env.size} samples end in a NaN-direction lookup")
    save("edge_nanenv_" + w.name, basic=np.frombuffer(basic, np.uint8), objects=np.frombuffer(objs, np.uint8), env_key=np.array(w.env),
         iparams=ip, fparams=fp, expected_each=acc_ref.astype(np.float32), nan_env=np.packbits(nan_env.reshape(-1)))


def gen_bench_fixture():
    """Only what round 2 added (keeps every older fixture byte-identical): the atmosphere_64 cube + the sparse fixture of
    the exact workload bench.py times."""
    p = os.path.join(HERE, "envs.npz")
    envs = dict(np.load(p))
    if "atmosphere_64" not in envs:
        envs["atmosphere_64"] = ref.run_atmosphere(64, pkg.camera.atmospheric_data_ubo(), pkg.camera.atmosphere_light_pos(0.5), 15.0, 50, 15)
        save("envs", **envs)
        configs._envs = None
    gen_sparse(only={configs.C2_ATMO.name})


GROUPS = {"edge": gen_edge_nanenv, "bench": gen_bench_fixture, "converged": gen_converged, "convergence": gen_convergence, "post": gen_post, "envs": gen_envs, "micro": gen_micro, "frames": gen_frames, "envonly": gen_envonly, "sparse": gen_sparse,
          "atmo": gen_atmo}

if __name__ == "__main__":
    if not ref.available():
        sys.exit("make_golden.py needs oracle/_ref/glsl_runner (make -C oracle ref), /root/reference and Mesa llvmpipe")
    todo = sys.argv[1:] or list(GROUPS)
    for g in todo:
        GROUPS[g]()
