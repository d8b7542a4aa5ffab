"""Python driver for oracle/_ref/glsl_runner (the reference's own GLSL executed on Mesa llvmpipe).

TEST INFRASTRUCTURE, build-container only: it needs /root/reference (shader text) and Mesa's swrast_dri.so,
neither of which exists on the GPU box.  Used by tests/golden/make_golden.py to produce the committed
fixtures and by the container-only pinning tests.
"""
from __future__ import annotations

import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
RUNNER = os.path.join(HERE, "..", "_ref", "glsl_runner")
REF_ROOT = os.environ.get("PT_REFERENCE_ROOT", "/root/reference")
PT_SHADER = os.path.join(REF_ROOT, "OpenTK-PathTracer/res/shaders/PathTracing/compute.glsl")
ATMO_SHADER = os.path.join(REF_ROOT, "OpenTK-PathTracer/res/shaders/AtmosphericScattering/compute.glsl")
POST_SHADER = os.path.join(REF_ROOT, "OpenTK-PathTracer/res/shaders/PostProcessing/fragment.glsl")
SWRAST = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so"


def available() -> bool:
    return os.path.exists(RUNNER) and os.path.exists(PT_SHADER) and os.path.exists(SWRAST)


def _run(shader: str, job: bytes, out_floats: int, threads: int | None = None) -> tuple[np.ndarray, str]:
    with tempfile.TemporaryDirectory() as td:
        jp, op = os.path.join(td, "job.bin"), os.path.join(td, "out.bin")
        with open(jp, "wb") as f:
            f.write(job)
        env = dict(os.environ)
        if threads is not None:
            env["LP_NUM_THREADS"] = str(threads)
        p = subprocess.run([RUNNER, shader, jp, op], env=env, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"glsl_runner failed ({p.returncode}): {p.stderr}")
        out = np.fromfile(op, dtype=np.float32)
        if out.size != out_floats:
            raise RuntimeError(f"glsl_runner wrote {out.size} floats, expected {out_floats}")
        return out, p.stderr


def run_pathtracer(width, height, basic_ubo: bytes, objects_ubo: bytes, env_faces: np.ndarray, *, num_spheres,
                   num_cuboids, ray_depth, spp=1, focal_length=20.0, aperture=0.14, frame_start=0, num_frames=1,
                   dump_each=False, threads=None, return_log=False):
    """Returns (frames, H, W, 4) float32 (frames = num_frames if dump_each else 1); row 0 = image y 0."""
    assert len(basic_ubo) == 144 and len(objects_ubo) == 26624
    env_faces = np.ascontiguousarray(env_faces)
    assert env_faces.shape[0] == 6 and env_faces.shape[1] == env_faces.shape[2] and env_faces.shape[3] == 4
    fmt = 0 if env_faces.dtype == np.float32 else 1
    assert fmt == 0 or env_faces.dtype == np.uint8
    job = struct.pack("<ii", 0x4A4C5347, 0)
    job += struct.pack("<9i", width, height, ray_depth, spp, frame_start, num_frames, int(dump_each),
                       env_faces.shape[1], fmt)
    job += struct.pack("<4f", float(num_spheres), float(num_cuboids), focal_length, aperture)
    job += basic_ubo + objects_ubo + env_faces.tobytes()
    nf = num_frames if dump_each else 1
    out, log = _run(PT_SHADER, job, nf * width * height * 4, threads)
    out = out.reshape(nf, height, width, 4)
    return (out, log) if return_log else out


def run_atmosphere(size, atmo_ubo: bytes, light_pos, light_intensity=15.0, i_steps=50, j_steps=15, threads=None):
    """Returns (6, size, size, 4) float32."""
    assert len(atmo_ubo) == 464
    job = struct.pack("<ii", 0x4A4C5347, 1) + struct.pack("<3i", size, i_steps, j_steps)
    job += struct.pack("<4f", float(light_pos[0]), float(light_pos[1]), float(light_pos[2]), light_intensity)
    job += atmo_ubo
    out, _ = _run(ATMO_SHADER, job, 6 * size * size * 4, threads)
    return out.reshape(6, size, size, 4)


def run_image_transform(shader_path: str, image: np.ndarray) -> np.ndarray:
    """Run a compute shader (8x8 groups) that transforms the RGBA32F image at image unit 0 in place."""
    image = np.ascontiguousarray(image, dtype=np.float32)
    h, w, c = image.shape
    assert c == 4
    job = struct.pack("<ii", 0x4A4C5347, 2) + struct.pack("<2i", w, h) + image.tobytes()
    out, _ = _run(shader_path, job, w * h * 4)
    return out.reshape(h, w, 4)


def run_postprocess(image: np.ndarray) -> np.ndarray:
    """The reference's post-process FUNCTIONS (ACESFilm, LinearToInverseGamma; PostProcessing/fragment.glsl:28-43) applied
    to `image` exactly as fragment.glsl's main does (:17-26), driven from a compute-stage test main: the function
    definitions are cut from the reference text at run time (nothing is stored in the repo)."""
    src = open(POST_SHADER, "rb").read().decode("utf-8-sig")
    start = src.index("vec3 LinearToInverseGamma(vec3 rgb, float gamma)\n{") if "vec3 LinearToInverseGamma(vec3 rgb, float gamma)\n{" in src \
        else src.index("vec3 LinearToInverseGamma(vec3 rgb, float gamma)\r\n{")
    funcs = src[start:]
    derived = ("#version 450 core\nlayout(local_size_x = 8, local_size_y = 8, local_size_z = 1) in;\n"
               "layout(binding = 0, rgba32f) restrict uniform image2D ImgResult;\n"
               "vec3 LinearToInverseGamma(vec3 rgb, float gamma);\nvec3 ACESFilm(vec3 x);\n"
               "void main() {\n  ivec2 c = ivec2(gl_GlobalInvocationID.xy);\n"
               "  if (c.x >= imageSize(ImgResult).x || c.y >= imageSize(ImgResult).y) return;\n"
               "  vec3 color = imageLoad(ImgResult, c).rgb;\n  color = ACESFilm(color);\n"
               "  color = LinearToInverseGamma(color, 2.4);\n  imageStore(ImgResult, c, vec4(color, 1.0));\n}\n" + funcs)
    with tempfile.NamedTemporaryFile("w", suffix=".glsl", delete=False) as f:
        f.write(derived)
        path = f.name
    try:
        return run_image_transform(path, image)
    finally:
        os.unlink(path)
