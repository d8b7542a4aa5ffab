"""Time the REFERENCE'S OWN CODE on the CPU: res/shaders/PathTracing/compute.glsl, unmodified, on Mesa llvmpipe
(SURVEY.md section 8d / BASELINE.md section 3.2 — the only way the reference runs without a GPU; its C# host cannot be
built here).  Build-container only (needs /root/reference + swrast_dri.so).  Writes profiles/reference_llvmpipe.json,
which bench.py echoes as cpu_baseline.reference_glsl_llvmpipe next to the oracle-port figure it measures live.

    python tools/time_reference_llvmpipe.py
"""
import importlib.util
import json
import os
import platform
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import configs  # noqa: E402

spec = importlib.util.spec_from_file_location("glsl_run", os.path.join(ROOT, "oracle", "glsl_ref", "run.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
if not ref.available():
    sys.exit("needs oracle/_ref/glsl_runner (make -C oracle ref), /root/reference and Mesa llvmpipe")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def time_one(w, threads, frames):
    sc, basic, objs, env, kw = configs.inputs(w)
    out, log = ref.run_pathtracer(w.width, w.height, basic, objs, env, num_frames=frames, return_log=True, threads=threads, **kw)
    m = re.search(r"([\d.]+) ms/frame, ([\d.]+) Msamples/s", log)
    return {"ms_per_frame": float(m.group(1)), "msamples_per_s": float(m.group(2)), "frames": frames, "threads": threads,
            "mean_radiance": float(out[0, ..., :3].mean())}


cores = os.cpu_count()
result = {
    "what": "the reference's unmodified PathTracing/compute.glsl executed by Mesa llvmpipe (GL 4.5 compute on the CPU), "
            "timed by oracle/glsl_ref/glsl_runner.c around glDispatchCompute + glFinish, shader compile excluded",
    "provenance": {"where": "build container (no GPU)", "cpu": cpu_model(), "logical_cores": cores,
                   "mesa": "llvmpipe (swrast_dri.so)", "tool": "tools/time_reference_llvmpipe.py"},
    "runs": {},
}
for name, w, frames in (("C1_default_512_d4", configs.C1, 8), ("C2_default_1080p_d8", configs.C2, 3),
                        ("C2_default_1080p_d8_atmo", configs.C2_ATMO, 3)):
    result["runs"][name] = {"all_threads": time_one(w, cores, frames), "one_thread": time_one(w, 1, max(1, frames // 3))}
    print(name, json.dumps(result["runs"][name]))
json.dump(result, open(os.path.join(ROOT, "profiles", "reference_llvmpipe.json"), "w"), indent=1)
