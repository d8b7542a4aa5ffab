"""ctypes binding of libmi355pt.so (the C ABI in include/mi355pt.h) + the build recipe.

The library is built IN-TREE (opentk-pathtracer_amd/libmi355pt.so) with hipcc for gfx950; there is no CPU
fallback anywhere in this package: if the shared object is missing or no HIP device is present, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("MI355PT_LIB") or os.path.join(HERE, "libmi355pt.so")  # MI355PT_LIB: A/B tuning builds
HEADER = os.path.join(REPO, "include", "mi355pt.h")
SOURCES = ["pt_integrate_persistent.hip", "pt_integrate_multisample.hip", "pt_helper_kernels.hip", "mi355pt.cpp", "mi355pt_multi.cpp"]
# -ffp-contract=off / -fno-fast-math are part of the pt-f32 arithmetic contract (csrc/pt_math.hpp)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-fvisibility=hidden"]

PT_OK = 0
PT_MAX_IMAGE_DIM, PT_MAX_RAY_DEPTH, PT_MAX_SPP, PT_PRESENT_SLOTS = 32767, 4095, 4095, 3
PT_E_BAD_HANDLE, PT_E_BAD_ARGUMENT, PT_E_OUT_OF_RANGE, PT_E_NO_ENVIRONMENT, PT_E_HIP, PT_E_NO_DEVICE, PT_E_OOM = \
    -1, -2, -3, -4, -5, -6, -7
PT_ENV_RGBA32F, PT_ENV_SRGB8_A8 = 0, 1


class NativeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libmi355pt error {code}: {message}")
        self.code = code


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libmi355pt.so")


def csrc_hash() -> str:
    """Identity of the kernel + host sources and build flags a library is built from (first 12 hex digits of a SHA-1).
    Profiles committed under profiles/ carry it, and bench.py refuses PMC numbers measured on other sources."""
    import hashlib
    h = hashlib.sha1()
    for name in sorted(os.listdir(CSRC)):
        h.update(name.encode())
        h.update(open(os.path.join(CSRC, name), "rb").read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:12]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 ... -> opentk-pathtracer_amd/libmi355pt.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc_path()] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or p.returncode != 0:
        print(" ".join(cmd))
        print(p.stdout + p.stderr)
    if p.returncode != 0:
        raise RuntimeError("hipcc failed building libmi355pt.so")
    return LIB_PATH


# Diagnostic builds of the same sources (never loaded by the product path; tools/handover_stress.cpp and the GPU test that runs it):
#   audit       -DPT_AUDIT            every pixel read-modify-write mirrored by a device-scope atomic side word (csrc/pt_debug_hooks.hpp)
#   chaos       -DPT_CHAOS            pseudo-random s_sleep delays at the hand-over protocol's decision points
#   audit_chaos both
VARIANTS = {"audit": ["-DPT_AUDIT"], "chaos": ["-DPT_CHAOS"], "audit_chaos": ["-DPT_AUDIT", "-DPT_CHAOS"]}
STRESS_BIN = os.path.join(REPO, "tools", "handover_stress.bin")


def variant_path(name: str) -> str:
    return os.path.join(HERE, f"libmi355pt_{name}.so")


def build_variant(name: str, force: bool = False) -> str:
    out = variant_path(name)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [HEADER]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = [hipcc_path()] + HIPCC_FLAGS + VARIANTS[name] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(f"hipcc failed building {out}:\n" + p.stdout + p.stderr)
    return out


def build_stress_tool(force: bool = False) -> str:
    """g++ tools/handover_stress.cpp (dlopens whichever build of the library it is pointed at)."""
    src = os.path.join(REPO, "tools", "handover_stress.cpp")
    if not force and os.path.exists(STRESS_BIN) and os.path.getmtime(src) <= os.path.getmtime(STRESS_BIN):
        return STRESS_BIN
    p = subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", src, "-ldl", "-o", STRESS_BIN], capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("g++ failed building handover_stress:\n" + p.stdout + p.stderr)
    return STRESS_BIN


HOST_DEMO = os.path.join(HERE, "host", "pt_host_demo")


def build_host_demo(force: bool = False) -> str:
    """g++ the C++ host mirror's demo (host/pt_host_demo.cpp) against the in-tree libmi355pt.so."""
    src = [os.path.join(HERE, "host", "pt_host_demo.cpp"), os.path.join(HERE, "host", "pt_host.hpp"), HEADER]
    if not force and os.path.exists(HOST_DEMO) and all(os.path.getmtime(f) <= os.path.getmtime(HOST_DEMO) for f in src):
        return HOST_DEMO
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", src[0], "-o", HOST_DEMO, "-L" + HERE, "-lmi355pt", "-Wl,-rpath,$ORIGIN/.."]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("g++ failed building pt_host_demo:\n" + p.stdout + p.stderr)
    return HOST_DEMO


def declared_symbols() -> list[str]:
    """Every function the public header declares with PT_API."""
    text = open(HEADER).read()
    return sorted(set(re.findall(r"PT_API\s+[\w\s\*]+?\b(pt_\w+)\s*\(", text)))


_lib = None


def _share_torch_hip_runtime() -> None:
    """PyTorch's ROCm wheels bundle their own libamdhip64.  Whichever copy is loaded first serves the whole process:
    if this library pulled in /opt/rocm's copy first, a later `import torch` could not initialise its GPU state
    ("No HIP GPUs are available").  So when torch is installed, its copy is loaded first (no torch import needed);
    the library binds to it through the common SONAME — the order bench.py and the tests always had."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load() -> C.CDLL:
    """dlopen the in-tree library (never builds implicitly on a box without hipcc; never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing — run __graft_entry__.build() (hipcc, gfx950); there is no CPU fallback")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    sig = {
        "pt_create": [C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
        "pt_destroy": [vp],
        "pt_set_size": [vp, C.c_int, C.c_int],
        "pt_set_tile": [vp, C.c_int, C.c_int],
        "pt_set_interleaved_tile": [vp, C.c_int, C.c_int, C.c_int],
        "pt_reset": [vp],
        "pt_set_params": [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float],
        "pt_upload_basic_data": [vp, C.c_int, C.c_int, vp],
        "pt_upload_game_objects": [vp, C.c_int, C.c_int, vp],
        "pt_set_environment": [vp, C.c_int, C.c_int, C.POINTER(vp)],
        "pt_render": [vp, ip],
        "pt_read_result": [vp, fp, C.c_size_t],
        "pt_write_result": [vp, fp, C.c_size_t, C.c_int],
        "pt_present_rgba8": [vp, C.POINTER(C.c_uint8), C.c_size_t],
        "pt_postprocess_device": [vp, C.POINTER(vp), C.POINTER(C.c_size_t)],
        "pt_get_frame_index": [vp, ip],
        "pt_synchronize": [vp],
        "pt_atmosphere_upload_data": [vp, C.c_int, C.c_int, vp],
        "pt_atmosphere_render": [vp, C.c_int, C.c_int, C.c_int, fp, C.c_float],
        "pt_read_environment": [vp, fp, ip],
        "pt_result_device_ptr": [vp, C.POINTER(vp), C.POINTER(C.c_size_t)],
        "pt_bind_result_buffer": [vp, vp, C.c_size_t],
        "pt_set_stream": [vp, vp],
        "pt_timer_begin": [vp],
        "pt_timer_end": [vp, fp],
        "pt_set_variant": [vp, C.c_int],
        "pt_set_frame_batch": [vp, C.c_int],
        "pt_device_count": [],
        "pt_create_multi": [ip, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
        "pt_multi_set_partition": [vp, C.c_int],
        "pt_multi_gather_is_direct": [vp, ip],
        "pt_device_count_of": [vp, ip],
        "pt_present_rgba8_async": [vp, C.c_int],
        "pt_present_wait": [vp, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), ip],
        "pt_present_bind_device_image": [vp, C.c_int, vp, C.c_size_t],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.pt_last_error.argtypes = [vp]
    L.pt_last_error.restype = C.c_char_p
    L.pt_version.argtypes = []
    L.pt_version.restype = C.c_char_p
    _lib = L
    return L


def check(rc: int, handle=None) -> int:
    if rc != PT_OK:
        msg = load().pt_last_error(handle)
        raise NativeError(rc, msg.decode() if msg else "")
    return rc


def debug_set(key: str, value: int) -> None:
    """Tuning knob of the library (csrc/pt_tuning.hpp) through pt_debug_set — an exported entry point that include/mi355pt.h does NOT
    declare.  The product library reads no environment variables; A/B runs, stress tools and a few tests use this instead."""
    L = load()
    L.pt_debug_set.argtypes = [C.c_char_p, C.c_longlong]
    L.pt_debug_set.restype = C.c_int
    rc = L.pt_debug_set(key.encode(), int(value))
    if rc != PT_OK:
        raise NativeError(rc, f"pt_debug_set({key!r}): unknown knob")



def debug_launch_stats(handle) -> dict:
    """How the handle has been launching (pt_debug_launch_stats — exported, not in the public header; neither flushes nor joins)."""
    L = load()
    L.pt_debug_launch_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    L.pt_debug_launch_stats.restype = C.c_int
    out = (C.c_ulonglong * 12)()
    check(L.pt_debug_launch_stats(handle, out), handle)
    return {"launches": int(out[0]), "tile_masks_valid": bool(out[1]), "pending_frames": int(out[2]), "mask_builds": int(out[3]),
            "frame": int(out[4]), "input_change_flushes": int(out[5]), "published": int(out[6]), "feed_opens": int(out[7]),
            "feed_idle": int(out[8]), "feed_open": bool(out[9]), "saw_batch": bool(out[10]), "snapshot_pixels_reread": int(out[11])}


def debug_handover_stats(handle) -> dict:
    """Counters of the hand-over bound (csrc/pt_kernel_common.hpp) through pt_debug_handover_stats — exported, not in the public header.
    Drains the handle.  `inconsistent` must stay 0."""
    L = load()
    if not hasattr(L, "pt_debug_handover_stats"):  # (an A/B library of an earlier round, MI355PT_LIB)
        return {"pairs_repaired": None, "inconsistent": None, "joins_with_repairs": None, "flag_seen": None}
    L.pt_debug_handover_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    L.pt_debug_handover_stats.restype = C.c_int
    out = (C.c_uint * 4)()
    check(L.pt_debug_handover_stats(handle, out), handle)
    return {"pairs_repaired": int(out[0]), "inconsistent": int(out[1]), "joins_with_repairs": int(out[2]), "flag_seen": int(out[3])}
