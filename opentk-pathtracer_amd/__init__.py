"""opentk-pathtracer_amd — MI355X (gfx950) replacement for the path-tracing compute dispatch of
BoyBaykiller/OpenTK-PathTracer, behind the C ABI of include/mi355pt.h.

The directory name contains a hyphen (repo contract), so import it through __graft_entry__.load_package(),
which registers it as the module `opentk_pathtracer_amd`.

  csrc/            hand-written HIP kernels (pt_integrate_persistent.hip, pt_integrate_multisample.hip, pt_helper_kernels.hip + device functions pt_device.hpp / pt_atmosphere.hpp), device math
                   contract (pt_math.hpp), C ABI (mi355pt.cpp)
  native.py        hipcc build recipe + ctypes binding of libmi355pt.so (fails loudly; no CPU fallback)
  path_tracer.py   host-side mirror of the reference's PathTracer / AtmosphericScatterer classes over the C ABI
  scene.py         Material / Sphere / Cuboid std140 packers + the reference's default scene data
  camera.py        Camera + OpenTK matrix formulas -> BasicDataUBO / AtmosphericDataUBO blobs
  envmap.py        procedural stand-in environment cubes
  distributed.py   row-block tiling across GPUs (one process per GPU) + gather at present time
  checkpoint.py    accumulation checkpoints (raw RGBA32F + frame index) and PNG screenshots
"""
from . import camera, checkpoint, envmap, native, scene  # noqa: F401
from .path_tracer import AtmosphericScatterer, EnvironmentMap, PathTracer  # noqa: F401

__all__ = ["camera", "envmap", "native", "scene", "PathTracer", "AtmosphericScatterer", "EnvironmentMap"]
