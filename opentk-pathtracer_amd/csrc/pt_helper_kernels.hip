// pt_helper_kernels.hip — the small kernels around the integrator (gfx950): zero-fill, alpha restore, multi-GPU band assembly,
// environment read-back, the ACES + gamma post-process and the atmosphere environment precompute.
// Build flags (see __graft_entry__.build): -O3 -ffp-contract=off -fno-fast-math --offload-arch=gfx950
#include "pt_atmosphere.hpp"
#include "pt_device.hpp"
#include "pt_kernels.hpp"
#include "pt_math.hpp"

namespace pt {

// ---------------------------------------------------------------------------------------------- clear
__global__ void pt_clear_kernel(float4 *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

hipError_t launch_clear(float4 *p, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pt_clear_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, n);
    return hipGetLastError();
}

__global__ void pt_set_alpha_kernel(float4 *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i].w = 1.0f;
}

hipError_t launch_set_alpha(float4 *p, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pt_set_alpha_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- multi-GPU gather
// Un-band the parts' compact rows (see AssembleArgs).  One thread per pixel, 16-byte or 4-byte elements, fully
// coalesced on both sides (a row is contiguous in the stage and in the image).
template <typename T>
__global__ __launch_bounds__(256) void pt_assemble_bands_kernel(const AssembleArgs a)
{
    const size_t n = (size_t)a.width * a.height;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const int y = (int)(i / a.width), x = (int)(i - (size_t)y * a.width);
        const int band = y / a.bandRows, g = band % a.world, lb = band / a.world;
        const size_t ly = (size_t)lb * a.bandRows + (size_t)(y - band * a.bandRows); // row inside part g's compact storage
        ((T *)a.out)[i] = ((const T *)a.stage)[a.partOffset[g] + ly * a.width + x];
    }
}

hipError_t launch_assemble_bands(const AssembleArgs &a, hipStream_t stream)
{
    const size_t n = (size_t)a.width * a.height;
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (a.bytesPerPixel == 16) hipLaunchKernelGGL(pt_assemble_bands_kernel<float4>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(pt_assemble_bands_kernel<uchar4>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

__global__ void pt_env_to_float_kernel(const void *env, int size, int format, const float *lut, float4 *out)
{
    size_t n = (size_t)6 * size * size;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (format == 0) {
        out[i] = ((const float4 *)env)[i];
    } else {
        uchar4 t = ((const uchar4 *)env)[i];
        out[i] = make_float4(lut[t.x], lut[t.y], lut[t.z], (float)t.w / 255.0f);
    }
}

hipError_t launch_env_to_float(const void *env, int envSize, int envFormat, const float *srgbLut, float4 *out,
                               hipStream_t stream)
{
    size_t n = (size_t)6 * envSize * envSize;
    hipLaunchKernelGGL(pt_env_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, env, envSize,
                       envFormat, srgbLut, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- post-process
// /root/reference/OpenTK-PathTracer/res/shaders/PostProcessing/fragment.glsl:17-26 (ScreenEffect.Render,
// src/Render/ScreenEffect.cs:29-37, into an RGBA8 target): ACES tone map + gamma 2.4, alpha = 1.  HBM-bound
// elementwise pass: 16 B read + 4 B written per pixel.
__global__ __launch_bounds__(256) void pt_postprocess_kernel(const float4 *accum, uchar4 *out, size_t n)
{
    // a few microseconds of work that the present path launches BESIDE resident persistent wavefronts (mi355pt.cpp,
    // pt_present_rgba8_async): take the issue slots first, or it runs at a sixth of its speed (75 instead of 11 us)
    __builtin_amdgcn_s_setprio(3);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        float4 c = accum[i];
        uchar4 o;
        o.x = to_unorm8(linear_to_inverse_gamma(aces_film(c.x), 2.4f));
        o.y = to_unorm8(linear_to_inverse_gamma(aces_film(c.y), 2.4f));
        o.z = to_unorm8(linear_to_inverse_gamma(aces_film(c.z), 2.4f));
        o.w = 255;
        out[i] = o;
    }
}

hipError_t launch_postprocess(const float4 *accum, void *outRgba8, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pt_postprocess_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, accum, (uchar4 *)outRgba8, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- atmosphere
// device functions: pt_atmosphere.hpp
__global__ __launch_bounds__(256) void atmo_precompute_kernel(const AtmoArgs a)
{
    const int S = a.size;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)6 * S * S) return;
    int x = (int)(i % S), y = (int)((i / S) % S), face = (int)(i / ((size_t)S * S));
    // main :30-56 — ndc from the texel's integer coordinate (no half-texel offset)
    float ndcx = f_fma((float)x / (float)S, 2.0f, -1.0f), ndcy = f_fma((float)y / (float)S, 2.0f, -1.0f);
    float eye[4], wd[4];
    mat_vec(a.invProj, ndcx, ndcy, -1.0f, 0.0f, eye);
    mat_vec(a.invView[face], eye[0], eye[1], -1.0f, 0.0f, wd);
    v3 dir = v_normalize(V(wd[0], wd[1], wd[2]));
    v3 col = atmosphere(dir, V(0.0f, 6376e3f, 0.0f), V(a.lightPos[0], a.lightPos[1], a.lightPos[2]), a.lightIntensity,
                        6371e3f, 6471e3f, V(5.5e-6f, 13.0e-6f, 22.4e-6f), 21e-6f, 8e3f, 1.2e3f, 0.758f, a.iSteps,
                        a.jSteps);
    a.out[i] = make_float4(col.x, col.y, col.z, 1.0f);
}

hipError_t launch_atmosphere(const AtmoArgs &a, hipStream_t stream)
{
    size_t n = (size_t)6 * a.size * a.size;
    hipLaunchKernelGGL(atmo_precompute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace pt
