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
// One lane per cube texel — of HALF the cube when the sun stands in the plane x = 0, which is where the reference's host always puts it
// (AtmosphericScatterer.cs:35-45: LightPosition = (0, sin, cos) x 1.496e11).  The scattering integral depends on the view direction r
// only through dot products with r0 = (0, R, 0) and pSun = (0, sy, sz) and through squares of the x components of points r0 + t r and
// their offsets along pSun: negating r.x negates those x components and changes nothing else, bit for bit.  The cube's texel grid is
// symmetric under x -> -x (texel x of faces +-Y, +-Z pairs with texel S - x of the same face, +X with -X; column x = 0 has no partner
// because ndc = 2 x / S - 1 never reaches +1), so a lane computes its texel's direction AND its partner's (two small matrix products),
// and when pSun.x == 0 and the two directions are exact mirror images the lower texel of the pair computes the colour once and stores it
// twice, so the kernel does half the work.  Any other sun position or view matrix fails the test and the lane computes its partner's
// colour too: every texel is then evaluated on its own, as before.  Verified against the oracle,
// which computes every texel directly: bit-identical cubes (tests/test_gpu_parity.py, sizes 24 ... 2048).
PT_DEV v3 atmo_texel_direction(const AtmoArgs &a, int face, int x, int y)
{
    // main :30-56 — ndc from the texel's integer coordinate (no half-texel offset)
    const int S = a.size;
    float ndcx = f_fma((float)x / (float)S, 2.0f, -1.0f), ndcy = f_fma((float)y / (float)S, 2.0f, -1.0f);
    float eye[4], wd[4];
    mat_vec(a.invProj, ndcx, ndcy, -1.0f, 0.0f, eye);
    mat_vec(a.invView[face], eye[0], eye[1], -1.0f, 0.0f, wd);
    return v_normalize(V(wd[0], wd[1], wd[2]));
}

// canonical texels per cube row: all S of face +X, column 0 of face -X, and columns 0 .. S / 2 of the four other faces
__host__ __device__ inline int atmo_half_columns(int S) { return S / 2 + 1; }
__host__ __device__ inline int atmo_row_lanes(int S) { return S + 1 + 4 * atmo_half_columns(S); }

__global__ __launch_bounds__(256) void atmo_precompute_kernel(const AtmoArgs a)
{
    const int S = a.size, C = atmo_half_columns(S), T = atmo_row_lanes(S);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)S * T) return;
    // lane -> the LOWER texel of a mirrored pair, or a texel without partner (the grid holds only those: no idle wavefronts, and the
    // work spreads over the 8 XCDs whatever the size — with one lane per texel the leaving half landed on 4 of the 8)
    const int y = (int)(i / T);
    int r = (int)(i % T), face, x;
    if (r < S) { face = 0; x = r; }
    else if (r == S) { face = 1; x = 0; }
    else { r -= S + 1; face = 2 + r / C; x = r % C; }
    const v3 dir = atmo_texel_direction(a, face, x, y);
    const v3 r0 = V(0.0f, 6376e3f, 0.0f), sun = V(a.lightPos[0], a.lightPos[1], a.lightPos[2]), kRlh = V(5.5e-6f, 13.0e-6f, 22.4e-6f);
    const v3 col = atmosphere(dir, r0, sun, a.lightIntensity, 6371e3f, 6471e3f, kRlh, 21e-6f, 8e3f, 1.2e3f, 0.758f, a.iSteps, a.jSteps);
    a.out[((size_t)face * S + y) * S + x] = make_float4(col.x, col.y, col.z, 1.0f);
    // the texel's mirror image under x -> -x, if it has one that is not itself
    const int mface = face < 2 ? 1 - face : face, mx = S - x;
    if (x < 1 || (mface == face && mx == x)) return;
    const v3 pdir = atmo_texel_direction(a, mface, mx, y);
    v3 pcol = col;
    if (!(a.lightPos[0] == 0.0f && pdir.x == -dir.x && pdir.y == dir.y && pdir.z == dir.z)) // not an exact mirror image: computed on its own
        pcol = atmosphere(pdir, r0, sun, a.lightIntensity, 6371e3f, 6471e3f, kRlh, 21e-6f, 8e3f, 1.2e3f, 0.758f, a.iSteps, a.jSteps);
    a.out[((size_t)mface * S + y) * S + mx] = make_float4(pcol.x, pcol.y, pcol.z, 1.0f);
}

hipError_t launch_atmosphere(const AtmoArgs &a, hipStream_t stream)
{
    const size_t n = (size_t)a.size * atmo_row_lanes(a.size);
    hipLaunchKernelGGL(atmo_precompute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace pt
