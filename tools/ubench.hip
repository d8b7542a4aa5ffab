// Micro-benchmarks that inform the integrator's design on gfx950 (development tool, not part of the product):
// plain vs packed fp32 FMA issue rate, v_sqrt_f32 / v_rcp_f32 rate, LDS broadcast ds_read_b128 rate, scalar loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(256) void k(float *out, const float4 *tbl, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f, c = 1e-7f;
    __shared__ float4 lds[256];
    lds[threadIdx.x] = tbl[threadIdx.x];
    __syncthreads();
    if (MODE == 0) { // 8 independent v_fma_f32 per iteration
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
        }
    } else if (MODE == 1) { // 4 v_pk_fma_f32 per iteration (same flops as mode 0)
        float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
        for (int i = 0; i < iters; i++) {
            p0 = __builtin_elementwise_fma(p0, mm, cc); p1 = __builtin_elementwise_fma(p1, mm, cc);
            p2 = __builtin_elementwise_fma(p2, mm, cc); p3 = __builtin_elementwise_fma(p3, mm, cc);
        }
        a0 = p0.x + p0.y; a1 = p1.x + p1.y; a2 = p2.x + p2.y; a3 = p3.x + p3.y;
    } else if (MODE == 2) { // v_sqrt_f32 (hardware approx) x8
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_amdgcn_sqrtf(a0) + 1.0f; a1 = __builtin_amdgcn_sqrtf(a1) + 1.0f; a2 = __builtin_amdgcn_sqrtf(a2) + 1.0f; a3 = __builtin_amdgcn_sqrtf(a3) + 1.0f;
            a4 = __builtin_amdgcn_sqrtf(a4) + 1.0f; a5 = __builtin_amdgcn_sqrtf(a5) + 1.0f; a6 = __builtin_amdgcn_sqrtf(a6) + 1.0f; a7 = __builtin_amdgcn_sqrtf(a7) + 1.0f;
        }
    } else if (MODE == 3) { // IEEE sqrtf x8
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_sqrtf(a0) + 1.0f; a1 = __builtin_sqrtf(a1) + 1.0f; a2 = __builtin_sqrtf(a2) + 1.0f; a3 = __builtin_sqrtf(a3) + 1.0f;
            a4 = __builtin_sqrtf(a4) + 1.0f; a5 = __builtin_sqrtf(a5) + 1.0f; a6 = __builtin_sqrtf(a6) + 1.0f; a7 = __builtin_sqrtf(a7) + 1.0f;
        }
    } else if (MODE == 4) { // LDS broadcast ds_read_b128 + 4 fma
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float4 v = lds[(i * 8 + j) & 255];
                a0 = __builtin_fmaf(a0, v.x, v.y); a1 = __builtin_fmaf(a1, v.z, v.w);
            }
        }
    } else if (MODE == 5) { // scalar (uniform) global load + fma
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float4 v = tbl[(i * 8 + j) & 255];
                a0 = __builtin_fmaf(a0, v.x, v.y); a1 = __builtin_fmaf(a1, v.z, v.w);
            }
        }
    } else if (MODE == 6) { // IEEE division x8
        for (int i = 0; i < iters; i++) {
            a0 = 1.0f / a0 + 1.5f; a1 = 1.0f / a1 + 1.5f; a2 = 1.0f / a2 + 1.5f; a3 = 1.0f / a3 + 1.5f;
            a4 = 1.0f / a4 + 1.5f; a5 = 1.0f / a5 + 1.5f; a6 = 1.0f / a6 + 1.5f; a7 = 1.0f / a7 + 1.5f;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(const char *name, double ops_per_iter)
{
    int blocks = 256 * 8, iters = 4096;
    float *out; float4 *tbl;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&tbl, 256 * 16);
    std::vector<float> h(1024, 1.0001f); hipMemcpy(tbl, h.data(), 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, tbl, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = blocks * 4.0, winstr = waves * iters * ops_per_iter;
    // per-SIMD issue cycles per wave-instruction at 2.4 GHz: time * 2.4e9 * 1024 SIMDs / winstr
    printf("%-28s %8.3f ms  %7.2f cycles/wave-op/SIMD (@2.4GHz)  %8.1f G wave-ops/s\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / winstr, winstr / ms / 1e6);
    hipFree(out); hipFree(tbl);
}
int main()
{
    run<0>("v_fma_f32 x8", 8); run<1>("v_pk_fma_f32 x4 (=8 fma)", 4); run<2>("v_sqrt_f32 x8 (+add)", 8); run<3>("ieee sqrtf x8 (+add)", 8);
    run<4>("ds_read_b128 bcast + 2 fma x8", 8); run<5>("s_load x4 + 2 fma x8", 8); run<6>("ieee 1/x x8 (+add)", 8);
    return 0;
}
