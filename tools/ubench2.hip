// Issue cost of wave64 VALU instructions on gfx950 in SHADER CLOCKS (s_memtime), independent of the clock frequency,
// and the shader clock itself (s_memtime vs the 100 MHz s_memrealtime).  Development tool.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o tools/ubench2.bin && tools/ubench2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, unsigned long long *t, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f, c = 1e-7f;
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { // fma
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
        } else if (MODE == 1) { // mul
            a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m;
        } else if (MODE == 2) { // add
            a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
        } else if (MODE == 3) { // cmp + cndmask
            a0 = a0 > a1 ? a0 : m; a1 = a1 > a2 ? a1 : c; a2 = a2 > a3 ? a2 : m; a3 = a3 > a4 ? a3 : c;
            a4 = a4 > a5 ? a4 : m; a5 = a5 > a6 ? a5 : c; a6 = a6 > a7 ? a6 : m; a7 = a7 > a0 ? a7 : c;
        } else if (MODE == 4) { // max
            a0 = __builtin_fmaxf(a0, a1); a1 = __builtin_fmaxf(a1, a2); a2 = __builtin_fmaxf(a2, a3); a3 = __builtin_fmaxf(a3, a4);
            a4 = __builtin_fmaxf(a4, a5); a5 = __builtin_fmaxf(a5, a6); a6 = __builtin_fmaxf(a6, a7); a7 = __builtin_fmaxf(a7, m);
        } else if (MODE == 5) { // integer: xor + shift (pcg-like)
            unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3);
            u0 = (u0 >> 3) ^ u1; u1 = (u1 >> 5) ^ u2; u2 = (u2 >> 7) ^ u3; u3 = (u3 >> 9) ^ u0;
            a0 = __uint_as_float(u0); a1 = __uint_as_float(u1); a2 = __uint_as_float(u2); a3 = __uint_as_float(u3);
        } else if (MODE == 6) { // v_mul_lo_u32
            unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3);
            u0 *= 747796405u; u1 *= 747796405u; u2 *= 277803737u; u3 *= 277803737u;
            u0 += 1; u1 += 1; u2 += 1; u3 += 1;
            a0 = __uint_as_float(u0); a1 = __uint_as_float(u1); a2 = __uint_as_float(u2); a3 = __uint_as_float(u3);
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}
template <int MODE> void run(const char *name, double ops, int blocks, int threads)
{
    int iters = 20000;
    float *out; unsigned long long *t, ht[2];
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&t, 16);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, t, iters, 1.0f);
    hipDeviceSynchronize(); hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost);
    double wavesPerSimd = (threads / 64) / 4.0; if (wavesPerSimd < 1) wavesPerSimd = 1; // one block per CU below
    printf("%-22s blocks %5d x %4d thr: %7.3f shader clocks per wave-instruction per wave, clock %.0f MHz\n", name, blocks, threads,
           (double)ht[0] / (iters * ops), (double)ht[0] / ht[1] * 100.0);
    hipFree(out); hipFree(t);
}
int main()
{
    // 1 wave on the chip: pure issue cost; then 4 waves/CU (1 per SIMD), 8, 16, 32 per CU with every CU busy
    for (int cfg = 0; cfg < 4; cfg++) {
        int blocks = cfg == 0 ? 1 : 256, threads = cfg == 0 ? 64 : (cfg == 1 ? 256 : (cfg == 2 ? 512 : 1024));
        run<0>("v_fma_f32 x8", 8, blocks, threads); run<1>("v_mul_f32 x8", 8, blocks, threads); run<2>("v_add_f32 x8", 8, blocks, threads);
        run<3>("v_cmp+v_cndmask x8", 16, blocks, threads); run<4>("v_max_f32 x8", 8, blocks, threads);
        run<5>("v_lshr+v_xor x4", 8, blocks, threads); run<6>("v_mul_lo_u32+add x4", 8, blocks, threads);
    }
    return 0;
}
