// Issue-cost model of the integrator's instruction mix on gfx950, measured in SHADER CLOCKS per wave-instruction PER SIMD at the
// occupancy the persistent kernel runs at (6 wavefronts per SIMD, every CU busy): which instruction forms are full rate, what scalar
// instructions and taken branches beside them cost, and what a dependent chain costs one wavefront.  Development tool (round 4):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench3.hip -o tools/ubench3.bin && tools/ubench3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X X X X X X X X
extern __shared__ float4 lds_dummy[];
template <int MODE> __global__ __launch_bounds__(256, 8) void k(float *out, unsigned long long *t, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f + seed * 1e-9f, c = 1e-7f + seed * 1e-12f, d = seed * 1e-3f;
    int sacc = iters;
    asm volatile("" : "+v"(m), "+v"(c), "+v"(d));
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { // 8 independent v_fma_f32 VOP3, 3 VGPR sources (dest = first source)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 1) { // 8 independent v_fmac_f32 (VOP2, two-address)
            asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                         "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 2) { // 8 independent v_mul_f32 (VOP2)
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 3) { // 8 v_fma_f32 with FOUR distinct registers each (dest != any source)
            asm volatile("v_fma_f32 %0, %1, %8, %9\n v_fma_f32 %1, %2, %8, %10\n v_fma_f32 %2, %3, %9, %10\n v_fma_f32 %3, %4, %8, %9\n"
                         "v_fma_f32 %4, %5, %8, %10\n v_fma_f32 %5, %6, %9, %10\n v_fma_f32 %6, %7, %8, %9\n v_fma_f32 %7, %0, %8, %10"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "v"(d));
        } else if (MODE == 4) { // 8 v_mul_f32 with one scalar instruction after each
            asm volatile("v_mul_f32 %0, %0, %9\n s_add_i32 %8, %8, 1\n v_mul_f32 %1, %1, %9\n s_add_i32 %8, %8, 1\n v_mul_f32 %2, %2, %9\n s_add_i32 %8, %8, 1\n"
                         "v_mul_f32 %3, %3, %9\n s_add_i32 %8, %8, 1\n v_mul_f32 %4, %4, %9\n s_add_i32 %8, %8, 1\n v_mul_f32 %5, %5, %9\n s_add_i32 %8, %8, 1\n"
                         "v_mul_f32 %6, %6, %9\n s_add_i32 %8, %8, 1\n v_mul_f32 %7, %7, %9\n s_add_i32 %8, %8, 1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(sacc) : "v"(m) : "scc");
        } else if (MODE == 5) { // 8 v_mul_f32 with TWO scalar instructions after each
            asm volatile("v_mul_f32 %0, %0, %9\n s_add_i32 %8, %8, 1\n s_xor_b32 %8, %8, 5\n v_mul_f32 %1, %1, %9\n s_add_i32 %8, %8, 1\n s_xor_b32 %8, %8, 5\n"
                         "v_mul_f32 %2, %2, %9\n s_add_i32 %8, %8, 1\n s_xor_b32 %8, %8, 5\n v_mul_f32 %3, %3, %9\n s_add_i32 %8, %8, 1\n s_xor_b32 %8, %8, 5\n"
                         "v_mul_f32 %4, %4, %9\n s_add_i32 %8, %8, 1\n s_xor_b32 %8, %8, 5\n v_mul_f32 %5, %5, %9\n s_add_i32 %8, %8, 1\n s_xor_b32 %8, %8, 5\n"
                         "v_mul_f32 %6, %6, %9\n s_add_i32 %8, %8, 1\n s_xor_b32 %8, %8, 5\n v_mul_f32 %7, %7, %9\n s_add_i32 %8, %8, 1\n s_xor_b32 %8, %8, 5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(sacc) : "v"(m) : "scc");
        } else if (MODE == 6) { // 8 v_mul_f32, a TAKEN scalar branch after every second one
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n s_branch 1f\n s_nop 0\n 1: v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n s_branch 2f\n s_nop 0\n"
                         "2: v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n s_branch 3f\n s_nop 0\n 3: v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n s_branch 4f\n s_nop 0\n 4:"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 7) { // 8 v_mul_f32, a NOT-taken conditional branch after every second one
            asm volatile("s_cmp_eq_u32 0, 1\n v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n s_cbranch_scc1 1f\n 1: v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n s_cbranch_scc1 2f\n"
                         "2: v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n s_cbranch_scc1 3f\n 3: v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n s_cbranch_scc1 4f\n 4:"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "scc");
        } else if (MODE == 8) { // one DEPENDENT chain of 8 v_fma_f32
            asm volatile(R8("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(m), "v"(c));
        } else if (MODE == 9) { // 8 x (v_cmp + s_and_saveexec + s_or exec): the per-sphere branch skeleton without the branch
            asm volatile(R8("v_cmp_ngt_f32 vcc, 0, %0\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n v_mul_f32 %0, %0, %1\n")
                         : "+v"(a0) : "v"(m) : "vcc", "s20", "s21", "scc");
        } else if (MODE == 10) { // 8 v_sub/v_mul mix with an SGPR operand
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                         "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));
        } else if (MODE == 11) { // 4 v_mov_b32 + 4 v_mul
            asm volatile("v_mov_b32 %0, %1\n v_mul_f32 %1, %1, %8\n v_mov_b32 %2, %3\n v_mul_f32 %3, %3, %8\n v_mov_b32 %4, %5\n v_mul_f32 %5, %5, %8\n v_mov_b32 %6, %7\n v_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 12) { // v_cmp (VOPC -> vcc) x8
            asm volatile(R8("v_cmp_lt_f32 vcc, %0, %1\n") : : "v"(a0), "v"(m) : "vcc");
        } else if (MODE == 13) { // v_cmp_e64 -> SGPR pair x8
            asm volatile(R8("v_cmp_lt_f32 s[20:21], %0, %1\n") : : "v"(a0), "v"(m) : "s20", "s21");
        } else if (MODE == 14) { // v_cndmask_b32 x8 (reads vcc)
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
        } else if (MODE == 15) { // v_mul_lo_u32 x8
            asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                         "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 16) { // v_max3_f32 x8
            asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                         "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 17) { // v_fmaak_f32 (literal) x8
            asm volatile("v_fmaak_f32 %0, %0, %8, 0x3fc00000\n v_fmaak_f32 %1, %1, %8, 0x3fc00000\n v_fmaak_f32 %2, %2, %8, 0x3fc00000\n v_fmaak_f32 %3, %3, %8, 0x3fc00000\n"
                         "v_fmaak_f32 %4, %4, %8, 0x3fc00000\n v_fmaak_f32 %5, %5, %8, 0x3fc00000\n v_fmaak_f32 %6, %6, %8, 0x3fc00000\n v_fmaak_f32 %7, %7, %8, 0x3fc00000"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 18) { // v_fma_f32 with a negated source (VOP3 modifiers) x8
            asm volatile("v_fma_f32 %0, -%0, %8, %9\n v_fma_f32 %1, -%1, %8, %9\n v_fma_f32 %2, -%2, %8, %9\n v_fma_f32 %3, -%3, %8, %9\n"
                         "v_fma_f32 %4, -%4, %8, %9\n v_fma_f32 %5, -%5, %8, %9\n v_fma_f32 %6, -%6, %8, %9\n v_fma_f32 %7, -%7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 19) { // v_lshrrev_b32 / v_xor_b32 / v_add_u32 mix x8 (PCG bookkeeping)
            asm volatile("v_lshrrev_b32 %0, 3, %0\n v_xor_b32 %1, %1, %0\n v_add_u32 %2, %2, %1\n v_lshrrev_b32 %3, 5, %3\n"
                         "v_xor_b32 %4, %4, %3\n v_add_u32 %5, %5, %4\n v_lshrrev_b32 %6, 7, %6\n v_xor_b32 %7, %7, %6"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 20) { // v_cvt_f32_u32 x8
            asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n"
                         "v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 21) { // v_readlane_b32 / v_writelane_b32 pairs x4 (SGPR spills)
            asm volatile("v_writelane_b32 %0, s20, 1\n v_readlane_b32 s21, %0, 2\n v_writelane_b32 %1, s20, 1\n v_readlane_b32 s21, %1, 2\n"
                         "v_writelane_b32 %2, s20, 1\n v_readlane_b32 s21, %2, 2\n v_writelane_b32 %3, s20, 1\n v_readlane_b32 s21, %3, 2"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21");
        } else if (MODE == 22) { // 4 broadcast ds_read_b128 + wait + 8 v_mul (the sphere step's skeleton)
            float4 q0, q1, q2, q3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                         : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(sacc & 1023) : "memory");
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %10\n v_mul_f32 %3, %3, %11\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %10\n v_mul_f32 %7, %7, %11"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q0.x), "v"(q1.y), "v"(q2.z), "v"(q3.w));
        } else if (MODE == 23) { // sphere discriminant as compiled today: 3 sub, mul, 2 fmac, mul, 2 fmac, 2 fma(VOP3), cmp  [12 VALU]
            asm volatile("v_sub_f32 %0, %8, %1\n v_sub_f32 %2, %8, %3\n v_mul_f32 %4, %9, %0\n v_mul_f32 %0, %0, %0\n v_sub_f32 %5, %8, %6\n"
                         "v_fmac_f32 %0, %2, %2\n v_fmac_f32 %4, %9, %2\n v_fmac_f32 %0, %5, %5\n v_fmac_f32 %4, %9, %5\n"
                         "v_fma_f32 %2, -%7, %7, %0\n v_fma_f32 %5, %4, %4, -%2\n v_cmp_ngt_f32 vcc, 0, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
        } else if (MODE == 24) { // sphere discriminant inside a run: 1 sub, 6 v_fma_f32 (VOP3), cmp [8 VALU]
            asm volatile("v_sub_f32 %0, %8, %1\n v_fma_f32 %2, %9, %0, %3\n v_fma_f32 %4, %0, %0, %5\n v_fma_f32 %2, %9, %6, %2\n v_fma_f32 %4, %6, %6, %4\n"
                         "v_fma_f32 %4, -%7, %7, %4\n v_fma_f32 %2, %2, %2, -%4\n v_cmp_ngt_f32 vcc, 0, %2"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
        } else if (MODE == 25) { // (v_cmp -> vcc, v_cndmask reading vcc) x4 [8 VALU]
            asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n"
                         "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");
        } else if (MODE == 26) { // (v_cmp -> SGPR pair, v_cndmask reading it) x4 [8 VALU]
            asm volatile("v_cmp_lt_f32 s[20:21], %0, %4\n v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cmp_lt_f32 s[22:23], %1, %4\n v_cndmask_b32 %1, %1, %4, s[22:23]\n"
                         "v_cmp_lt_f32 s[20:21], %2, %4\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cmp_lt_f32 s[22:23], %3, %4\n v_cndmask_b32 %3, %3, %4, s[22:23]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "s20", "s21", "s22", "s23");
        } else if (MODE == 27) { // 4 independent v_cmp first, then 4 v_cndmask (compare results in 4 SGPR pairs) [8 VALU]
            asm volatile("v_cmp_lt_f32 s[20:21], %0, %4\n v_cmp_lt_f32 s[22:23], %1, %4\n v_cmp_lt_f32 s[24:25], %2, %4\n v_cmp_lt_f32 s[26:27], %3, %4\n"
                         "v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[22:23]\n v_cndmask_b32 %2, %2, %4, s[24:25]\n v_cndmask_b32 %3, %3, %4, s[26:27]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (MODE == 28) { // v_cndmask on an SGPR pair that nothing in the loop writes x8
            asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
                         "v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 29) { // v_add_f32 x8 (VOP2)
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 30) { // 32 v_mul_f32 per iteration (loop overhead amortised)
            asm volatile(R8("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
        } else if (MODE == 31) { // 32 v_fma_f32 (VOP3) per iteration
            asm volatile(R8("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        } else if (MODE == 32) { // 32 per iteration: v_mul / v_fma alternating
            asm volatile(R8("v_mul_f32 %0, %0, %4\n v_fma_f32 %1, %1, %4, %5\n v_mul_f32 %2, %2, %4\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        } else if (MODE >= 33 && MODE <= 39) {
            // LDS read cost against VALU work: R broadcast reads of width B (all lanes one address), then V v_mul.
            //   33: 4 x b32 + 8   34: 4 x b64 + 8   35: 2 x b128 + 8   36: 1 x b128 + 8   37: 4 x b128 + 40   38: 2 x b128 + 40   39: 0 reads + 40
            float4 q0 = make_float4(m, m, m, m), q1 = q0, q2 = q0, q3 = q0;
            if (MODE == 33)
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:16\n ds_read_b32 %2, %4 offset:32\n ds_read_b32 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                             : "=v"(q0.x), "=v"(q1.y), "=v"(q2.z), "=v"(q3.w) : "v"(sacc & 1023) : "memory");
            if (MODE == 34) {
                float2 p0, p1, p2, p3;
                asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:16\n ds_read_b64 %2, %4 offset:32\n ds_read_b64 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                             : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(sacc & 1023) : "memory");
                q0.x = p0.x; q1.y = p1.y; q2.z = p2.x; q3.w = p3.y;
            }
            if (MODE == 35 || MODE == 38)
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1) : "v"(sacc & 1023) : "memory");
            if (MODE == 36) asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q0) : "v"(sacc & 1023) : "memory");
            if (MODE == 37)
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                             : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(sacc & 1023) : "memory");
#define MUL8 "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %10\n v_mul_f32 %3, %3, %11\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %10\n v_mul_f32 %7, %7, %11\n"
            if (MODE >= 37)
                asm volatile(MUL8 MUL8 MUL8 MUL8 MUL8
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q0.x), "v"(q1.y), "v"(q2.z), "v"(q3.w));
            else
                asm volatile(MUL8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q0.x), "v"(q1.y), "v"(q2.z), "v"(q3.w));
#undef MUL8
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)sacc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}
template <int MODE> void run(const char *name, int valuPerIter, int blocksPerCU)
{
    // THROUGHPUT of a full machine: 256 CUs x blocksPerCU workgroups of 4 wavefronts, long enough that ramp and tail vanish; the
    // launch's wall time from HIP events (a single wavefront's own clock says nothing about what the SIMD sustains: the dispatcher
    // does not keep all workgroups of a large grid resident together)
    const int iters = 100000, blocks = 256 * blocksPerCU, threads = 256;
    float *out; unsigned long long *t, ht[2];
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&t, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 4096, 0, out, t, iters / 10, 1.0f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 4096, 0, out, t, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize(); hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double mhz = (double)ht[0] / ht[1] * 100.0;
    const double instrPerSimd = (double)iters * valuPerIter * blocksPerCU; // 4 wavefronts per workgroup on 4 SIMDs
    printf("%-58s %d waves/SIMD: %6.3f shader clocks per VALU instruction per SIMD (launch %.3f ms at %4.0f MHz; one wavefront: %.2f clocks per instruction)\n",
           name, blocksPerCU, ms * 1e-3 * mhz * 1e6 / instrPerSimd, ms, mhz, (double)ht[0] / ((double)iters * valuPerIter));
    hipFree(out); hipFree(t);
}
#define ALL(W)                                                                 \
    run<30>("v_mul_f32 x32", 32, W);                                           \
    run<31>("v_fma_f32 (VOP3) x32", 32, W);                                    \
    run<32>("v_mul_f32 / v_fma_f32 alternating x32", 32, W);                   \
    run<29>("v_add_f32 (VOP2) x8", 8, W);                                      \
    run<14>("v_cndmask_b32 on vcc (asm clobbers vcc) x8", 8, W);               \
    run<28>("v_cndmask_b32 on an untouched SGPR pair x8", 8, W);               \
    run<25>("(v_cmp -> vcc, v_cndmask) x4 [8 VALU]", 8, W);                    \
    run<26>("(v_cmp -> SGPR pair, v_cndmask) x4 [8 VALU]", 8, W);              \
    run<27>("4 v_cmp -> SGPR pairs, then 4 v_cndmask [8 VALU]", 8, W);
#define LDS(W)                                                                 \
    run<22>("4 broadcast ds_read_b128 + wait + 8 v_mul", 8, W);                \
    run<33>("4 broadcast ds_read_b32 + wait + 8 v_mul", 8, W);                 \
    run<34>("4 broadcast ds_read_b64 + wait + 8 v_mul", 8, W);                 \
    run<35>("2 broadcast ds_read_b128 + wait + 8 v_mul", 8, W);                \
    run<36>("1 broadcast ds_read_b128 + wait + 8 v_mul", 8, W);                \
    run<39>("40 v_mul", 40, W);                                                \
    run<37>("4 broadcast ds_read_b128 + wait + 40 v_mul", 40, W);              \
    run<38>("2 broadcast ds_read_b128 + wait + 40 v_mul", 40, W);
int main(int argc, char **argv)
{
    const bool lds = argc > 1 && argv[1][0] == 'l'; // "lds": what a broadcast LDS read costs beside VALU work
    for (int w : {1, 6}) {
        printf("--- %d wavefront(s) per SIMD\n", w);
        if (lds) { LDS(w) } else { ALL(w) }
    }
    return 0;
}
