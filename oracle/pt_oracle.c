/*
 * pt_oracle.c — CPU restatement of the reference's path-tracing integrator.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this file.
 * The product (opentk-pathtracer_amd/csrc, libmi355pt.so) never includes, links or falls back to it.
 *
 * What it restates, function by function (all paths relative to /root/reference/OpenTK-PathTracer/):
 *   res/shaders/PathTracing/compute.glsl:101-369          the integrator (cited per function below)
 *   res/shaders/AtmosphericScattering/compute.glsl:30-171 the atmosphere env-map precompute
 *   res/shaders/PostProcessing/fragment.glsl:17-44        ACES tone map + gamma -> RGBA8 (the step after the path)
 *   src/Render/PathTracer.cs:114-129                      frame counter / dispatch semantics
 *
 * PINNING: the reference has no tests or golden vectors of its own.  This oracle is
 * pinned against outputs of the reference ITSELF run in the build container: the unmodified GLSL executed
 * by Mesa llvmpipe through oracle/glsl_ref/glsl_runner.c; the resulting fixtures are committed under
 * tests/golden/ (generator: tests/golden/make_golden.py) and checked by tests/test_oracle_vs_reference.py.
 *
 * ARITHMETIC CONTRACT ("pt-f32", shared with the HIP kernel so that HIP == oracle BIT-FOR-BIT):
 *   - every value is IEEE-754 binary32; +,-,* are correctly rounded; denormals are kept;
 *   - a*b+c is fused ONLY where this file writes fmaf() — compile with -ffp-contract=off;
 *   - dot(a,b)      = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   - 1/x           = f_rcp(x): bit-trick seed 0x7EF311C7 - bits(x), three Newton steps y += y*fma(-x,y,1)
 *                     (measured <= 0.51 ulp over 6e6 samples); |x| < FLT_MIN -> +-inf.  a/b is evaluated as a * f_rcp(b)
 *                     everywhere on the per-bounce path (GLSL 4.50 section 4.7.1 allows 2.5 ulp for a/b), and a
 *                     vector divided by a scalar uses ONE reciprocal.  Per-frame uniform reciprocals (1/W, 1/H, 1/SPP,
 *                     1/(frame+1)) and 1/radius (computed once per sphere) use the correctly rounded IEEE quotient.
 *   - inversesqrt(x)= f_rsqrt(x): seed 0x5F3759DF - (bits(x)>>1), three Newton steps (<= 1.7 ulp; GLSL allows 2);
 *                     x < FLT_MIN -> +inf (x >= 0) or NaN (x < 0), so normalize(vec3(0)) is still NaN
 *   - normalize(v)  = v * f_rsqrt(dot(v,v))
 *   - sqrt(x)       = pt_sqrt(x) on the per-bounce path (sphere roots, hemisphere / lens sampling, refract): the same
 *                     seed, TWO Newton steps, then s = x*y; s += fma(-s,s,x) * (y/2)  (<= 0.501 ulp; GLSL inherits
 *                     sqrt's precision from 1/inversesqrt = 2 ulp).  The atmosphere precompute uses it for its per-step
 *                     roots and heights too (round 5; the once-per-texel terms keep IEEE sqrtf and `/`).
 *   - mix(x,y,a)    = fma(y, a, x*(1-a))                       (GLSL 4.50 section 8.3 definition)
 *   - min/max       = IEEE minNum/maxNum (fminf/fmaxf); GLSL leaves NaN handling undefined
 *   - sin/cos/exp   = the fixed polynomial algorithms below (<= ~1.5 ulp), pow(x,5) = x*(x^2)^2,
 *                     pow(x,1.5) = x*sqrt(x)
 *   - cuboid slabs  : (Min-O)/D is evaluated as (Min-O) * f_rcp(D) (one reciprocal per ray component instead of six
 *                     divisions per cuboid).  Define PT_SLAB_TRUE_DIVISION to get the literal IEEE a/b form (kept for
 *                     the fidelity study in DESIGN.md; the HIP kernel implements the reciprocal form).
 *   Why software reciprocals: the correctly rounded IEEE divide / sqrt expand to 43 / 52 issue cycles on gfx950
 *   (tools/ubench.hip) and were 22 % of the integrator's vector work; hardware v_rcp/v_rsq cannot be reproduced on a
 *   CPU, these sequences can — so the GPU result stays bit-identical to this file.
 *   GLSL itself leaves precision of all of these implementation-defined; llvmpipe is one realisation, this
 *   contract is another.  The stated tolerance against llvmpipe lives in tests/test_oracle_vs_reference.py.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -shared -fPIC pt_oracle.c -o _build/libpt_oracle.so -lm -lpthread
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PTO_API __attribute__((visibility("default")))

#define FLOAT_MAX 3.4028235e+38f /* compute.glsl:2 */
#define FLOAT_MIN -3.4028235e+38f /* compute.glsl:3 */
#define EPSILON 0.001f            /* compute.glsl:4 */
#define PI 3.14159265f            /* compute.glsl:5 */

typedef struct { float x, y, z; } v3;

/* ------------------------------------------------------------------ pt-f32 primitives */
static inline float f_min(float a, float b) { return fminf(a, b); }
static inline float f_max(float a, float b) { return fmaxf(a, b); }
static inline uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float f_unbits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
/* ---- witness build (-DPT_ORACLE_PERTURB, _build/libpt_oracle_perturb.so; tests/test_decision_margins.py).  GLSL leaves the precision of
 * 1/x, inversesqrt, sqrt, sin, cos, exp implementation-defined: an implementation whose primitive P returns results ONE ULP LARGER (or
 * smaller) in magnitude than this contract's is as conforming as the contract.  pto_set_perturbation(P, ulps) turns this library into
 * that implementation (every call of P, every pixel); the margin test uses the family as constructive witnesses: a pixel of the
 * reference that the contract misses must be HIT by one of its neighbours.  P: 0 rcp, 1 rsqrt, 2 sqrt, 3 sin, 4 cos, 5 exp, 6 pow5. */
#ifdef PT_ORACLE_PERTURB
static int g_perturb_prim = -1, g_perturb_ulps = 0;
/* TARGETED witnesses (round 6).  A global shift of a primitive moves every value of the path; what separates the contract from the
 * reference in an out-of-band pixel is usually ONE comparison that came out the other way.  Two single-site variants, both conforming
 * (GLSL fixes neither the last bits of a primitive at one particular argument nor, therefore, the outcome of a comparison whose operands
 * are closer than those bits):
 *   - g_flip_at[]: the pixel's k-th data-dependent comparison (DECIDE sites: compute.glsl:169,201,208,234,247,269,293,322-332,347-350,
 *     refract's k < 0) is inverted, everything else evaluated by the contract;
 *   - g_tprim / g_tcall / g_tulps [g_tn]: the n-th call of primitive P in this pixel returns a result `ulps` off, every other call the
 *     contract's (up to WIT_MAX_SITES such calls at once: pixels whose paths AMPLIFY — a few bounces on curved surfaces turn one ulp
 *     into 1e-3 of the colour — are reached by a handful of calls a few ulps off, not by one).
 * pto_witness_search tries them for one pixel (single-threaded; the counters are per thread, the targets are globals). */
#define WIT_MAX_CLOSE 2048
#define WIT_MAX_FLIPS 3
static int g_flip_at[WIT_MAX_FLIPS] = { -1, -1, -1 };
#define WIT_MAX_SITES 32
static int g_tn = 0, g_tprim[WIT_MAX_SITES], g_tcall[WIT_MAX_SITES], g_tulps[WIT_MAX_SITES]; /* targeted primitive calls */
static float g_record_gap = 0.0f; /* > 0: record the decisions whose operands are closer than this (relative to their scale) */
static __thread int tl_dec_n, tl_call_n[10], tl_close_n, tl_nan_env;
static __thread struct { int idx; float gap; int line; float diff; } tl_close[WIT_MAX_CLOSE];
static inline float ulp_shift(float y, int ulps)
{
    if (!(fabsf(y) > 1.17549435e-38f) || isinf(y)) return y;
    uint32_t u; memcpy(&u, &y, 4);
    u = (uint32_t)((int32_t)u + ulps); /* (sign-magnitude: + = away from zero) */
    memcpy(&y, &u, 4);
    return y;
}
/* ENSEMBLE members (round 6, pto_set_ensemble; tests/test_ensemble_stability.py).  A member is ONE conforming implementation that differs
 * from the contract everywhere at once, the way a real driver does: its primitive P'(x) = P(x) shifted by s ulps, s a fixed pseudo-random
 * function of (member seed, primitive, the bits of P(x)) in [-a_P, +a_P] with a_P = min(amplitude, GLSL's / the search's allowance for P);
 * each a * b + c is fused or not, each division literal or by reciprocal, as a fixed function of the member and the operands' bits.  The
 * shift depends on the value only, so P' is a function (the same argument gives the same result in every pixel and frame).  A pixel
 * whose value does not move under any member of an ensemble is insensitive to what conforming implementations differ by — the
 * statement the first-order margins can only bound from one side. */
static int g_sig_alpha = 0;            /* pto_set_signature_alpha: the alpha channel carries the pixel's PATH SIGNATURE instead of 1 */
static __thread int tl_ub;            /* the pixel touched something GLSL / GL leave undefined (pow of a base that is negative or within four ulps of zero, a comparison on a NaN, texture(env, NaN)) */
static __thread uint32_t tl_sig;       /* hash of the path's discrete events: object hit, lobe taken, how it ended — per bounce, sample, frame */
static inline void sig_note(uint32_t ev) { tl_sig = (tl_sig ^ ev) * 0x01000193u + 0x9E3779B9u; tl_sig ^= tl_sig >> 15; }
#define SIG_NOTE(ev) sig_note((uint32_t)(ev))
static uint32_t g_ens_seed = 0; /* 0 = off */
static int g_ens_amp = 0;
static const int ens_allow[7] = { 2, 2, 2, 4, 4, 4, 16 }; /* (= wit_ulps below: rcp, rsqrt, sqrt, sin, cos, exp, pow5) */
static inline uint32_t ens_hash(uint32_t a, uint32_t b)
{
    uint32_t h = (g_ens_seed ^ (a * 0x9E3779B9u)) + b * 0x85EBCA6Bu;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
static inline float perturbed(int prim, float y)
{
    const int n = tl_call_n[prim]++;
    for (int t = 0; t < g_tn; t++)
        if (prim == g_tprim[t] && n == g_tcall[t]) return ulp_shift(y, g_tulps[t]);
    if (g_ens_seed != 0 && prim < 7) {
        uint32_t u; memcpy(&u, &y, 4);
        const int a = g_ens_amp < ens_allow[prim] ? g_ens_amp : ens_allow[prim];
        return ulp_shift(y, (int)(ens_hash((uint32_t)prim, u) % (uint32_t)(2 * a + 1)) - a);
    }
    if (prim != g_perturb_prim || g_perturb_ulps == 0) return y;
    return ulp_shift(y, g_perturb_ulps);
}
static inline int decide_at(int cond, float diff, float scale, int line)
{
    const int k = tl_dec_n++;
    if (g_record_gap > 0.0f) {
        const float gap = fabsf(diff) / fmaxf(fabsf(scale), 1e-30f);
        if (gap < g_record_gap && tl_close_n < WIT_MAX_CLOSE) { tl_close[tl_close_n].idx = k; tl_close[tl_close_n].gap = gap; tl_close[tl_close_n].line = line; tl_close[tl_close_n].diff = diff; tl_close_n++; }
    }
    /* an ensemble member decides comparisons ON A NaN for itself (GLSL 4.60 section 4.7.1: "operations and built-in functions that operate on
       a NaN are not required to return a NaN", min / max of a NaN are undefined: after refract() = 0 -> normalize(0) the ray is NaN and
       whether it "hits" a slab is the implementation's choice; llvmpipe's misses everything and looks the environment up at NaN) */
    if (diff != diff) tl_ub = 1;
    if (g_ens_seed != 0 && diff != diff) return (int)(ens_hash(200u + (uint32_t)(k & 15), 0u) >> 31);
    return (k == g_flip_at[0] || k == g_flip_at[1] || k == g_flip_at[2]) ? !cond : cond;
}
#define DECIDE(cond, diff, scale) decide_at((cond), (diff), (scale), __LINE__)
/* "primitive" 7: a * b + c.  GLSL lets an implementation fuse it or not; the contract fuses where this file says fmaf, llvmpipe never
 * does.  A targeted site (any non-zero shift) evaluates that ONE multiply-add the other way; pto_set_unfused(1) all of them (outside
 * the primitives above, whose own Newton steps are part of their definition). */
static int g_unfuse_all = 0, g_pow_neg_nan = 0, g_nan_env_set = 0;
/* base variants (pto_set_base_variant): the searches above run AROUND a conforming implementation, by default the contract; llvmpipe's
 * arithmetic differs from it everywhere at once (correctly rounded 1/x, sqrt, 1/sqrt; the literal a / b; never fused), and a pixel that
 * amplifies is closer to the reference's value from a base that shares those than from the contract */
static int g_base_exact = 0, g_base_truediv = 0;
static int g_base_sampler_lerp = 0; /* (base variant bit 512) */
static int g_base_mix_lerp = 0; /* (base variant bit 256: mix(x, y, a) = x + a (y - x), llvmpipe's form — probed: 100 % bit-identical) */
static int g_base_matvec = 0, g_base_dot = 0; /* (base variants, bits 8 / 16 and 32 / 64: the order in which matrix-vector and dot products sum their terms) */
static float g_nan_env[3]; /* what texture(env, NaN direction) returns instead of the contract's clamped lookup (pto_set_nan_env) */
static inline float wit_fma(float a, float b, float c)
{
    const int n = tl_call_n[7]++;
    int unfused = g_unfuse_all;
    if (g_ens_seed != 0) { /* this member fuses about half of the multiply-adds: a fixed function of the operands */
        uint32_t ua, ub, uc; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4); memcpy(&uc, &c, 4);
        unfused = (int)(ens_hash(7u + ua, ub ^ (uc * 0xC2B2AE35u)) >> 31);
    }
    for (int t = 0; t < g_tn; t++)
        if (g_tprim[t] == 7 && n == g_tcall[t] && g_tulps[t] != 0) unfused = !unfused;
    if (unfused) { const float m = a * b; return m + c; } /* (-ffp-contract=off: two roundings) */
    return __builtin_fmaf(a, b, c);
}
/* "primitive" 8: a / b where the contract multiplies by a reciprocal it has already (sphere normal (p - c) / r, throughput /= prob,
 * throughput /= p: compute.glsl:318,164,170); a targeted site divides.  "primitive" 9: mix(x, y, a) as x + a (y - x) instead of
 * x (1 - a) + y a (GLSL: "the linear blend"; both forms are in use). */
static inline int wit_targeted(int prim)
{
    const int n = tl_call_n[prim]++;
    for (int t = 0; t < g_tn; t++)
        if (g_tprim[t] == prim && n == g_tcall[t] && g_tulps[t] != 0) return 1;
    return 0;
}
static inline float wit_quot(float a, float b, float rb)
{
    int literal = (wit_targeted(8) != 0) != (g_base_truediv != 0);
    if (g_ens_seed != 0) { uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4); literal = (int)(ens_hash(8u + ua, ub) >> 31); }
    return literal ? a / b : a * rb;
}
#define QUOT(a, b, rb) wit_quot((a), (b), (rb))
#define MIX_OTHER_FORM() ((wit_targeted(9) != 0) != (g_base_mix_lerp != 0))
#else
#define QUOT(a, b, rb) ((a) * (rb))
#define MIX_OTHER_FORM() 0
#define perturbed(prim, y) (y)
#define DECIDE(cond, diff, scale) (cond)
#define SIG_NOTE(ev) ((void)0)
#endif
/* pt-f32 reciprocal: seed by exponent negation, three Newton steps; zero and denormals give +-inf */
static inline float f_rcp(float x)
{
#ifdef PT_EXACT_DIVSQRT /* fidelity study only (oracle/Makefile): correctly rounded 1/x, 1/sqrt(x), sqrt(x) as llvmpipe's / and sqrt are */
    return 1.0f / x;
#endif
#ifdef PT_ORACLE_PERTURB
    if (g_base_exact) return perturbed(0, 1.0f / x);
#endif
    float y = f_unbits(0x7EF311C7u - f_bits(x));
    float e = fmaf(-x, y, 1.0f); y = fmaf(y, e, y);
    e = fmaf(-x, y, 1.0f); y = fmaf(y, e, y);
    e = fmaf(-x, y, 1.0f); y = fmaf(y, e, y);
    if (fabsf(x) < 1.17549435e-38f) y = copysignf(INFINITY, x);
    return perturbed(0, y);
}
/* pt-f32 inverse square root: classic seed, three Newton steps; zero/denormal -> +inf, negative -> NaN */
static inline float f_rsqrt(float x)
{
#ifdef PT_EXACT_DIVSQRT
    return 1.0f / sqrtf(x);
#endif
#ifdef PT_ORACLE_PERTURB
    if (g_base_exact) return perturbed(1, 1.0f / sqrtf(x));
#endif
    float y = f_unbits(0x5F3759DFu - (f_bits(x) >> 1));
    float h = 0.5f * x, t;
    t = y * y; t = fmaf(-h, t, 1.5f); y = y * t;
    t = y * y; t = fmaf(-h, t, 1.5f); y = y * t;
    t = y * y; t = fmaf(-h, t, 1.5f); y = y * t;
    if (x < 1.17549435e-38f) y = x < 0.0f ? NAN : INFINITY;
    return perturbed(1, y);
}
/* pt-f32 square root: two Newton steps y *= 1.5 - (x/2*y)*y on the same seed (4.7e-6), then one residual correction
 * s += (x - s*s) * y/2  (<= 0.501 ulp); sqrt(0) = 0 exactly; negative, infinite and NaN inputs give a non-finite value
 * (every call site guards its argument: discriminant >= 0, 1 - z*z >= 0, k >= 0, rand in [0,1]) */
static inline float pt_sqrt(float x)
{
#ifdef PT_EXACT_DIVSQRT
    return sqrtf(x);
#endif
#ifdef PT_ORACLE_PERTURB
    if (g_base_exact) return perturbed(2, sqrtf(x));
#endif
    float y = f_unbits(0x5F3759DFu - (f_bits(x) >> 1));
    float h = 0.5f * x, t;
    t = h * y; t = fmaf(-t, y, 1.5f); y = y * t;
    t = h * y; t = fmaf(-t, y, 1.5f); y = y * t;
    float s = x * y;
    float r = fmaf(-s, s, x);
    return perturbed(2, fmaf(r, 0.5f * y, s));
}
#ifdef PT_ORACLE_PERTURB
#define fmaf(a, b, c) wit_fma((a), (b), (c)) /* (the vector helpers and the integrator; not the primitives) */
#endif
static inline float f_mix(float x, float y, float a)
{
    if (MIX_OTHER_FORM()) return x + a * (y - x);
    return fmaf(y, a, x * (1.0f - a));
}

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v_scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 v_neg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* a + b*s, fused */
static inline v3 v_fma(v3 b, float s, v3 a) { return V(fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z)); }
#ifdef PT_ORACLE_PERTURB
/* an ensemble member also sums the three products of a dot product in an order of its own (GLSL does not fix one): a fixed function of
   the member and the operands */
static inline float v_dot(v3 a, v3 b)
{
    if (g_ens_seed != 0 || g_base_dot != 0) {
        uint32_t ua, ub; memcpy(&ua, &a.x, 4); memcpy(&ub, &b.y, 4);
        switch (g_ens_seed != 0 ? ens_hash(11u + ua, ub) % 3u : (uint32_t)g_base_dot) {
        case 1: return fmaf(a.x, b.x, fmaf(a.z, b.z, a.y * b.y));
        case 2: return fmaf(a.y, b.y, fmaf(a.x, b.x, a.z * b.z));
        default: break;
        }
    }
    return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x));
}
#else
static inline float v_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
#endif
static inline v3 v_normalize(v3 a) { return v_scale(a, f_rsqrt(v_dot(a, a))); }
static inline v3 v_mix(v3 x, v3 y, float a)
{
    float ia = 1.0f - a;
    if (MIX_OTHER_FORM()) return V(x.x + a * (y.x - x.x), x.y + a * (y.y - x.y), x.z + a * (y.z - x.z));
    return V(fmaf(y.x, a, x.x * ia), fmaf(y.y, a, x.y * ia), fmaf(y.z, a, x.z * ia));
}

static inline float f_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#ifdef PT_ORACLE_PERTURB
#undef fmaf
#endif

/* sin and cos of a (radians), |a| small (the integrator only passes [0, 2*pi]).  Cody-Waite reduction by pi/2
 * with two fused steps, then the classic single-precision minimax polynomials on [-pi/4, pi/4]. */
#ifdef PT_ORACLE_PERTURB
/* base variant bit 128: sin, cos, exp, pow the way llvmpipe's gallivm evaluates them (Mesa, src/gallium/auxiliary/gallivm/lp_bld_arit.c —
 * a third-party dependency of the REFERENCE'S TEST RIG, absent from /root/reference; restated from its published algorithm and pinned by
 * black-box probing: tests/test_arithmetic_choices.py runs the GLSL built-ins on the live llvmpipe through oracle/_ref/glsl_runner and
 * finds these functions BIT-IDENTICAL on 65,536 arguments each).  sin / cos: the Cephes-derived SSE routine (reduction by pi/4 in three
 * steps, j = (int(|x| 4/pi) + 1) & ~1, two minimax polynomials, multiply-adds fused).  exp2: floor / fraction split, degree-5 polynomial of
 * the fraction evaluated as even and odd halves with fused multiply-adds, scaled by 2^floor.  log2: exponent + y P(y^2) with
 * y = (m - 1) / (m + 1), degree-4 P, same evaluation.  exp(x) = exp2(x log2 e), pow(x, y) = exp2(log2(x) y): negative base NaN, zero 0. */
static int g_base_llvm_math = 0;
static inline float ll_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static float ll_poly(float x, const float *co, int n) /* lp_build_polynomial: even and odd powers separately, then odd * x + even */
{
    const float x2 = x * x;
    float even = 0.0f, odd = 0.0f;
    int haveEven = 0, haveOdd = 0;
    for (int i = n; i--;) {
        if ((i & 1) == 0) { even = haveEven ? ll_fma(x2, even, co[i]) : co[i]; haveEven = 1; }
        else { odd = haveOdd ? ll_fma(x2, odd, co[i]) : co[i]; haveOdd = 1; }
    }
    return haveOdd ? ll_fma(odd, x, even) : even;
}
static float ll_exp2(float x)
{
    static const float co[6] = { 1.000000000000000000000f, 0.693153073200168932794f, 0.240153617044375388211f, 0.0558263180532956664775f,
                                 0.00898934009049466391101f, 0.00187757667519147912699f };
    if (x != x) return x;
    if (x > 128.0f) x = 128.0f;
    if (x < -126.99999f) x = -126.99999f;
    const float ip = floorf(x), fp = x - ip;
    return f_unbits((uint32_t)((int)ip + 127) << 23) * ll_poly(fp, co, 6);
}
static float ll_log2(float x)
{
    static const float co[5] = { 2.88539009343309178325f, 0.961791550404184197881f, 0.577440339438736392009f, 0.403343858251329912514f,
                                 0.406718052498846252698f };
    if (x != x || x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (isinf(x)) return x;
    const uint32_t i = f_bits(x);
    const float e = (float)((int)((i >> 23) & 0xffu) - 127);
    const float m = f_unbits((i & 0x007fffffu) | 0x3f800000u);
    const float y = (m - 1.0f) / (m + 1.0f);
    return ll_fma(y, ll_poly(y * y, co, 5), e);
}
static float ll_exp(float x) { return ll_exp2(x * 1.44269504088896340735992f); }
static float ll_pow(float x, float y)
{
    if (x != x) return 0.0f; /* (measured: pow(NaN, 5.0) = 0 on llvmpipe) */
    if (x == 0.0f) return 0.0f;
    return ll_exp2(ll_log2(x) * y);
}
static float ll_sin_or_cos(float a, int want_cos)
{
    const float x_abs = fabsf(a);
    int j = (int)(x_abs * 1.27323954473516f);
    j = (j + 1) & ~1;
    const float y = (float)j;
    const int j2 = want_cos ? j - 2 : j;
    const uint32_t sign = want_cos ? ((~(uint32_t)j2 & 4u) << 29) : ((((uint32_t)j2 & 4u) << 29) ^ (f_bits(a) & 0x80000000u));
    float x = ll_fma(y, -0.78515625f, x_abs);
    x = ll_fma(y, -2.4187564849853515625e-4f, x);
    x = ll_fma(y, -3.77489497744594108e-8f, x);
    const float z = x * x;
    float c = ll_fma(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    c = ll_fma(c, z, 4.166664568298827E-002f);
    c = c * z; c = c * z;
    c = ll_fma(z, -0.5f, c); c = c + 1.0f;
    float sv = ll_fma(z, -1.9515295891E-4f, 8.3321608736E-3f);
    sv = ll_fma(sv, z, -1.6666654611E-1f);
    sv = sv * z;
    sv = ll_fma(sv, x, x);
    return f_unbits(f_bits((j2 & 2) == 0 ? sv : c) ^ sign);
}
/* (exported for the probe test: the four functions on arrays) */
#endif
static void f_sincos(float a, float *sn, float *cs)
{
#ifdef PT_ORACLE_PERTURB
    if (g_base_llvm_math) {
        *sn = perturbed(3, ll_sin_or_cos(a, 0));
        *cs = perturbed(4, ll_sin_or_cos(a, 1));
        return;
    }
#endif
    float k = rintf(a * 0.636619772f);
    float r = fmaf(k, -1.57079637050628662109375f, a);
    r = fmaf(k, 4.37113900018624283e-8f, r);
    float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = fmaf(ps * z, r, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float c = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    int q = (int)k & 3;
    float s_out = (q & 1) ? c : s;
    float c_out = (q & 1) ? s : c;
    if (q == 1 || q == 2) c_out = -c_out;
    if (q >= 2) s_out = -s_out;
    *sn = perturbed(3, s_out);
    *cs = perturbed(4, c_out);
}

/* e^x.  n = rint(x*log2 e), r = x - n*ln2 (two fused steps), degree-6 polynomial, 2^n applied as two exact
 * power-of-two factors so that denormal results are rounded once. */
static float f_exp(float x)
{
#ifdef PT_ORACLE_PERTURB
    if (g_base_llvm_math) return perturbed(5, ll_exp(x));
#endif
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -104.0f) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.428606765330187045e-06f, r);
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, r * r, r) + 1.0f;
    int ni = (int)n;
    int n1 = ni >> 1, n2 = ni - n1;
    y = y * f_from_bits((uint32_t)(n1 + 127) << 23);
    return perturbed(5, y * f_from_bits((uint32_t)(n2 + 127) << 23));
}

static inline float f_pow5(float x) /* (GLSL: pow(x, 5.0) — llvmpipe's is ~22 ulp off) */
{
#ifdef PT_ORACLE_PERTURB
    /* pow(x, y) is undefined for x < 0 (GLSL 4.60 section 8.2); llvmpipe's exp2(y log2 x) is NaN.  Mode 2: also for a base within four
       ulps of 1 - cos = 0 — whether 1 - dot(-d, n) of two unit vectors comes out as +-1e-7 or 0 is the last bit of the dot product */
    if (g_base_llvm_math && g_ens_seed == 0) { if (x < 0.0f) tl_ub = 1; return perturbed(6, ll_pow(x, 5.0f)); }
    if (x < 4.8e-7f) tl_ub = 1;
    if (g_ens_seed != 0) { /* an ensemble member: a negative base is NaN for two members in three; a base within four ulps of zero is one
                              whose sign the member's own last bits decide — NaN for about half of such calls (by call, not by value:
                              the same 1 - cos comes out of different dot products) */
        if (x < 0.0f ? g_pow_neg_nan != 0 : (x < 4.8e-7f && (ens_hash(60u, (uint32_t)tl_call_n[6]) >> 31))) { tl_call_n[6]++; return NAN; }
    } else
    if (g_pow_neg_nan && (x < 0.0f || (g_pow_neg_nan == 2 && x < 4.8e-7f))) return NAN;
#endif
    float x2 = x * x;
    return perturbed(6, x * (x2 * x2));
}
#ifdef PT_ORACLE_PERTURB
#define fmaf(a, b, c) wit_fma((a), (b), (c))
#endif

/* ------------------------------------------------------------------ scene blob accessors (std140, compute.glsl:13-42,66-70) */
#define SPHERE_STRIDE 20  /* floats: 80 B  */
#define CUBOID_STRIDE 24  /* floats: 96 B  */
#define CUBOIDS_OFFSET 5120 /* floats: 20480 B = 256 * 80 */

typedef struct {
    v3 albedo;   float specularChance;
    v3 emissiv;  float specularRoughness;
    v3 absorbance; float refractionChance;
    float refractionRoughness, ior;
} Material;

static Material load_material(const float *m)
{
    Material r;
    r.albedo = V(m[0], m[1], m[2]);      r.specularChance = m[3];
    r.emissiv = V(m[4], m[5], m[6]);     r.specularRoughness = m[7];
    r.absorbance = V(m[8], m[9], m[10]); r.refractionChance = m[11];
    r.refractionRoughness = m[12];       r.ior = m[13];
    return r;
}

typedef struct {
    /* BasicDataUBO, compute.glsl:59-64: float[4c+r] = element (row r, col c) of the GLSL matrix */
    float invProj[16], invView[16];
    v3 viewPos;
    const float *objects; /* 6656 floats */
    int numSpheres, numCuboids; /* loops are `int i < float n` (compute.glsl:231,244): i < n  <=>  i < ceil(n) */
    int rayDepth, spp;
    float focalLength, apertureDiameter;
    int width, height;
    int envSize, envFormat; /* 0 = RGBA32F (float[6][S][S][4]), 1 = SRGB8_A8 (uint8[6][S][S][4]) */
    const void *env;
    float srgbLut[256];
} Ctx;

typedef struct { uint64_t samples, bounces, sphereTests, cuboidTests, envLookups, rngDraws; } Stats;

/* ---- decision margins (-DPT_ORACLE_MARGINS, _build/libpt_oracle_margins.so; tests/test_decision_margins.py).
 * The integrator BRANCHES on computed floats: object acceptance (compute.glsl:234,247 with :269 `discriminant < 0`, :293 `t1 <= t2`,
 * `t2 > 0`, `t1 < T`, and GetSmallestPositive's `t1 < 0`, :347-350), lobe selection (:201, :208), refract's `k < 0`, Russian roulette
 * (:169) and the cuboid normal's step() (:322-332).  Two conforming evaluations of the same GLSL differ in the last bits of their
 * floats — GLSL leaves the precision of /, sqrt, inversesqrt, sin, cos, exp implementation-defined — so in a small fraction of pixels
 * one of these comparisons comes out the other way and the path is a different path.  And a path tracer AMPLIFIES: a direction that
 * is off by delta moves the next hit point by T * delta, the normal of a sphere of radius r there by T * delta / r, and the bounce
 * doubles that, so after a few bounces on curved surfaces last-bit differences are percent-level differences.
 *
 * This variant carries, next to every path, a first-order bound of its own error PER UNIT OF RELATIVE ERROR eps of the arithmetic's
 * primitives: position error dp (world units / eps), direction error dd (1 / eps), relative throughput error dthr — started at the
 * camera, propagated through every intersection, normal and BSDF lobe — and records per pixel and frame
 *   margin = the smallest eps at which ONE of the path's comparisons would come out the other way: |a - b| / (error of a - b per
 *            unit eps), over every comparison that could change the result (single-flip analysis, see trace_margins);
 *   cont   = the absolute colour error per unit eps that the pixel suffers WITHOUT any flip (environment gradient x direction error,
 *            radiance x throughput error).
 * The image it renders is bit for bit the plain oracle's (tested).  With that, layer-2 parity is a per-pixel statement: a pixel that
 * differs from the reference's own output by more than the band must have margin < TAU_FLIP or band / cont < TAU_FLIP (it is
 * sensitive to errors of the size conforming implementations differ by), and every pixel that needs more than TAU_SAFE agrees. */
#ifdef PT_ORACLE_MARGINS
#define ERR_FRESH 4.0f /* error a freshly computed quantity carries, in units of eps x its magnitude (a handful of roundings) */
static __thread float tl_margin = INFINITY, tl_cont = 0.0f;
static __thread float tl_dp, tl_dd, tl_dthr;  /* the current ray's error bounds per unit eps (see above) */
static __thread float tl_hit_dT, tl_hit_r;    /* set by ray_trace for the accepted hit: error of T per unit eps; sphere radius (0: cuboid) */
static __thread float tl_dcos, tl_refr_k;     /* error of dot(direction, normal) at the current hit; refract()'s k of the current bounce */
static __thread int tl_lobe;                  /* lobe the current bounce took: 0 diffuse, 1 specular, 2 refractive */
static inline void margin_eps(float diff, float err)
{
    float m = fabsf(diff) / fmaxf(err, 1e-30f);
    if (m < tl_margin) tl_margin = m; /* (NaN: not smaller, ignored — a NaN comparison is false on every implementation) */
}
static inline float fin0(float x) { return fabsf(x) < FLOAT_MAX ? fabsf(x) : 0.0f; }
#endif

/* ------------------------------------------------------------------ RNG (compute.glsl:334-344) */
static inline uint32_t pcg_hash(uint32_t *seed)
{
    *seed = *seed * 747796405u + 2891336453u;
    uint32_t word = ((*seed >> ((*seed >> 28u) + 4u)) ^ *seed) * 277803737u;
    return (word >> 22u) ^ word;
}
static inline float rand01(uint32_t *seed) { return (float)pcg_hash(seed) * 2.3283064365386962890625e-10f; /* / 2^32, exact */ }

/* ------------------------------------------------------------------ environment lookup (compute.glsl:177)
 * texture(samplerCube, dir) from a compute stage: no derivatives -> LOD 0 -> MAG filter = LINEAR
 * (MainWindow.cs:178, AtmosphericScatterer.cs:68), GL_TEXTURE_CUBE_MAP_SEAMLESS on (MainWindow.cs:168).
 * Face selection / (s,t) mapping: OpenGL 4.5 core spec, table 8.19; ties go Z, then X, then Y (llvmpipe).
 * Seamless filtering: taps that fall off a face edge are fetched from the adjacent face; at a cube corner the
 * tap that falls off two edges has no texel and is replaced by the average of the other three.
 * SRGB8_A8 texels are linearised before filtering (GL 4.5 section 8.24). */
typedef struct { float r, g, b; } rgb;

static rgb env_texel(const Ctx *c, int face, int x, int y)
{
    size_t idx = (((size_t)face * c->envSize + (size_t)y) * c->envSize + (size_t)x) * 4;
    rgb o;
    if (c->envFormat == 0) {
        const float *p = (const float *)c->env + idx;
        o.r = p[0]; o.g = p[1]; o.b = p[2];
    } else {
        const uint8_t *p = (const uint8_t *)c->env + idx;
        o.r = c->srgbLut[p[0]]; o.g = c->srgbLut[p[1]]; o.b = c->srgbLut[p[2]];
    }
    return o;
}

/* direction of the point (s,t) in [-1,1]^2 on `face` (inverse of table 8.19), not normalised */
static void face_to_dir(int face, float sc, float tc, float *x, float *y, float *z)
{
    switch (face) {
    case 0: *x = 1.0f;  *y = -tc;  *z = -sc;  break;
    case 1: *x = -1.0f; *y = -tc;  *z = sc;   break;
    case 2: *x = sc;    *y = 1.0f; *z = tc;   break;
    case 3: *x = sc;    *y = -1.0f; *z = -tc; break;
    case 4: *x = sc;    *y = -tc;  *z = 1.0f; break;
    default: *x = -sc;  *y = -tc;  *z = -1.0f; break;
    }
}

static void dir_to_face(float x, float y, float z, int *face, float *sc, float *tc, float *ma)
{
    float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    if (az >= f_max(ax, ay)) { *face = z < 0.0f ? 5 : 4; *ma = az; *sc = z < 0.0f ? -x : x; *tc = -y; }
    else if (ax >= ay)       { *face = x < 0.0f ? 1 : 0; *ma = ax; *sc = x < 0.0f ? z : -z; *tc = -y; }
    else                     { *face = y < 0.0f ? 3 : 2; *ma = ay; *sc = x; *tc = y < 0.0f ? -z : z; }
}

/* integer texel (ix,iy) possibly one step outside `face` in ONE direction -> texel on the neighbouring face */
static rgb env_texel_wrapped(const Ctx *c, int face, int ix, int iy)
{
    int S = c->envSize;
    if (ix >= 0 && ix < S && iy >= 0 && iy < S) return env_texel(c, face, ix, iy);
    /* texel centre in face coordinates, then re-project through 3D onto the neighbour face */
    float fs = (float)S;
    float rfs = 1.0f / fs; /* uniform */
    float sc = ((float)ix + 0.5f) * rfs * 2.0f - 1.0f;
    float tc = ((float)iy + 0.5f) * rfs * 2.0f - 1.0f;
    float x, y, z, ma, nsc, ntc;
    int nface;
    face_to_dir(face, sc, tc, &x, &y, &z);
    /* push the major axis below the overflowing one so the neighbour wins the selection */
    dir_to_face(x, y, z, &nface, &nsc, &ntc, &ma);
    float rma = f_rcp(ma);
    float u = (nsc * rma * 0.5f + 0.5f) * fs;
    float v = (ntc * rma * 0.5f + 0.5f) * fs;
    int nx = (int)floorf(u), ny = (int)floorf(v);
    if (nx < 0) nx = 0; if (nx > S - 1) nx = S - 1;
    if (ny < 0) ny = 0; if (ny > S - 1) ny = S - 1;
    return env_texel(c, nface, nx, ny);
}

static rgb sample_env(const Ctx *c, v3 d)
{
#ifdef PT_ORACLE_MARK_NAN_ENV /* diagnostic build only (oracle/Makefile): flag the paths that end in texture(env, NaN direction) */
    if (d.x != d.x || d.y != d.y || d.z != d.z) { rgb mark = { 1000.0f, 1000.0f, 1000.0f }; return mark; }
#endif
#ifdef PT_ORACLE_PERTURB
    if (d.x != d.x || d.y != d.y || d.z != d.z) { /* texture(env, NaN): undefined in GL (see pto_witness_search) */
        tl_nan_env = 1;
        if (g_nan_env_set) { rgb o_ = { g_nan_env[0], g_nan_env[1], g_nan_env[2] }; return o_; }
    }
#endif
    int S = c->envSize, face;
    float sc, tc, ma;
    dir_to_face(d.x, d.y, d.z, &face, &sc, &tc, &ma);
    float ima = 0.5f * f_rcp(ma);
    float fs = (float)S;
    float u = fmaf(sc, ima, 0.5f) * fs - 0.5f;
    float v = fmaf(tc, ima, 0.5f) * fs - 0.5f;
    /* NaN / inf coordinates (NaN ray directions, compute.glsl:211 with total internal reflection):
       clamp so that the integer conversion below is defined identically on CPU and GPU */
    u = f_min(f_max(u, -1.0f), fs);
    v = f_min(f_max(v, -1.0f), fs);
    float fu = floorf(u), fv = floorf(v);
    float wu = u - fu, wv = v - fv;
    int x0 = (int)fu, y0 = (int)fv, x1 = x0 + 1, y1 = y0 + 1;
    int offx0 = x0 < 0, offx1 = x1 >= S, offy0 = y0 < 0, offy1 = y1 >= S;
    float w00 = (1.0f - wu) * (1.0f - wv), w10 = wu * (1.0f - wv), w01 = (1.0f - wu) * wv, w11 = wu * wv;
    int miss00 = offx0 && offy0, miss10 = offx1 && offy0, miss01 = offx0 && offy1, miss11 = offx1 && offy1;
    rgb t00 = { 0, 0, 0 }, t10 = t00, t01 = t00, t11 = t00;
    if (!miss00) t00 = env_texel_wrapped(c, face, x0, y0);
    if (!miss10) t10 = env_texel_wrapped(c, face, x1, y0);
    if (!miss01) t01 = env_texel_wrapped(c, face, x0, y1);
    if (!miss11) t11 = env_texel_wrapped(c, face, x1, y1);
    if (miss00 || miss10 || miss01 || miss11) {
        /* cube corner: the tap that fell off two edges has no texel; it is replaced by the average of the other
           three, i.e. its bilinear weight is shared equally among them (llvmpipe behaviour — pinned by the
           tiny-cube fixtures in tests/golden; GL 4.5 section 8.14.2 recommends exactly this average) */
        float a = (miss00 ? w00 : miss10 ? w10 : miss01 ? w01 : w11) * 0.333333343f;
        w00 = miss00 ? 0.0f : w00 + a; w10 = miss10 ? 0.0f : w10 + a;
        w01 = miss01 ? 0.0f : w01 + a; w11 = miss11 ? 0.0f : w11 + a;
    }
    rgb o;
#ifdef PT_ORACLE_PERTURB
    if (g_base_sampler_lerp && !(miss00 || miss10 || miss01 || miss11)) { /* base variant bit 512: two nested lerps a + w (b - a), x first (llvmpipe's filter) */
#define LERP_(a_, b_, w_) __builtin_fmaf((w_), (b_) - (a_), (a_)) /* (lp_build_lerp: a multiply-add of the sampler's own code, fused like the built-ins' polynomials) */
        o.r = LERP_(LERP_(t00.r, t10.r, wu), LERP_(t01.r, t11.r, wu), wv);
        o.g = LERP_(LERP_(t00.g, t10.g, wu), LERP_(t01.g, t11.g, wu), wv);
        o.b = LERP_(LERP_(t00.b, t10.b, wu), LERP_(t01.b, t11.b, wu), wv);
#undef LERP_
        return o;
    }
#endif
    o.r = fmaf(t11.r, w11, fmaf(t01.r, w01, fmaf(t10.r, w10, t00.r * w00)));
    o.g = fmaf(t11.g, w11, fmaf(t01.g, w01, fmaf(t10.g, w10, t00.g * w00)));
    o.b = fmaf(t11.b, w11, fmaf(t01.b, w01, fmaf(t10.b, w10, t00.b * w00)));
    return o;
}

/* ------------------------------------------------------------------ intersections */
/* compute.glsl:261-277 RaySphereIntersect */
static int ray_sphere(v3 o, v3 d, v3 pos, float radius, float *t1, float *t2)
{
    *t1 = *t2 = FLOAT_MAX;
    v3 oc = v_sub(o, pos);
    float b = v_dot(d, oc);
    float c = fmaf(-radius, radius, v_dot(oc, oc));
    float disc = fmaf(b, b, -c);
    if (DECIDE(disc < 0.0f, disc, f_max(b * b, fabsf(c)))) return 0;
#ifdef PT_ORACLE_PERTURB
    if (disc < 0.0f) disc = 0.0f; /* (the inverted decision: an implementation whose discriminant came out >= 0 grazes the sphere) */
#endif
    float s = pt_sqrt(disc);
    *t1 = -b - s;
    *t2 = -b + s;
    return *t1 <= *t2;
}

/* compute.glsl:280-294 RayCuboidIntersect */
static int ray_cuboid(v3 o, v3 d, v3 invd, v3 mn, v3 mx, float *t1, float *t2)
{
#ifdef PT_SLAB_TRUE_DIVISION
    (void)invd;
    v3 t0s = V((mn.x - o.x) / d.x, (mn.y - o.y) / d.y, (mn.z - o.z) / d.z);
    v3 t1s = V((mx.x - o.x) / d.x, (mx.y - o.y) / d.y, (mx.z - o.z) / d.z);
#else
    v3 t0s = v_mul(v_sub(mn, o), invd);
    v3 t1s = v_mul(v_sub(mx, o), invd);
#ifdef PT_ORACLE_PERTURB
    if (g_base_truediv) {
        t0s = V((mn.x - o.x) / d.x, (mn.y - o.y) / d.y, (mn.z - o.z) / d.z);
        t1s = V((mx.x - o.x) / d.x, (mx.y - o.y) / d.y, (mx.z - o.z) / d.z);
    }
#else
    (void)d;
#endif
#endif
    v3 sm = V(f_min(t0s.x, t1s.x), f_min(t0s.y, t1s.y), f_min(t0s.z, t1s.z));
    v3 bg = V(f_max(t0s.x, t1s.x), f_max(t0s.y, t1s.y), f_max(t0s.z, t1s.z));
    *t1 = f_max(FLOAT_MIN, f_max(sm.x, f_max(sm.y, sm.z)));
    *t2 = f_min(FLOAT_MAX, f_min(bg.x, f_min(bg.y, bg.z)));
    return DECIDE(*t1 <= *t2, *t2 - *t1, f_max(fabsf(*t1), fabsf(*t2)));
}

static inline float f_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline float f_step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }

/* compute.glsl:322-332 GetNormal(Cuboid) */
static v3 cuboid_normal(v3 mn, v3 mx, v3 p)
{
    v3 half = v_scale(v_sub(mx, mn), 0.5f);
    v3 cs = v_sub(p, v_scale(v_add(mx, mn), 0.5f));
    v3 n;
#ifdef PT_ORACLE_PERTURB /* (step(edge, x) = x < edge ? 0 : 1 with x = EPSILON: the same comparison, as a DECIDE site) */
#define STEP_EPS(c_, h_) (DECIDE(EPSILON < fabsf(fabsf(c_) - (h_)), fabsf(fabsf(c_) - (h_)) - EPSILON, f_max(f_max(fabsf(c_), (h_)), 1.0f)) ? 0.0f : 1.0f)
    n.x = f_sign(cs.x) * STEP_EPS(cs.x, half.x);
    n.y = f_sign(cs.y) * STEP_EPS(cs.y, half.y);
    n.z = f_sign(cs.z) * STEP_EPS(cs.z, half.z);
#undef STEP_EPS
#else
    n.x = f_sign(cs.x) * f_step(fabsf(fabsf(cs.x) - half.x), EPSILON);
    n.y = f_sign(cs.y) * f_step(fabsf(fabsf(cs.y) - half.y), EPSILON);
    n.z = f_sign(cs.z) * f_step(fabsf(fabsf(cs.z) - half.z), EPSILON);
#endif
#ifdef PT_ORACLE_MARGINS
    { /* compute.glsl:322-332: step(EPSILON, | |p - centre| - halfsize |) per axis decides which faces the normal sees (a hit point within
         EPSILON of an edge); error = the hit point's + fresh rounding of the coordinates involved */
        float sc_ = f_max(f_max(fabsf(p.x), fabsf(p.y)), fabsf(p.z));
        sc_ = f_max(sc_, f_max(f_max(fabsf(mx.x), fabsf(mx.y)), fabsf(mx.z)));
        sc_ = f_max(sc_, f_max(f_max(fabsf(mn.x), fabsf(mn.y)), fabsf(mn.z)));
        const float e_ = tl_dp + ERR_FRESH * sc_; /* (the caller has put the hit point's error into tl_dp) */
        margin_eps(fabsf(fabsf(cs.x) - half.x) - EPSILON, e_);
        margin_eps(fabsf(fabsf(cs.y) - half.y) - EPSILON, e_);
        margin_eps(fabsf(fabsf(cs.z) - half.z) - EPSILON, e_);
    }
#endif
    return v_normalize(n);
}

typedef struct {
    float T; int fromInside; v3 nearHitPos, normal; Material m;
} HitInfo;

#ifdef PT_ORACLE_MARGINS
/* Margins of RayTrace's acceptance chains (compute.glsl:234,247: `Intersect(...) && t2 > 0 && t1 < T`, with :269 `discriminant < 0`,
 * :293 `t1 <= t2`, and GetSmallestPositive's `t1 < 0`, :347-350), as a SINGLE-FLIP analysis: which one comparison, coming out the
 * other way, changes the object the ray hits or the distance?
 *   - the winner's own chain, against the T it was compared with;
 *   - any other object of which exactly ONE condition fails when T is the final distance: that flip would make it the hit (this
 *     includes the second-nearest candidate's `t1 < T`);
 *   - when some accepted object CONTAINS the origin the reference's rule depends on the visiting order (the entry-distance quirk); then
 *     every object's chain is counted against the running T at its turn (a superset of the relevant flips).
 * A negative discriminant is treated as a grazing hit (square root 0).  Errors per unit eps, with e(t) = dp + |t| dd the ray's
 * sideways displacement at parameter t: sphere — discriminant r^2 - dperp^2: 2 dperp e(|b|) + fresh; b: |oc| dd + dp + fresh; a root
 * -b -+ sqrt(disc): error of b + error of disc / (2 sqrt(disc)); cuboid — a slab distance (m - o) / d along an axis:
 * (dp + fresh) / |d| + |t| dd / |d|, the worst axis that is not parallel to the ray. */
typedef struct { int d1, d2, d3, isSphere; float u1, u2, du, disc, ddisc; } Chain;
static Chain chain_sphere(v3 o, v3 d, const float *s)
{
    Chain k;
    v3 oc = v_sub(o, V(s[0], s[1], s[2]));
    float b = v_dot(d, oc), oo = v_dot(oc, oc), c = fmaf(-s[3], s[3], oo);
    k.disc = fmaf(b, b, -c);
    float sq = pt_sqrt(f_max(k.disc, 0.0f));
    k.u1 = -b - sq; k.u2 = -b + sq;
    const float e = tl_dp + fabsf(b) * tl_dd;
    const float dperp = sqrtf(f_max(oo - b * b, 0.0f));
    k.ddisc = 2.0f * dperp * e + ERR_FRESH * f_max(b * b, f_max(oo, s[3] * s[3]));
    const float db = sqrtf(oo) * tl_dd + tl_dp + ERR_FRESH * fabsf(b);
    k.du = db + k.ddisc / (2.0f * f_max(sq, 1e-20f)) + ERR_FRESH * f_max(fabsf(b), sq);
    k.d1 = !(k.disc < 0.0f); k.d2 = 1; k.d3 = k.u2 > 0.0f; k.isSphere = 1;
    return k;
}
static Chain chain_cuboid(v3 o, v3 d, v3 invd, const float *q)
{
    Chain k;
    (void)d;
    k.d2 = ray_cuboid(o, d, invd, V(q[0], q[1], q[2]), V(q[4], q[5], q[6]), &k.u1, &k.u2);
    const float tmax = f_max(fin0(k.u1), fin0(k.u2));
    const float oa[3] = { o.x, o.y, o.z }, ia[3] = { invd.x, invd.y, invd.z };
    float du = 0.0f;
    for (int a = 0; a < 3; a++) {
        const float iv = fabsf(ia[a]);
        if (!(iv < 1e18f)) continue; /* (an axis the ray is parallel to has infinite slab distances that never decide anything) */
        const float m = f_max(f_max(fabsf(q[a]), fabsf(q[4 + a])), fabsf(oa[a]));
        du = f_max(du, (tl_dp + ERR_FRESH * m) * iv + tmax * (tl_dd * iv + ERR_FRESH));
    }
    k.du = du; k.d1 = 1; k.d3 = k.u2 > 0.0f; k.isSphere = 0; k.disc = 1.0f; k.ddisc = 1.0f;
    return k;
}
/* the chain's comparisons against distance T (error dT): asWinner = all of them (they all hold), else the single failing one */
static void chain_margins(const Chain *k, float T, float dT, int asWinner)
{
    const int d4 = k->u1 < T;
    const int fails = !k->d1 + !k->d2 + !k->d3 + !d4;
    if (asWinner ? fails != 0 : fails != 1) return;
    if (k->isSphere && (asWinner || !k->d1)) margin_eps(k->disc, k->ddisc);
    if (!k->isSphere && (asWinner || !k->d2)) margin_eps(k->u2 - k->u1, 2.0f * k->du);
    if (asWinner || !k->d3) margin_eps(k->u2, k->du);
    if ((asWinner || !d4) && T != FLOAT_MAX) margin_eps(k->u1 - T, k->du + dT);
    if (asWinner) margin_eps(k->u1, k->du); /* GetSmallestPositive: entry or exit distance */
}
static void trace_margins(const Ctx *c, v3 o, v3 d, v3 invd, int winner, int prevWinner, float Tfinal, float TbeforeWinner, int anyInside)
{
    const float *ob = c->objects;
    Chain ch[320];
    const int ns = c->numSpheres, nc = c->numCuboids;
    for (int i = 0; i < ns; i++) ch[i] = chain_sphere(o, d, ob + (size_t)i * SPHERE_STRIDE);
    for (int i = 0; i < nc; i++) ch[256 + i] = chain_cuboid(o, d, invd, ob + CUBOIDS_OFFSET + (size_t)i * CUBOID_STRIDE);
    tl_hit_dT = winner >= 0 ? ch[winner].du : 0.0f;
    if (anyInside) {
        float T = FLOAT_MAX, dT = 0.0f; /* running distance and its error: the order-dependent case */
        for (int pass = 0; pass < 2; pass++)
            for (int i = 0; i < (pass ? nc : ns); i++) {
                const Chain *k = &ch[pass ? 256 + i : i];
                const int d4 = k->u1 < T, fails = !k->d1 + !k->d2 + !k->d3 + !d4;
                if (fails == 0) chain_margins(k, T, dT, 1);
                else if (fails == 1) chain_margins(k, T, dT, 0);
                if (fails == 0) { T = k->u1 < 0.0f ? k->u2 : k->u1; dT = k->du; }
            }
        return;
    }
    const float dTfinal = winner >= 0 ? ch[winner].du : 0.0f, dTbefore = prevWinner >= 0 ? ch[prevWinner].du : 0.0f;
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < (pass ? nc : ns); i++) {
            const int id = pass ? 256 + i : i;
            if (id == winner) chain_margins(&ch[id], TbeforeWinner, dTbefore, 1);
            else chain_margins(&ch[id], Tfinal, dTfinal, 0);
        }
}
#endif

/* compute.glsl:226-258 RayTrace.  The acceptance test uses the ENTRY distance t1 against the stored
 * GetSmallestPositive(t1,t2) (compute.glsl:234,247,347-350): an object that contains the origin (t1<0) always
 * replaces the current hit.  Objects are visited in reference order.  Material/normal are evaluated once for
 * the surviving candidate (the reference evaluates them per accepted candidate and overwrites). */
static int ray_trace(const Ctx *c, v3 o, v3 d, HitInfo *h, Stats *st)
{
    float T = FLOAT_MAX, t1, t2, wt2 = 0.0f;
    int winner = -1;
    const float *ob = c->objects;
#ifdef PT_ORACLE_MARGINS
    float Tbefore = FLOAT_MAX; /* the T the final winner was compared with ... */
    int prevWinner = -1;       /* ... and the object that had set it */
    int anyInside = 0;         /* an accepted object contained the origin */
#define NOTE_ACCEPT() do { Tbefore = T; prevWinner = winner; anyInside |= t1 < 0.0f; } while (0)
#else
#define NOTE_ACCEPT() ((void)0)
#endif
    for (int i = 0; i < c->numSpheres; i++) {
        const float *s = ob + (size_t)i * SPHERE_STRIDE;
        if (ray_sphere(o, d, V(s[0], s[1], s[2]), s[3], &t1, &t2) && DECIDE(t2 > 0.0f, t2, f_max(fabsf(t1), fabsf(t2))) && DECIDE(t1 < T, t1 - T, f_max(fabsf(t1), fabsf(T)))) {
            NOTE_ACCEPT();
            T = DECIDE(t1 < 0.0f, t1, f_max(fabsf(t1), fabsf(t2))) ? t2 : t1;
            wt2 = t2;
            winner = i;
        }
    }
    v3 invd = V(f_rcp(d.x), f_rcp(d.y), f_rcp(d.z));
    for (int i = 0; i < c->numCuboids; i++) {
        const float *q = ob + CUBOIDS_OFFSET + (size_t)i * CUBOID_STRIDE;
        if (ray_cuboid(o, d, invd, V(q[0], q[1], q[2]), V(q[4], q[5], q[6]), &t1, &t2) && DECIDE(t2 > 0.0f, t2, f_max(fabsf(t1), fabsf(t2))) && DECIDE(t1 < T, t1 - T, f_max(fabsf(t1), fabsf(T)))) {
            NOTE_ACCEPT();
            T = DECIDE(t1 < 0.0f, t1, f_max(fabsf(t1), fabsf(t2))) ? t2 : t1;
            wt2 = t2;
            winner = 256 + i;
        }
    }
#undef NOTE_ACCEPT
#ifdef PT_ORACLE_MARGINS
    trace_margins(c, o, d, invd, winner, prevWinner, T, Tbefore, anyInside);
    tl_hit_r = (winner >= 0 && winner < 256) ? fabsf(ob[(size_t)winner * SPHERE_STRIDE + 3]) : 0.0f;
    if (winner >= 0 && T != FLOAT_MAX) tl_dp = tl_dp + T * tl_dd + tl_hit_dT; /* from here on: the error of the hit point (o + d T) */
#endif
    if (st) { st->sphereTests += (uint64_t)c->numSpheres; st->cuboidTests += (uint64_t)c->numCuboids; }
    if (winner < 0 || !(T != FLOAT_MAX)) return 0; /* compute.glsl:257 */
    h->T = T;
    h->fromInside = (T == wt2);
    SIG_NOTE(0x1000 + winner * 2 + h->fromInside);
    h->nearHitPos = v_fma(d, T, o);
    if (winner < 256) {
        const float *s = ob + (size_t)winner * SPHERE_STRIDE;
        h->m = load_material(s + 4);
        v3 pc = v_sub(h->nearHitPos, V(s[0], s[1], s[2])); /* compute.glsl:316-319 GetNormal(Sphere) */
        const float ir = 1.0f / s[3]; /* 1/radius: IEEE quotient, computed once per sphere */
        h->normal = V(QUOT(pc.x, s[3], ir), QUOT(pc.y, s[3], ir), QUOT(pc.z, s[3], ir));
    } else {
        const float *q = ob + CUBOIDS_OFFSET + (size_t)(winner - 256) * CUBOID_STRIDE;
        h->m = load_material(q + 8);
        h->normal = cuboid_normal(V(q[0], q[1], q[2]), V(q[4], q[5], q[6]), h->nearHitPos);
    }
    return 1;
}

/* ------------------------------------------------------------------ sampling */
/* compute.glsl:297-307 */
static v3 cosine_sample_hemisphere(v3 n, uint32_t *seed)
{
    float z = fmaf(rand01(seed), 2.0f, -1.0f);
    float a = rand01(seed) * 2.0f * PI;
    float r = pt_sqrt(fmaf(-z, z, 1.0f));
    float sn, cs;
    f_sincos(a, &sn, &cs);
    return v_normalize(v_add(n, V(r * cs, r * sn, z)));
}

/* compute.glsl:359-364 */
static float fresnel_schlick(float cosTheta, float n1, float n2)
{
    float r0 = QUOT(n1 - n2, n1 + n2, f_rcp(n1 + n2)); /* (compute.glsl:361 divides; the contract multiplies by the reciprocal) */
    r0 *= r0;
    return fmaf(1.0f - r0, f_pow5(1.0f - cosTheta), r0);
}

static v3 f_reflect(v3 i, v3 n) { return v_fma(n, -(2.0f * v_dot(n, i)), i); }

static v3 f_refract(v3 i, v3 n, float eta)
{
    float ni = v_dot(n, i);
    float k = fmaf(-(eta * eta), fmaf(-ni, ni, 1.0f), 1.0f);
#ifdef PT_ORACLE_MARGINS
    /* GLSL refract(): k < 0 = total internal reflection.  k = 1 - eta^2 (1 - (n.i)^2): error 2 eta^2 |n.i| x (error of n.i) + fresh;
       the caller keeps the error of n.i (direction error + normal error) in tl_dthr's neighbour tl_dcos */
    margin_eps(k, 2.0f * eta * eta * fabsf(ni) * tl_dcos + ERR_FRESH * f_max(1.0f, eta * eta));
    tl_refr_k = k;
#endif
    if (DECIDE(k < 0.0f, k, f_max(1.0f, eta * eta))) return V(0.0f, 0.0f, 0.0f);
#ifdef PT_ORACLE_PERTURB
    if (k < 0.0f) k = 0.0f; /* (the inverted decision) */
#endif
    float f = fmaf(eta, ni, pt_sqrt(k));
    return V(fmaf(eta, i.x, -(f * n.x)), fmaf(eta, i.y, -(f * n.y)), fmaf(eta, i.z, -(f * n.z)));
}

/* compute.glsl:184-224 BSDF */
static float bsdf(v3 *ro, v3 *rd, const HitInfo *h, int *isRefractive, uint32_t *seed)
{
    *isRefractive = 0;
    float spec = h->m.specularChance, refr = h->m.refractionChance;
    if (spec > 0.0f) {
        float n1 = h->fromInside ? h->m.ior : 1.0f, n2 = !h->fromInside ? h->m.ior : 1.0f;
        spec = f_mix(spec, 1.0f, fresnel_schlick(v_dot(v_neg(*rd), h->normal), n1, n2));
        float diffuse = 1.0f - spec - refr;
        refr = 1.0f - spec - diffuse;
    }
    v3 diffuseRay = cosine_sample_hemisphere(h->normal, seed);
    float prob;
    float roll = rand01(seed);
#ifdef PT_ORACLE_MARGINS
    /* compute.glsl:201,208.  The roll is exact (integer hash); spec carries the Fresnel term's error when the material is specular:
       F = r0 + (1 - r0)(1 - cos)^5 -> at most 5 x the error of cos.  (A material with neither lobe takes the third branch whatever the roll.) */
    float dspec_ = ERR_FRESH + (h->m.specularChance > 0.0f ? 5.0f * tl_dcos : 0.0f);
    if (spec > 0.0f || refr > 0.0f) {
        margin_eps(spec - roll, dspec_);
        if (!(spec > roll)) margin_eps(spec + refr - roll, dspec_ + ERR_FRESH);
    }
    tl_lobe = spec > roll ? 1 : (spec + refr > roll ? 2 : 0);
    tl_refr_k = 1.0f;
    /* the chosen lobe's probability divides the throughput (compute.glsl:164): its relative error */
    {
        const float prob_ = f_max(tl_lobe == 1 ? spec : tl_lobe == 2 ? refr : 1.0f - spec - refr, EPSILON);
        tl_dthr += (h->m.specularChance > 0.0f ? dspec_ / prob_ : 0.0f) + ERR_FRESH;
    }
#endif
    const int lobeSpec = DECIDE(spec > roll, spec - roll, 1.0f);
    if (lobeSpec) {
        v3 refl = f_reflect(*rd, h->normal);
        *rd = v_normalize(v_mix(refl, diffuseRay, h->m.specularRoughness * h->m.specularRoughness));
        prob = spec;
        SIG_NOTE(0x2001);
    } else if (DECIDE(spec + refr > roll, spec + refr - roll, 1.0f)) {
        v3 rf = f_refract(*rd, h->normal, h->fromInside ? h->m.ior : f_rcp(h->m.ior));
        v3 rough = cosine_sample_hemisphere(v_neg(h->normal), seed);
        *rd = v_normalize(v_mix(rf, rough, h->m.refractionRoughness * h->m.refractionRoughness));
        prob = refr;
        *isRefractive = 1;
        SIG_NOTE(0x2002);
    } else {
        *rd = diffuseRay;
        prob = 1.0f - spec - refr;
        SIG_NOTE(0x2000);
    }
    *ro = v_fma(*rd, EPSILON, h->nearHitPos);
    return f_max(prob, EPSILON);
}

/* compute.glsl:132-182 Radiance */
static v3 radiance(const Ctx *c, v3 ro, v3 rd, uint32_t *seed, Stats *st)
{
    v3 throughput = V(1.0f, 1.0f, 1.0f), rad = V(0.0f, 0.0f, 0.0f);
    HitInfo h;
    for (int i = 0; i < c->rayDepth; i++) {
        if (st) st->bounces++;
        if (ray_trace(c, ro, rd, &h, st)) {
            if (h.fromInside) {
                h.normal = v_neg(h.normal);
                throughput.x *= f_exp(-h.m.absorbance.x * h.T);
                throughput.y *= f_exp(-h.m.absorbance.y * h.T);
                throughput.z *= f_exp(-h.m.absorbance.z * h.T);
            }
#ifdef PT_ORACLE_MARGINS
            /* (ray_trace left the hit point's error in tl_dp and the error of T in tl_hit_dT.)  Normal: a sphere's is (p - c) / r, a
               cuboid's is constant on a face.  Beer's law: exp(-a T).  dot(direction, normal): both errors. */
            const float dn_ = tl_hit_r > 0.0f ? tl_dp / tl_hit_r + ERR_FRESH : ERR_FRESH;
            if (h.fromInside) tl_dthr += f_max(h.m.absorbance.x, f_max(h.m.absorbance.y, h.m.absorbance.z)) * tl_hit_dT + ERR_FRESH;
            tl_dcos = tl_dd + dn_;
#endif
            int isRefractive;
            float prob = bsdf(&ro, &rd, &h, &isRefractive, seed);
#ifdef PT_ORACLE_MARGINS
            /* the new ray: a diffuse direction depends on the normal only; a reflection doubles the normal's error and keeps the
               incoming one; a refraction does the same and blows up towards the critical angle (1 / sqrt(k)); roughness mixes in the
               diffuse direction (bounded by the same).  New origin = hit point + EPSILON x direction. */
            {
                float dd_;
                if (tl_lobe == 0) dd_ = dn_ + ERR_FRESH;
                else if (tl_lobe == 1) dd_ = tl_dd + 2.0f * dn_ + ERR_FRESH;
                else dd_ = (tl_dd + 2.0f * dn_) * (1.0f + 2.0f / sqrtf(f_max(tl_refr_k, 1e-12f))) + ERR_FRESH;
                tl_dd = dd_;
                tl_dp = tl_dp + EPSILON * dd_ + ERR_FRESH * f_max(f_max(fabsf(ro.x), fabsf(ro.y)), f_max(fabsf(ro.z), 1.0f));
                /* emissive hit: radiance += emissiv x throughput (its relative error so far) */
                const float em_ = f_max(h.m.emissiv.x * throughput.x, f_max(h.m.emissiv.y * throughput.y, h.m.emissiv.z * throughput.z));
                tl_cont += em_ * tl_dthr;
            }
#endif
            rad = V(fmaf(h.m.emissiv.x, throughput.x, rad.x), fmaf(h.m.emissiv.y, throughput.y, rad.y),
                    fmaf(h.m.emissiv.z, throughput.z, rad.z));
            if (!isRefractive) throughput = v_mul(throughput, h.m.albedo);
            {
                const float rprob = f_rcp(prob);
                throughput = V(QUOT(throughput.x, prob, rprob), QUOT(throughput.y, prob, rprob), QUOT(throughput.z, prob, rprob));
            }
            float p = f_max(throughput.x, f_max(throughput.y, throughput.z));
#ifdef PT_ORACLE_MARGINS
            {
                uint32_t peek = *seed;
                float roll_ = rand01(&peek);
                if (i + 1 < c->rayDepth) margin_eps(roll_ - p, p * tl_dthr + ERR_FRESH * p); /* compute.glsl:169 (after the last bounce the outcome no longer matters) */
            }
#endif
            const float rr_ = rand01(seed);
            if (DECIDE(rr_ > p, rr_ - p, f_max(p, 1.0f))) { SIG_NOTE(0x3001); break; }
            {
                const float rp = f_rcp(p);
                throughput = V(QUOT(throughput.x, p, rp), QUOT(throughput.y, p, rp), QUOT(throughput.z, p, rp));
            }
        } else {
            rgb e = sample_env(c, rd);
            SIG_NOTE(0x3002);
            if (st) st->envLookups++;
#ifdef PT_ORACLE_MARGINS
            { /* no flip, still an error: the environment's change over the direction's error (finite differences along two tangents,
                 1e-3 rad) and the throughput's relative error — absolute colour error per unit eps, summed over the pixel's paths */
                v3 t1_ = fabsf(rd.x) < 0.9f ? V(1.0f, 0.0f, 0.0f) : V(0.0f, 1.0f, 0.0f);
                v3 ta = v_normalize(V(rd.y * t1_.z - rd.z * t1_.y, rd.z * t1_.x - rd.x * t1_.z, rd.x * t1_.y - rd.y * t1_.x));
                v3 tb = V(rd.y * ta.z - rd.z * ta.y, rd.z * ta.x - rd.x * ta.z, rd.x * ta.y - rd.y * ta.x);
                const float hstep = 1e-3f;
                rgb ea = sample_env(c, v_normalize(v_fma(ta, hstep, rd))), eb = sample_env(c, v_normalize(v_fma(tb, hstep, rd)));
                float worst = 0.0f;
                const float er[3] = { e.r, e.g, e.b }, ear[3] = { ea.r, ea.g, ea.b }, ebr[3] = { eb.r, eb.g, eb.b }, th[3] = { throughput.x, throughput.y, throughput.z };
                for (int ch = 0; ch < 3; ch++) {
                    const float grad = (fabsf(ear[ch] - er[ch]) + fabsf(ebr[ch] - er[ch])) / hstep;
                    const float err = fabsf(th[ch]) * (grad * tl_dd + fabsf(er[ch]) * (tl_dthr + ERR_FRESH));
                    if (err > worst) worst = err;
                }
                tl_cont += worst;
            }
#endif
            rad = V(fmaf(e.r, throughput.x, rad.x), fmaf(e.g, throughput.y, rad.y), fmaf(e.b, throughput.z, rad.z));
            break;
        }
    }
    return rad;
}

/* GLSL mat4 * vec4 with the column-major view of the UBO bytes */
static void mat_vec(const float *m, float x, float y, float z, float w, float *out)
{
#ifdef PT_ORACLE_PERTURB
    if (g_ens_seed != 0 || g_base_matvec != 0) { /* (an ensemble member's own order of the four column terms, per product) */
        uint32_t ux, uy; memcpy(&ux, &x, 4); memcpy(&uy, &y, 4);
        const uint32_t order = g_ens_seed != 0 ? ens_hash(12u + ux, uy) % 4u : (uint32_t)g_base_matvec;
        for (int r = 0; r < 4; r++) {
            const float cx = m[r], cy = m[4 + r], cz = m[8 + r], cw = m[12 + r];
            out[r] = order == 1 ? fmaf(cx, x, fmaf(cy, y, fmaf(cz, z, cw * w)))        /* w first */
                   : order == 2 ? fmaf(cy, y, fmaf(cz, z, fmaf(cx, x, cw * w)))        /* ((w + x) + z) + y: llvmpipe's, found by matching its primary rays bit for bit */
                   : order == 3 ? fmaf(cz, z, fmaf(cw, w, fmaf(cy, y, cx * x)))
                   : fmaf(cw, w, fmaf(cz, z, fmaf(cy, y, cx * x)));
        }
        return;
    }
#endif
    for (int r = 0; r < 4; r++)
        out[r] = fmaf(m[12 + r], w, fmaf(m[8 + r], z, fmaf(m[4 + r], y, m[r] * x)));
}

/* compute.glsl:101-130 main, for one pixel; `last` is the pixel's current accumulation value */
static void shade_pixel(const Ctx *c, int px, int py, int frame, const float *last, float *out, Stats *st)
{
    uint32_t seed = ((uint32_t)px * 1973u + (uint32_t)py * 9277u + (uint32_t)frame * 2699u) | 1u; /* :106 */
    v3 irr = V(0.0f, 0.0f, 0.0f);
#ifdef PT_ORACLE_PERTURB
    tl_dec_n = 0; tl_close_n = 0; tl_nan_env = 0;
    tl_sig = g_sig_alpha ? (f_bits(last[3]) & 0x7FFFFFu) : 0u; /* (chained over the frames of an accumulation) */
    tl_ub = 0;
    memset(tl_call_n, 0, sizeof tl_call_n);
#endif
    for (int s = 0; s < c->spp; s++) {
        float u0 = rand01(&seed), u1 = rand01(&seed); /* :113, x first */
        float ndcx = fmaf(((float)px + u0) * (1.0f / (float)c->width), 2.0f, -1.0f);  /* uniform 1/W, 1/H */
        float ndcy = fmaf(((float)py + u1) * (1.0f / (float)c->height), 2.0f, -1.0f);
#ifdef PT_ORACLE_PERTURB
        if (g_base_truediv) { /* (base variant: the literal / imgResultSize of compute.glsl:114 — llvmpipe divides) */
            ndcx = fmaf(((float)px + u0) / (float)c->width, 2.0f, -1.0f);
            ndcy = fmaf(((float)py + u1) / (float)c->height, 2.0f, -1.0f);
        }
#endif
        /* GetWorldSpaceRay :352-357 */
        float eye[4], wd[4];
        mat_vec(c->invProj, ndcx, ndcy, -1.0f, 0.0f, eye);
        mat_vec(c->invView, eye[0], eye[1], -1.0f, 0.0f, wd);
        v3 dir = v_normalize(V(wd[0], wd[1], wd[2]));
        v3 focal = v_fma(dir, c->focalLength, c->viewPos); /* :117 */
        /* UniformSampleUnitCircle :309-314 */
        float angle = rand01(&seed) * 2.0f * PI;
        float rr = pt_sqrt(rand01(&seed));
        float sn, cs;
        f_sincos(angle, &sn, &cs);
        float half_ap = c->apertureDiameter * 0.5f;
        float ox = half_ap * (cs * rr), oy = half_ap * (sn * rr);
        float org[4];
        mat_vec(c->invView, ox, oy, 0.0f, 1.0f, org); /* :120 */
        v3 ro = V(org[0], org[1], org[2]);
        v3 rd = v_normalize(v_sub(focal, ro));
#ifdef PT_ORACLE_MARGINS
        /* the primary ray's own error: a few roundings of two matrix products, a normalisation, the lens sample's sin / cos */
        tl_dd = 2.0f * ERR_FRESH;
        tl_dp = 2.0f * ERR_FRESH * f_max(f_max(fabsf(ro.x), fabsf(ro.y)), f_max(fabsf(ro.z), 1.0f));
        tl_dthr = 0.0f;
#endif
        if (st) st->samples++;
        irr = v_add(irr, radiance(c, ro, rd, &seed, st));
    }
    irr = v_scale(irr, 1.0f / (float)c->spp); /* :125, uniform reciprocal */
    float w = 1.0f / (float)(frame + 1);                  /* :128 */
    out[0] = f_mix(last[0], irr.x, w);
    out[1] = f_mix(last[1], irr.y, w);
    out[2] = f_mix(last[2], irr.z, w);
    out[3] = 1.0f; /* :129 */
#ifdef PT_ORACLE_PERTURB
    if (tl_ub || tl_nan_env) sig_note(0xDEAD0000u ^ g_ens_seed ^ 0x5bd1e995u); /* undefined behaviour on the way: no two implementations "follow the same path" */
    if (g_sig_alpha) out[3] = f_unbits(0x3F800000u | (tl_sig & 0x7FFFFFu)); /* a float in [1, 2): 23 bits of the signature, survives copies */
#endif
}

/* ------------------------------------------------------------------ public C API (ctypes) */
typedef struct {
    int width, height;         /* FULL image size (NDC and seeds use global coordinates) */
    int numSpheres, numCuboids;
    int rayDepth, spp;
    float focalLength, apertureDiameter;
    int envSize, envFormat;
} PtoParams;

static float srgb_to_linear(int v)
{
    /* GL 4.5 section 8.24, evaluated in double and rounded once: a fixed 256-entry table */
    double cs = v / 255.0;
    double cl = cs <= 0.04045 ? cs / 12.92 : pow((cs + 0.055) / 1.055, 2.4);
    return (float)cl;
}

/* Test-only knob: replace the exact GL sRGB decode table, e.g. with llvmpipe's cubic approximation, so that the
 * llvmpipe pinning test can isolate everything else (tests/test_oracle_vs_reference.py). NULL restores exact. */
static float g_lut_override[256];
static int g_lut_overridden = 0;
PTO_API void pto_set_srgb_lut(const float *lut256)
{
    g_lut_overridden = lut256 != NULL;
    if (lut256) memcpy(g_lut_override, lut256, sizeof g_lut_override);
}
static void fill_lut(float *lut)
{
    for (int i = 0; i < 256; i++) lut[i] = g_lut_overridden ? g_lut_override[i] : srgb_to_linear(i);
}

static void make_ctx(Ctx *c, const PtoParams *p, const float *basic, const float *objects, const void *env)
{
    memcpy(c->invProj, basic, 64);
    memcpy(c->invView, basic + 16, 64);
    c->viewPos = V(basic[32], basic[33], basic[34]);
    c->objects = objects;
    c->numSpheres = p->numSpheres; c->numCuboids = p->numCuboids;
    c->rayDepth = p->rayDepth; c->spp = p->spp;
    c->focalLength = p->focalLength; c->apertureDiameter = p->apertureDiameter;
    c->width = p->width; c->height = p->height;
    c->envSize = p->envSize; c->envFormat = p->envFormat; c->env = env;
    fill_lut(c->srgbLut);
}

/* ---- host threading of one frame (bench.py's cpu_baseline leg and the tests).  A persistent pool of worker threads (created on
 * first use, grown on demand, never more than PTO_MAX_THREADS) renders the frame's rows in DYNAMIC chunks of PTO_CHUNK_ROWS rows
 * taken from one atomic counter: rows near the floor cost ~2x sky rows, so a static split leaves most threads idle behind the
 * slowest one, and creating 256 threads per frame costs milliseconds of a sub-second frame.  Which thread renders which row never
 * changes a pixel (every pixel owns its RNG stream, compute.glsl:106): the image is the same for any thread count (tested). */
#define PTO_MAX_THREADS 256
#define PTO_CHUNK_ROWS 4

typedef struct {
    const Ctx *c; float *image; int y0, rows, frame, wantStats;
    float *margins;              /* optional (PT_ORACLE_MARGINS builds): rows x width x 2: the frame's smallest decision margin per pixel (the
                                    eps that flips a comparison), and its flip-free colour error per unit eps */
    int nextChunk;               /* atomic: next chunk of PTO_CHUNK_ROWS rows to hand out */
    Stats st[PTO_MAX_THREADS];   /* per participant (slot 0 = the calling thread) */
} FrameJob;

static void render_chunks(FrameJob *j, int slot)
{
    const Ctx *c = j->c;
    const int chunks = (j->rows + PTO_CHUNK_ROWS - 1) / PTO_CHUNK_ROWS;
    for (;;) {
        const int k = __atomic_fetch_add(&j->nextChunk, 1, __ATOMIC_RELAXED);
        if (k >= chunks) break;
        const int r1 = (k + 1) * PTO_CHUNK_ROWS < j->rows ? (k + 1) * PTO_CHUNK_ROWS : j->rows;
        for (int r = k * PTO_CHUNK_ROWS; r < r1; r++) {
            const int y = j->y0 + r;
            float *row = j->image + (size_t)r * c->width * 4;
            for (int x = 0; x < c->width; x++) {
                float out[4];
#ifdef PT_ORACLE_MARGINS
                tl_margin = INFINITY;
                tl_cont = 0.0f;
#endif
                shade_pixel(c, x, y, j->frame, row + 4 * x, out, j->wantStats ? &j->st[slot] : NULL);
                memcpy(row + 4 * x, out, 16);
#ifdef PT_ORACLE_MARGINS
                if (j->margins) {
                    j->margins[2 * ((size_t)r * c->width + x)] = tl_margin;
                    j->margins[2 * ((size_t)r * c->width + x) + 1] = tl_cont / (float)c->spp;
                }
#endif
            }
        }
    }
}

static struct {
    pthread_mutex_t mu;
    pthread_cond_t wake, done;
    pthread_t th[PTO_MAX_THREADS];
    int created;        /* workers that exist (worker w has slot w + 1) */
    unsigned long gen;  /* job generation: a worker runs each generation at most once */
    int wanted;         /* workers with slot <= wanted take part in the current generation */
    int running;        /* participants of the current generation that have not finished yet */
    FrameJob *job;
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, { 0 }, 0, 0, 0, 0, NULL };

static void *pool_worker(void *arg)
{
    const int slot = (int)(intptr_t)arg;
    unsigned long seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.gen == seen || slot > g_pool.wanted) {
            if (g_pool.gen != seen && slot > g_pool.wanted) seen = g_pool.gen; /* not invited to this one */
            pthread_cond_wait(&g_pool.wake, &g_pool.mu);
        }
        seen = g_pool.gen;
        FrameJob *j = g_pool.job;
        pthread_mutex_unlock(&g_pool.mu);
        render_chunks(j, slot);
        pthread_mutex_lock(&g_pool.mu);
        if (--g_pool.running == 0) pthread_cond_signal(&g_pool.done);
    }
    return NULL;
}

/* Render one frame into `image` (rows [y0, y0+rows) of the full image, tightly packed RGBA32F, row 0 = y0),
 * accumulating onto its current contents exactly like one PathTracer.Render() call (PathTracer.cs:114-123).
 * stats (optional, 6 x uint64): samples, bounces, sphereTests, cuboidTests, envLookups, reserved.
 * Not re-entrant (one frame at a time per process: the pool is shared); the callers are single-threaded test / bench code. */
static float *g_next_margins = NULL; /* handed to the next pto_render_frame by pto_render_frame_margins (single-threaded callers) */
PTO_API int pto_render_frame(const PtoParams *p, const float *basic144, const float *objects26624, const void *env,
                             float *image, int y0, int rows, int frame, int nthreads, uint64_t *stats)
{
    static pthread_mutex_t serial = PTHREAD_MUTEX_INITIALIZER;
    Ctx c;
    make_ctx(&c, p, basic144, objects26624, env);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > PTO_MAX_THREADS) nthreads = PTO_MAX_THREADS;
    const int chunks = (rows + PTO_CHUNK_ROWS - 1) / PTO_CHUNK_ROWS;
    if (nthreads > chunks) nthreads = chunks > 0 ? chunks : 1; /* more threads than chunks would idle */
    static FrameJob job; /* (16 KB of per-thread statistics: not on the stack) */
    pthread_mutex_lock(&serial);
    memset(&job, 0, sizeof job);
    job.c = &c; job.image = image; job.y0 = y0; job.rows = rows; job.frame = frame; job.wantStats = stats != NULL;
    job.margins = g_next_margins;
    g_next_margins = NULL;
    int helpers = nthreads - 1;
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.created < helpers) { /* grow the pool; a thread that cannot be created (EAGAIN) just means fewer helpers */
        if (pthread_create(&g_pool.th[g_pool.created], NULL, pool_worker, (void *)(intptr_t)(g_pool.created + 1)) != 0) break;
        pthread_detach(g_pool.th[g_pool.created]);
        g_pool.created++;
    }
    if (helpers > g_pool.created) helpers = g_pool.created;
    g_pool.job = &job;
    g_pool.wanted = helpers;
    g_pool.running = helpers;
    g_pool.gen++;
    if (helpers > 0) pthread_cond_broadcast(&g_pool.wake);
    pthread_mutex_unlock(&g_pool.mu);
    render_chunks(&job, 0); /* the calling thread takes chunks too (and all of them when it has no helpers) */
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.running > 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
    if (stats) {
        memset(stats, 0, 6 * sizeof(uint64_t));
        for (int t = 0; t <= helpers; t++) {
            stats[0] += job.st[t].samples; stats[1] += job.st[t].bounces; stats[2] += job.st[t].sphereTests;
            stats[3] += job.st[t].cuboidTests; stats[4] += job.st[t].envLookups;
        }
    }
    pthread_mutex_unlock(&serial);
    return 0;
}

/* Decision margins (see "decision margins" above): pto_render_frame + margins[rows * width * 2] = per pixel (the smallest relative
 * error eps of the arithmetic's primitives that flips one of the frame's data-dependent comparisons (+inf: none), the absolute colour
 * error per unit eps without a flip).  Returns -1 in builds without -DPT_ORACLE_MARGINS. */
PTO_API int pto_render_frame_margins(const PtoParams *p, const float *basic144, const float *objects26624, const void *env,
                                     float *image, int y0, int rows, int frame, int nthreads, float *margins)
{
#ifdef PT_ORACLE_MARGINS
    g_next_margins = margins;
    return pto_render_frame(p, basic144, objects26624, env, image, y0, rows, frame, nthreads, NULL);
#else
    (void)p; (void)basic144; (void)objects26624; (void)env; (void)image; (void)y0; (void)rows; (void)frame; (void)nthreads; (void)margins;
    return -1;
#endif
}

/* Evaluate `n` listed pixels of frame `frame` starting from `last` (n x 4 floats; pass zeros for frame 0).  margins (optional,
 * PT_ORACLE_MARGINS builds): n x 2 floats, each pixel's (smallest decision margin, flip-free colour error per unit eps) of this frame. */
PTO_API int pto_render_pixels_margins(const PtoParams *p, const float *basic144, const float *objects26624, const void *env,
                                      const int *xy, int n, int frame, const float *last, float *out, float *margins)
{
    Ctx c;
    make_ctx(&c, p, basic144, objects26624, env);
    for (int i = 0; i < n; i++) {
#ifdef PT_ORACLE_MARGINS
        tl_margin = INFINITY;
        tl_cont = 0.0f;
#endif
        shade_pixel(&c, xy[2 * i], xy[2 * i + 1], frame, last + 4 * i, out + 4 * i, NULL);
#ifdef PT_ORACLE_MARGINS
        if (margins) { margins[2 * i] = tl_margin; margins[2 * i + 1] = tl_cont / (float)c.spp; }
#else
        if (margins) { margins[2 * i] = INFINITY; margins[2 * i + 1] = 0.0f; }
#endif
    }
    return 0;
}
PTO_API int pto_render_pixels(const PtoParams *p, const float *basic144, const float *objects26624, const void *env,
                              const int *xy, int n, int frame, const float *last, float *out)
{
    return pto_render_pixels_margins(p, basic144, objects26624, env, xy, n, frame, last, out, NULL);
}

/* Per-pixel bounce counts of one frame (diagnostics for the divergence study in DESIGN.md): counts[y*W+x] = number
 * of RayTrace calls the pixel's samples made. */
PTO_API int pto_bounce_counts(const PtoParams *p, const float *basic144, const float *objects26624, const void *env,
                              int frame, int *counts)
{
    Ctx c;
    make_ctx(&c, p, basic144, objects26624, env);
    float last[4] = { 0, 0, 0, 0 }, out[4];
    for (int y = 0; y < c.height; y++)
        for (int x = 0; x < c.width; x++) {
            Stats st = { 0, 0, 0, 0, 0, 0 };
            shade_pixel(&c, x, y, frame, last, out, &st);
            counts[(size_t)y * c.width + x] = (int)st.bounces;
        }
    return 0;
}

/* ---- micro entry points for unit tests ---- */
/* witness build only: primitive `prim` returns results `ulps` units in the last place further from zero (negative: nearer); -1 / 0 = off.
 * Returns -1 in builds without -DPT_ORACLE_PERTURB.  Set while nothing renders. */
PTO_API int pto_set_perturbation(int prim, int ulps)
{
#ifdef PT_ORACLE_PERTURB
    g_perturb_prim = prim;
    g_perturb_ulps = ulps;
    return 0;
#else
    (void)prim; (void)ulps;
    return -1;
#endif
}

/* witness build: 1 = every multiply-add outside the primitives is evaluated with two roundings (what llvmpipe does); 0 = the contract. */
PTO_API int pto_set_unfused(int on)
{
#ifdef PT_ORACLE_PERTURB
    g_unfuse_all = on != 0;
    return 0;
#else
    (void)on;
    return -1;
#endif
}

/* witness build: the implementation the searches and replays run around.  bits: 1 = never fuse a * b + c (outside the primitives), 2 =
 * correctly rounded 1/x, 1/sqrt, sqrt, 4 = the literal a / b where the contract multiplies by a reciprocal (cuboid slabs, sphere normal,
 * throughput).  7 = all three, what llvmpipe does; 0 = the contract. */
PTO_API int pto_set_base_variant(int bits)
{
#ifdef PT_ORACLE_PERTURB
    g_unfuse_all = (bits & 1) != 0;
    g_base_exact = (bits & 2) != 0;
    g_base_truediv = (bits & 4) != 0;
    g_base_matvec = (bits >> 3) & 3; /* 0 = the contract's x, y, z, w chain; 1 = w, z, y, x; 2 = ((w + x) + z) + y, llvmpipe's; 3 = x, y, w, z */
    g_base_dot = (bits >> 5) & 3;    /* 0 = the contract's x, y, z chain; 1 = y, z, x; 2 = z, x, y */
    g_base_sampler_lerp = (bits >> 9) & 1;
    g_base_mix_lerp = (bits >> 8) & 1;
    g_base_llvm_math = (bits >> 7) & 1; /* 128 = sin, cos, exp, pow as llvmpipe's gallivm evaluates them (bit-identical on the probe) */
    return 0;
#else
    (void)bits;
    return -1;
#endif
}

/* witness build: llvmpipe's built-ins as restated above, on arrays (which: 0 sin, 1 cos, 2 exp, 3 pow(x, y), 4 exp2, 5 log2) — for the
 * probe test that compares them bit for bit with the live llvmpipe.  Returns -1 in builds without the hooks. */
PTO_API int pto_llvmpipe_like(int which, const float *x, const float *y, int n, float *out)
{
#ifdef PT_ORACLE_PERTURB
    for (int i = 0; i < n; i++)
        out[i] = which == 0 ? ll_sin_or_cos(x[i], 0) : which == 1 ? ll_sin_or_cos(x[i], 1) : which == 2 ? ll_exp(x[i])
               : which == 3 ? ll_pow(x[i], y[i]) : which == 4 ? ll_exp2(x[i]) : ll_log2(x[i]);
    return 0;
#else
    (void)which; (void)x; (void)y; (void)n; (void)out;
    return -1;
#endif
}

/* witness build: 1 = the alpha channel of every rendered pixel carries 23 bits of its PATH SIGNATURE (a hash of which object each bounce
 * hit and from which side, which lobe it took and how the path ended, over the samples of the pixel and — through the previous alpha —
 * the frames accumulated so far) instead of 1.0; 0 = the reference's alpha again.  Two implementations whose pixel has the same
 * signature followed the same path through the scene, whatever their colours are. */
PTO_API int pto_set_signature_alpha(int on)
{
#ifdef PT_ORACLE_PERTURB
    g_sig_alpha = on != 0;
    return 0;
#else
    (void)on;
    return -1;
#endif
}

/* witness build, diagnostic: the comparisons of pixel (x, y) whose operands are closer than closeGap (relative to their scale), in path
 * order: out4[4 k] = decision index, [4 k + 1] = source line of the DECIDE site in this file, [4 k + 2] = relative gap, [4 k + 3] = a - b.
 * Returns how many (at most cap), -1 in builds without the hooks. */
PTO_API int pto_list_close_decisions(const PtoParams *p, const float *basic144, const float *objects26624, const void *env,
                                     int x, int y, int frame, const float *last4, float closeGap, int cap, float *out4)
{
#ifdef PT_ORACLE_PERTURB
    Ctx c;
    float v[4];
    make_ctx(&c, p, basic144, objects26624, env);
    g_record_gap = closeGap;
    shade_pixel(&c, x, y, frame, last4, v, NULL);
    g_record_gap = 0.0f;
    const int n = tl_close_n < cap ? tl_close_n : cap;
    for (int k = 0; k < n; k++) {
        out4[4 * k] = (float)tl_close[k].idx; out4[4 * k + 1] = (float)tl_close[k].line;
        out4[4 * k + 2] = tl_close[k].gap; out4[4 * k + 3] = tl_close[k].diff;
    }
    return n;
#else
    (void)p; (void)basic144; (void)objects26624; (void)env; (void)x; (void)y; (void)frame; (void)last4; (void)closeGap; (void)cap; (void)out4;
    return -1;
#endif
}

/* witness build: the library becomes ensemble member `seed` (0: the contract / the base variant again): every primitive call up to
 * min(amplitude, its allowance) ulps off, every multiply-add fused or not, every division literal or by reciprocal — each a fixed
 * pseudo-random function of the member and the operands (see ens_hash).  Thread-safe to render with; set while nothing renders. */
PTO_API int pto_set_ensemble(unsigned seed, int amplitude)
{
#ifdef PT_ORACLE_PERTURB
    g_ens_seed = seed;
    g_ens_amp = amplitude < 0 ? 0 : amplitude;
    /* what GLSL / GL leave UNDEFINED a member also chooses for itself: pow(x, 5) of a negative base (and, every third member, of a base
       within four ulps of zero: the last bit of 1 - dot(-d, n)) is NaN or the product; texture(env, NaN direction) is some colour */
    g_pow_neg_nan = seed == 0 ? 0 : (int)(seed % 3u);
    g_nan_env_set = seed != 0;
    for (int ch = 0; ch < 3; ch++) g_nan_env[ch] = (float)(ens_hash(100u + (uint32_t)ch, 0u) >> 8) * (1.0f / 16777216.0f);
    return 0;
#else
    (void)seed; (void)amplitude;
    return -1;
#endif
}

/* witness build: texture(env, NaN direction) returns rgb3 (NULL: the contract's clamped lookup again).  The pixel is linear in this value,
 * so two replays (0 and 1) tell which value of the undefined lookup would reproduce a given pixel of the reference. */
PTO_API int pto_set_nan_env(const float *rgb3)
{
#ifdef PT_ORACLE_PERTURB
    g_nan_env_set = rgb3 != NULL;
    if (rgb3) memcpy(g_nan_env, rgb3, sizeof g_nan_env);
    return 0;
#else
    (void)rgb3;
    return -1;
#endif
}

/* witness build: evaluate ONE pixel with up to three of its comparisons inverted (flips3[k] = decision index, -1 = none) and nsites
 * primitive calls shifted (sites[3 t] = primitive, [3 t + 1] = call index within the pixel, [3 t + 2] = ulps); powNegNan 1 / 2: pow() of a
 * negative (or within four ulps of zero) base is NaN.  Returns the number of
 * DECIDE sites the evaluation passed (-1 in builds without the hooks). */
PTO_API int pto_render_pixel_variant(const PtoParams *p, const float *basic144, const float *objects26624, const void *env,
                                     int x, int y, int frame, const float *last4, const int *flips3, int nsites, const int *sites, int powNegNan,
                                     float *out4)
{
#ifdef PT_ORACLE_PERTURB
    Ctx c;
    make_ctx(&c, p, basic144, objects26624, env);
    if (nsites > WIT_MAX_SITES) return -2;
    g_pow_neg_nan = powNegNan;
    for (int k = 0; k < WIT_MAX_FLIPS; k++) g_flip_at[k] = flips3 ? flips3[k] : -1;
    for (int t = 0; t < nsites; t++) { g_tprim[t] = sites[3 * t]; g_tcall[t] = sites[3 * t + 1]; g_tulps[t] = sites[3 * t + 2]; }
    g_tn = nsites;
    shade_pixel(&c, x, y, frame, last4, out4, NULL);
    for (int k = 0; k < WIT_MAX_FLIPS; k++) g_flip_at[k] = -1;
    g_tn = 0;
    g_pow_neg_nan = 0;
    return tl_dec_n;
#else
    (void)p; (void)basic144; (void)objects26624; (void)env; (void)x; (void)y; (void)frame; (void)last4; (void)flips3; (void)nsites; (void)sites; (void)powNegNan; (void)out4;
    return -1;
#endif
}

#ifdef PT_ORACLE_PERTURB
/* distance from the reference in units of the band: <= 1 is inside (tests/tolerances.py within(); NaN == NaN agrees: the reference has
 * NaN pixels by design) */
static double wit_distance(const float *ref3, const float *got, double band)
{
    int refNan = 0, gotNan = 0;
    double worst = 0.0;
    for (int ch = 0; ch < 3; ch++) { refNan |= ref3[ch] != ref3[ch]; gotNan |= got[ch] != got[ch]; }
    if (refNan || gotNan) return refNan && gotNan ? 0.0 : INFINITY;
    for (int ch = 0; ch < 3; ch++) {
        const double r = ref3[ch], tolc = band * (fabs(r) > 1.0 ? fabs(r) : 1.0), d = fabs(r - (double)got[ch]) / tolc;
        if (!(d <= worst)) worst = d; /* (inf / NaN stay) */
    }
    return worst;
}
static int wit_cmp_gap(const void *a, const void *b)
{
    const float ga = ((const float *)a)[1], gb = ((const float *)b)[1];
    return ga < gb ? -1 : ga > gb;
}
typedef struct { int prim, call; double move; } WitSite;
static int wit_cmp_move(const void *a, const void *b)
{
    const double ma = ((const WitSite *)a)->move, mb = ((const WitSite *)b)->move;
    return ma > mb ? -1 : ma < mb;
}
/* what implementations may differ by, in ulps, per primitive (0 rcp, 1 rsqrt, 2 sqrt, 3 sin, 4 cos, 5 exp, 6 pow5): GLSL 4.60 section 4.7.1
 * allows 2.5 ulp for a / b, 2 for inversesqrt, leaves sin / cos / exp to the implementation and derives pow from exp2 / log2 (llvmpipe's
 * pow(x, 5) is ~22 ulps from the product, its exp ~16); the search stays well inside */
static const int wit_ulps[10] = { 2, 2, 2, 4, 4, 4, 16, 1, 1, 1 }; /* (7 = a multiply-add unfused, 8 = a true division, 9 = the other form of mix: on or off) */
#endif

/* witness build: search a conforming neighbour of the contract that puts pixel (x, y) of frame `frame` inside band * max(1, |ref|) of
 * the reference's value ref3.  Order: (0) pow() of a negative base returns NaN (undefined in GLSL; llvmpipe does); (1) each comparison whose operands are closer than closeGap (relative to their scale), nearest
 * first, inverted alone; (2) each call of each primitive alone, +-1 .. its allowance; (3) pairs: a close comparison inverted + one LATER
 * close comparison of the changed path inverted; (4) several calls at once: the calls that move the pixel at all, most sensitive first,
 * each set to the shift (within its allowance) that brings the pixel nearest, two sweeps (coordinate descent).
 * (5) every combination of -2 .. +2 ulps on the six most sensitive calls.
 * Returns 0 = none, 1 = single flip, 2 = single call, 3 = pair of flips, 4 / 5 = several calls, 9 = pow(x < 0, 5) = NaN, 7 / 8 = no neighbour inside but the path
 * (with one comparison inverted / as it is) ends in the environment lookup of a NaN direction, undefined in GL; flips3 / sites (capacity 3 * 32) / *nsites
 * describe the witness for pto_render_pixel_variant; stats4 = { variants evaluated, calls that move the pixel by more than the band
 * when one ulp off, the largest such move in units of the band x 1000 (saturated), the remaining distance in units of the band x 1000 }.
 * out4 = the witness's (or the nearest variant's) pixel.  Single-threaded. */
PTO_API int pto_witness_search(const PtoParams *p, const float *basic144, const float *objects26624, const void *env,
                               int x, int y, int frame, const float *last4, const float *ref3, double band,
                               float closeGap, int maxFlips, int *flips3, int *sites, int *nsites, int *stats4, float *out4)
{
#ifdef PT_ORACLE_PERTURB
    Ctx c;
    make_ctx(&c, p, basic144, objects26624, env);
    int tried = 0, found = 0;
    float base[4], v[4];
    flips3[0] = flips3[1] = flips3[2] = -1;
    *nsites = 0;
    /* dry pass: the decisions worth inverting and the primitives' call counts */
    static float close1[WIT_MAX_CLOSE][2], close2[WIT_MAX_CLOSE][2];
    static WitSite moved[65536];
    int calls[10], nmoved = 0, unstable = 0;
    double largest = 0.0;
    g_record_gap = closeGap;
    shade_pixel(&c, x, y, frame, last4, base, NULL);
    memcpy(out4, base, sizeof base);
    const int baseNanEnv = tl_nan_env;
    int flipToNanEnv = -1;
    const int n1 = tl_close_n;
    for (int k = 0; k < n1; k++) { close1[k][0] = (float)tl_close[k].idx; close1[k][1] = tl_close[k].gap; }
    memcpy(calls, tl_call_n, sizeof calls);
    qsort(close1, (size_t)n1, sizeof close1[0], wit_cmp_gap);
    const int nf = n1 < maxFlips ? n1 : maxFlips;
    g_record_gap = 0.0f;
    for (int mode = 1; mode <= 2 && !found; mode++) { /* (0) pow(x < 0, 5) = NaN: 1 - cos(theta) an ulp below zero in the Fresnel term (a camera at the centre of a glass sphere) */
        g_pow_neg_nan = mode;
        shade_pixel(&c, x, y, frame, last4, v, NULL);
        g_pow_neg_nan = 0;
        tried++;
        if (wit_distance(ref3, v, band) <= 1.0) { found = 9; flips3[2] = mode; memcpy(out4, v, sizeof v); } /* (flips3[2]: the mode, for the replay) */
        else if (tl_nan_env && !baseNanEnv && flipToNanEnv == -1) { flipToNanEnv = -2; flips3[2] = mode; }
    }
    for (int k = 0; k < nf && !found; k++) { /* (1) */
        g_flip_at[0] = (int)close1[k][0];
        shade_pixel(&c, x, y, frame, last4, v, NULL);
        tried++;
        if (wit_distance(ref3, v, band) <= 1.0) { found = 1; flips3[0] = g_flip_at[0]; memcpy(out4, v, sizeof v); }
        else if (tl_nan_env && flipToNanEnv < 0) flipToNanEnv = g_flip_at[0];
    }
    g_flip_at[0] = -1;
    g_tn = 1;
    for (int prim = 0; prim < 10 && !found; prim++) /* (2) */
        for (int n = 0; n < calls[prim] && !found; n++)
            for (int u = 1; u <= wit_ulps[prim] && !found; u++)
                for (int sgn = 1; sgn >= (prim >= 7 ? 1 : -1) && !found; sgn -= 2) {
                    g_tprim[0] = prim; g_tcall[0] = n; g_tulps[0] = sgn * u;
                    shade_pixel(&c, x, y, frame, last4, v, NULL);
                    tried++;
                    if (wit_distance(ref3, v, band) <= 1.0) {
                        found = 2; sites[0] = prim; sites[1] = n; sites[2] = sgn * u; *nsites = 1; memcpy(out4, v, sizeof v);
                    }
                    if (u == 1 && sgn == 1) { /* how far ONE ulp at this call moves the pixel, in units of the band around the contract's value */
                        const double mv = wit_distance(base, v, band);
                        if (mv > 1.0) unstable++;
                        if (mv > largest) largest = mv;
                        if (mv > 0.0 && nmoved < 65536) { moved[nmoved].prim = prim; moved[nmoved].call = n; moved[nmoved].move = mv; nmoved++; }
                    }
                }
    g_tn = 0;
    const int npair = nf < 24 ? nf : 24;
    for (int k = 0; k < npair && !found; k++) { /* (3) */
        const int first = (int)close1[k][0];
        g_flip_at[0] = first;
        g_record_gap = closeGap;
        shade_pixel(&c, x, y, frame, last4, v, NULL);
        g_record_gap = 0.0f;
        int n2 = 0;
        for (int q = 0; q < tl_close_n; q++)
            if (tl_close[q].idx > first) { close2[n2][0] = (float)tl_close[q].idx; close2[n2][1] = tl_close[q].gap; n2++; }
        qsort(close2, (size_t)n2, sizeof close2[0], wit_cmp_gap);
        if (n2 > 24) n2 = 24;
        for (int q = 0; q < n2 && !found; q++) {
            g_flip_at[1] = (int)close2[q][0];
            shade_pixel(&c, x, y, frame, last4, v, NULL);
            tried++;
            if (wit_distance(ref3, v, band) <= 1.0) { found = 3; flips3[0] = first; flips3[1] = g_flip_at[1]; memcpy(out4, v, sizeof v); }
        }
        g_flip_at[1] = -1;
    }
    g_flip_at[0] = g_flip_at[1] = -1;
    double best = wit_distance(ref3, base, band);
    if (!found && nmoved > 0 && best < INFINITY) { /* (4) */
        qsort(moved, (size_t)nmoved, sizeof moved[0], wit_cmp_move);
        const int ns = nmoved < WIT_MAX_SITES ? nmoved : WIT_MAX_SITES;
        for (int t = 0; t < ns; t++) { g_tprim[t] = moved[t].prim; g_tcall[t] = moved[t].call; g_tulps[t] = 0; }
        g_tn = ns;
        for (int sweep = 0; sweep < 2 && !found; sweep++)
            for (int t = 0; t < ns && !found; t++) {
                const int U = wit_ulps[g_tprim[t]];
                int keep = g_tulps[t];
                for (int u = (g_tprim[t] >= 7 ? 0 : -U); u <= U && !found; u++) {
                    if (u == keep) continue;
                    g_tulps[t] = u;
                    shade_pixel(&c, x, y, frame, last4, v, NULL);
                    tried++;
                    const double dist = wit_distance(ref3, v, band);
                    if (dist < best) { best = dist; keep = u; memcpy(out4, v, sizeof v); }
                    if (dist <= 1.0) found = 4;
                }
                g_tulps[t] = keep;
            }
        /* (5) the paths that amplify answer a shifted call CHAOTICALLY (the roundings downstream change too: +1 ulp at one normalisation
           moved a pixel by -0.3 bands, -1 by -1.2, +2 by +2.8), so shifts do not add up and descent is a poor guide: enumerate every
           combination of -2 .. +2 ulps on the six calls the pixel is most sensitive to (15,625 neighbours of the contract) */
        if (!found) {
            const int K = ns < 6 ? ns : 6;
            int odo[6], lo[6], hi[6];
            for (int t = 0; t < K; t++) { lo[t] = g_tprim[t] >= 7 ? 0 : -2; hi[t] = g_tprim[t] >= 7 ? 1 : 2; odo[t] = lo[t]; }
            for (int t = 0; t < ns; t++) g_tulps[t] = 0;
            g_tn = K;
            for (;;) {
                for (int t = 0; t < K; t++) g_tulps[t] = odo[t];
                shade_pixel(&c, x, y, frame, last4, v, NULL);
                tried++;
                const double dist = wit_distance(ref3, v, band);
                if (dist < best) { best = dist; memcpy(out4, v, sizeof v); }
                if (dist <= 1.0) { found = 5; break; }
                int t = 0;
                while (t < K && ++odo[t] > hi[t]) { odo[t] = lo[t]; t++; }
                if (t == K) break;
            }
            if (!found) for (int t = 0; t < K; t++) g_tulps[t] = 0;
        }
        if (found) {
            int m = 0;
            for (int t = 0; t < ns; t++)
                if (g_tulps[t] != 0) { sites[3 * m] = g_tprim[t]; sites[3 * m + 1] = g_tcall[t]; sites[3 * m + 2] = g_tulps[t]; m++; }
            *nsites = m;
        }
        g_tn = 0;
    }
    /* no neighbour lands inside, but the pixel's path — the contract's (8), or the contract's with one close comparison inverted (7, e.g.
       refract's k < 0: total internal reflection -> refract() = 0 -> normalize(0) = NaN) — ends in texture(env, NaN direction), which GL
       leaves undefined: llvmpipe returns one deterministic texel average, the contract another (docs/parity.md) */
    if (!found && baseNanEnv) found = 8;
    if (!found && flipToNanEnv != -1) { found = 7; flips3[0] = flipToNanEnv; /* (-2: through pow(x < 0) = NaN) */ }
    stats4[0] = tried;
    stats4[1] = unstable;
    stats4[2] = largest * 1000.0 < 2e9 ? (int)(largest * 1000.0) : 2000000000;
    stats4[3] = found ? 0 : (best * 1000.0 < 2e9 ? (int)(best * 1000.0) : 2000000000);
    return found;
#else
    (void)p; (void)basic144; (void)objects26624; (void)env; (void)x; (void)y; (void)frame; (void)last4; (void)ref3; (void)band;
    (void)closeGap; (void)maxFlips; (void)flips3; (void)sites; (void)nsites; (void)stats4; (void)out4;
    return -1;
#endif
}
PTO_API uint32_t pto_pcg_hash(uint32_t *seed) { return pcg_hash(seed); }
PTO_API float pto_rand01(uint32_t *seed) { return rand01(seed); }
PTO_API uint32_t pto_pixel_seed(int x, int y, int frame)
{ return ((uint32_t)x * 1973u + (uint32_t)y * 9277u + (uint32_t)frame * 2699u) | 1u; }
PTO_API void pto_sincos(float a, float *s, float *c) { f_sincos(a, s, c); }
PTO_API float pto_exp(float x) { return f_exp(x); }
PTO_API int pto_ray_sphere(const float *o, const float *d, const float *pos_r, float *t12)
{ return ray_sphere(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), V(pos_r[0], pos_r[1], pos_r[2]), pos_r[3], t12, t12 + 1); }
PTO_API int pto_ray_cuboid(const float *o, const float *d, const float *mn, const float *mx, float *t12)
{
    v3 dd = V(d[0], d[1], d[2]);
    return ray_cuboid(V(o[0], o[1], o[2]), dd, V(f_rcp(dd.x), f_rcp(dd.y), f_rcp(dd.z)), V(mn[0], mn[1], mn[2]),
                      V(mx[0], mx[1], mx[2]), t12, t12 + 1);
}
PTO_API void pto_cuboid_normal(const float *mn, const float *mx, const float *p, float *n)
{ v3 r = cuboid_normal(V(mn[0], mn[1], mn[2]), V(mx[0], mx[1], mx[2]), V(p[0], p[1], p[2])); n[0] = r.x; n[1] = r.y; n[2] = r.z; }
PTO_API void pto_sample_env(const void *env, int size, int format, const float *dir, float *rgb_out)
{
    Ctx c;
    memset(&c, 0, sizeof c);
    c.env = env; c.envSize = size; c.envFormat = format;
    fill_lut(c.srgbLut);
    rgb o = sample_env(&c, V(dir[0], dir[1], dir[2]));
    rgb_out[0] = o.r; rgb_out[1] = o.g; rgb_out[2] = o.b;
}
PTO_API float pto_srgb_to_linear(int v) { return srgb_to_linear(v); }
PTO_API float pto_pow5(float x) { return f_pow5(x); }
PTO_API float pto_fresnel_schlick(float cosTheta, float n1, float n2) { return fresnel_schlick(cosTheta, n1, n2); }
PTO_API void pto_refract(const float *i, const float *n, float eta, float *out)
{ v3 r = f_refract(V(i[0], i[1], i[2]), V(n[0], n[1], n[2]), eta); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
PTO_API void pto_reflect(const float *i, const float *n, float *out)
{ v3 r = f_reflect(V(i[0], i[1], i[2]), V(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
PTO_API void pto_cosine_sample_hemisphere(const float *n, uint32_t *seed, float *out)
{ v3 r = cosine_sample_hemisphere(V(n[0], n[1], n[2]), seed); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
PTO_API void pto_normalize(const float *v, float *out)
{ v3 r = v_normalize(V(v[0], v[1], v[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }

/* ------------------------------------------------------------------ post-process ("next" row 1)
 * res/shaders/PostProcessing/fragment.glsl:17-44, run by src/Render/ScreenEffect.cs:29-37 into an RGBA8 target:
 *   color = texture(Sampler0, uv).rgb (+ an unbound sampler = 0); ACESFilm; LinearToInverseGamma(color, 2.4); alpha 1.
 * pt-f32 contract additions: log(x) = the cephes-style degree-9 polynomial below (~1 ulp), pow(x,y) = exp(y*log(x)),
 * a/b = a * f_rcp(b); float -> unorm8 = round-half-up of clamp(x,0,1)*255 (GL 4.5 section 2.3.5.1 leaves ties open). */
static float f_log(float x)
{
    uint32_t u = f_bits(x);
    int e = (int)(u >> 23) - 127;
    float m = f_unbits((u & 0x007fffffu) | 0x3f800000u); /* [1,2) */
    if (m > 1.41421356237f) { m *= 0.5f; e += 1; }
    float t = m - 1.0f, z = t * t;
    float y = fmaf(7.0376836292e-2f, t, -1.1514610310e-1f);
    y = fmaf(y, t, 1.1676998740e-1f);
    y = fmaf(y, t, -1.2420140846e-1f);
    y = fmaf(y, t, 1.4249322787e-1f);
    y = fmaf(y, t, -1.6668057665e-1f);
    y = fmaf(y, t, 2.0000714765e-1f);
    y = fmaf(y, t, -2.4999993993e-1f);
    y = fmaf(y, t, 3.3333331174e-1f);
    y = y * t * z;
    float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    return fmaf(0.693359375f, fe, t + y);
}

static inline float f_clamp01(float x) { return f_min(f_max(x, 0.0f), 1.0f); }

static float aces_film(float x) /* fragment.glsl:35-43 */
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    float num = x * fmaf(a, x, b), den = fmaf(x, fmaf(c, x, d), e);
    return f_clamp01(num * f_rcp(den));
}

static float linear_to_inverse_gamma(float v, float gamma) /* fragment.glsl:28-32 */
{
    if (v < 0.0031308f) return v * 12.92f;
    return fmaf(f_exp(f_rcp(gamma) * f_log(v)), 1.055f, -0.055f);
}

static inline uint8_t to_unorm8(float v)
{
    float c = f_clamp01(v) * 255.0f + 0.5f;
    return (uint8_t)(int)c;
}

/* out_f (optional): the shader's float colour per pixel (n x 3); out_u8 (optional): the RGBA8 target (n x 4) */
PTO_API int pto_postprocess(const float *rgba, int n, float *out_f, uint8_t *out_u8)
{
    for (int i = 0; i < n; i++) {
        float c[3];
        for (int k = 0; k < 3; k++) c[k] = linear_to_inverse_gamma(aces_film(rgba[4 * i + k]), 2.4f);
        if (out_f) { out_f[3 * i] = c[0]; out_f[3 * i + 1] = c[1]; out_f[3 * i + 2] = c[2]; }
        if (out_u8) {
            out_u8[4 * i] = to_unorm8(c[0]); out_u8[4 * i + 1] = to_unorm8(c[1]); out_u8[4 * i + 2] = to_unorm8(c[2]);
            out_u8[4 * i + 3] = 255;
        }
    }
    return 0;
}
PTO_API float pto_log(float x) { return f_log(x); }
/* the contract's software reciprocal / inverse square root / square root, array form (accuracy property tests) */
PTO_API void pto_rcp_array(const float *x, float *y, int n) { for (int i = 0; i < n; i++) y[i] = f_rcp(x[i]); }
PTO_API void pto_rsqrt_array(const float *x, float *y, int n) { for (int i = 0; i < n; i++) y[i] = f_rsqrt(x[i]); }
PTO_API void pto_sqrt_array(const float *x, float *y, int n) { for (int i = 0; i < n; i++) y[i] = pt_sqrt(x[i]); }

/* ------------------------------------------------------------------ atmosphere precompute
 * res/shaders/AtmosphericScattering/compute.glsl:30-171 (algorithm credited there to
 * github.com/wwwtyro/glsl-atmosphere).  Same pt-f32 contract. */
static void atmo_rsi(v3 r0, v3 rd, float sr, float *x, float *y) /* :58-71 */
{
    float a = v_dot(rd, rd);
    float b = 2.0f * v_dot(rd, r0);
    float c = fmaf(-sr, sr, v_dot(r0, r0));
    float d = fmaf(b, b, -(4.0f * a * c));
    if (d < 0.0f) { *x = 1e5f; *y = -1e5f; return; }
    /* (round 5: the per-step roots use the contract's pt_sqrt and ONE pt-f32 reciprocal instead of sqrtf and two IEEE divisions —
       the correctly rounded forms are 52 / 43 issue cycles each on gfx950, and this function runs 53 times per texel) */
    float sq = pt_sqrt(d), rden = f_rcp(2.0f * a);
    *x = (-b - sq) * rden;
    *y = (-b + sq) * rden;
}

static v3 atmosphere(v3 r, v3 r0, v3 pSun, float iSun, float rPlanet, float rAtmos, v3 kRlh, float kMie,
                     float shRlh, float shMie, float g, int iSteps, int jSteps) /* :73-159 */
{
    pSun = v_normalize(pSun);
    r = v_normalize(r);
    float px, py, qx, qy;
    atmo_rsi(r0, r, rAtmos, &px, &py);
    if (px > py) return V(0.0f, 0.0f, 0.0f);
    atmo_rsi(r0, r, rPlanet, &qx, &qy);
    py = f_min(py, qx);
    float iStepSize = (py - px) / (float)iSteps;
    float iTime = 0.0f;
    v3 totalRlh = V(0, 0, 0), totalMie = V(0, 0, 0);
    float iOdRlh = 0.0f, iOdMie = 0.0f;
    float mu = v_dot(r, pSun), mumu = mu * mu, gg = g * g;
    float pRlh = 3.0f / (16.0f * PI) * (1.0f + mumu);
    float base = 1.0f + gg - 2.0f * mu * g;
    float pMie = 3.0f / (8.0f * PI) * ((1.0f - gg) * (mumu + 1.0f)) / ((base * sqrtf(base)) * (2.0f + gg));
    float invShRlh = -1.0f / shRlh, invShMie = -1.0f / shMie; /* exp(-h/sh) evaluated as exp(h * (-1/sh)) */
    const float invJSteps = 1.0f / (float)jSteps;               /* uniform IEEE reciprocal: sy / jSteps is evaluated as sy * (1 / jSteps) */
    for (int i = 0; i < iSteps; i++) {
        v3 iPos = v_fma(r, fmaf(iStepSize, 0.5f, iTime), r0);
        float iHeight = pt_sqrt(v_dot(iPos, iPos)) - rPlanet;
        float odStepRlh = f_exp(iHeight * invShRlh) * iStepSize;
        float odStepMie = f_exp(iHeight * invShMie) * iStepSize;
        iOdRlh += odStepRlh;
        iOdMie += odStepMie;
        float sx, sy;
        atmo_rsi(iPos, pSun, rAtmos, &sx, &sy);
        float jStepSize = sy * invJSteps;
        float jTime = 0.0f, jOdRlh = 0.0f, jOdMie = 0.0f;
        for (int j = 0; j < jSteps; j++) {
            v3 jPos = v_fma(pSun, fmaf(jStepSize, 0.5f, jTime), iPos);
            float jHeight = pt_sqrt(v_dot(jPos, jPos)) - rPlanet;
            jOdRlh = fmaf(f_exp(jHeight * invShRlh), jStepSize, jOdRlh);
            jOdMie = fmaf(f_exp(jHeight * invShMie), jStepSize, jOdMie);
            jTime += jStepSize;
        }
        float mieTerm = kMie * (iOdMie + jOdMie), rl = iOdRlh + jOdRlh;
        v3 attn = V(f_exp(-fmaf(kRlh.x, rl, mieTerm)), f_exp(-fmaf(kRlh.y, rl, mieTerm)), f_exp(-fmaf(kRlh.z, rl, mieTerm)));
        totalRlh = v_fma(attn, odStepRlh, totalRlh);
        totalMie = v_fma(attn, odStepMie, totalMie);
        iTime += iStepSize;
    }
    float pm = pMie * kMie;
    return V(iSun * fmaf(pRlh * kRlh.x, totalRlh.x, pm * totalMie.x),
             iSun * fmaf(pRlh * kRlh.y, totalRlh.y, pm * totalMie.y),
             iSun * fmaf(pRlh * kRlh.z, totalRlh.z, pm * totalMie.z));
}

typedef struct { const float *ubo; const float *lightPos; float intensity; int size, iSteps, jSteps; float *out; int tid, nthreads; } AtmoJob;

static void *atmo_worker(void *arg)
{
    AtmoJob *j = (AtmoJob *)arg;
    int S = j->size;
    for (int idx = j->tid; idx < 6 * S; idx += j->nthreads) {
        int face = idx / S, y = idx % S;
        const float *invView = j->ubo + 16 + 16 * face;
        for (int x = 0; x < S; x++) {
            /* main :30-56 : ndc = vec2(imgCoord.xy) / size * 2 - 1 (texel corner, no +0.5) */
            float ndcx = fmaf((float)x / (float)S, 2.0f, -1.0f), ndcy = fmaf((float)y / (float)S, 2.0f, -1.0f);
            float eye[4], wd[4];
            mat_vec(j->ubo, ndcx, ndcy, -1.0f, 0.0f, eye);
            mat_vec(invView, eye[0], eye[1], -1.0f, 0.0f, wd);
            v3 dir = v_normalize(V(wd[0], wd[1], wd[2]));
            v3 col = atmosphere(dir, V(0.0f, 6376e3f, 0.0f), V(j->lightPos[0], j->lightPos[1], j->lightPos[2]), j->intensity,
                                6371e3f, 6471e3f, V(5.5e-6f, 13.0e-6f, 22.4e-6f), 21e-6f, 8e3f, 1.2e3f, 0.758f,
                                j->iSteps, j->jSteps);
            float *o = j->out + (((size_t)face * S + y) * S + x) * 4;
            o[0] = col.x; o[1] = col.y; o[2] = col.z; o[3] = 1.0f;
        }
    }
    return NULL;
}

/* out: float[6][size][size][4]; ubo464: InvProjection + 6 InvView (AtmosphericScatterer.cs:72-89) */
PTO_API int pto_atmosphere(const float *ubo464, const float *lightPos, float lightIntensity, int size, int iSteps,
                           int jSteps, float *out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    AtmoJob jobs[256];
    for (int t = 0; t < nthreads; t++) {
        AtmoJob j = { ubo464, lightPos, lightIntensity, size, iSteps, jSteps, out, t, nthreads };
        jobs[t] = j;
        if (nthreads > 1) pthread_create(&th[t], NULL, atmo_worker, &jobs[t]);
    }
    if (nthreads == 1) atmo_worker(&jobs[0]);
    else for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    return 0;
}
