// pt_math.hpp — the "pt-f32" arithmetic contract on the device side (gfx950).
//
// Every routine here is a fixed sequence of IEEE-754 binary32 operations (correctly rounded +,-,*,sqrt and
// explicitly written fma), so the HIP integrator reproduces the CPU oracle bit for bit.  Reciprocals and inverse
// square roots on the per-bounce path are the contract's Newton sequences f_rcp / f_rsqrt (the correctly rounded
// IEEE divide / sqrt cost 43 / 52 issue cycles on gfx950, tools/ubench.hip; the hardware approximations cannot be
// reproduced on a CPU); per-frame uniform quotients use the IEEE operator `/` via f_div_ieee.  The translation unit
// that includes this header must be compiled with  -ffp-contract=off -fno-fast-math  and WITHOUT
// -fgpu-flush-denormals-to-zero (hipcc's default keeps denormals and uses correctly rounded fp32 divide/sqrt).
// No hardware approximations (v_rcp/v_rsq/v_sin/v_exp) are used on the parity path.
//
// GLSL built-ins of the reference shader (res/shaders/PathTracing/compute.glsl) map as follows:
//   dot -> v_dot (fma chain), normalize -> v * f_rsqrt(dot), a/b -> a * f_rcp(b), mix -> fma(y,a,x*(1-a)),
//   min/max -> minNum/maxNum,
//   sin/cos -> pt_sincos, exp -> pt_exp, pow(x,5.0) -> pt_pow5, reflect/refract -> GLSL 4.50 section 8.5 formulas.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pt {

#define PT_DEV __device__ __forceinline__

constexpr float FLOAT_MAX = 3.4028235e+38f;  // compute.glsl:2
constexpr float FLOAT_MIN = -3.4028235e+38f; // compute.glsl:3
constexpr float EPSILON = 0.001f;            // compute.glsl:4
constexpr float PI = 3.14159265f;            // compute.glsl:5

struct v3 {
    float x, y, z;
};

PT_DEV float f_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_DEV float f_min(float a, float b) { return __builtin_fminf(a, b); }
PT_DEV float f_max(float a, float b) { return __builtin_fmaxf(a, b); }
PT_DEV float f_div_ieee(float a, float b) { return a / b; } // correctly rounded; uniform / once-per-object uses only
// pt-f32 reciprocal: seed by exponent negation, three Newton steps (<= 0.51 ulp); zero and denormals give +-inf
PT_DEV float f_rcp(float x)
{
    float y = __uint_as_float(0x7EF311C7u - __float_as_uint(x));
    float e = f_fma(-x, y, 1.0f); y = f_fma(y, e, y);
    e = f_fma(-x, y, 1.0f); y = f_fma(y, e, y);
    e = f_fma(-x, y, 1.0f); y = f_fma(y, e, y);
    if (__builtin_fabsf(x) < 1.17549435e-38f) y = __builtin_copysignf(__builtin_inff(), x);
    return y;
}
// pt-f32 inverse square root: classic seed, three Newton steps (<= 1.7 ulp); zero/denormal -> +inf, negative -> NaN
PT_DEV float f_rsqrt(float x)
{
    float y = __uint_as_float(0x5F3759DFu - (__float_as_uint(x) >> 1));
    float h = 0.5f * x, t;
    t = y * y; t = f_fma(-h, t, 1.5f); y = y * t;
    t = y * y; t = f_fma(-h, t, 1.5f); y = y * t;
    t = y * y; t = f_fma(-h, t, 1.5f); y = y * t;
    if (x < 1.17549435e-38f) y = x < 0.0f ? __builtin_nanf("") : __builtin_inff();
    return y;
}
PT_DEV float f_sqrt(float a) { return __builtin_sqrtf(a); } // correctly rounded (atmosphere precompute, uniform uses)
// pt-f32 square root: two Newton steps y *= 1.5 - (x/2*y)*y on the classic inverse-square-root seed (4.7e-6), then one residual
// correction s += (x - s*s) * y/2 (<= 0.501 ulp).  sqrt(0) = 0 exactly; negative, infinite and NaN inputs give a non-finite
// value (every call site guards its argument)..
PT_DEV float pt_sqrt(float x)
{
    float y = __uint_as_float(0x5F3759DFu - (__float_as_uint(x) >> 1));
    float h = 0.5f * x, t;
    t = h * y; t = f_fma(-t, y, 1.5f); y = y * t;
    t = h * y; t = f_fma(-t, y, 1.5f); y = y * t;
    float s = x * y;
    float r = f_fma(-s, s, x);
    return f_fma(r, 0.5f * y, s);
}
PT_DEV float f_abs(float a) { return __builtin_fabsf(a); }
PT_DEV float f_mix(float x, float y, float a) { return f_fma(y, a, x * (1.0f - a)); }

PT_DEV v3 V(float x, float y, float z) { return v3{x, y, z}; }
PT_DEV v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV v3 v_mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV v3 v_scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
PT_DEV v3 v_neg(v3 a) { return V(-a.x, -a.y, -a.z); }
PT_DEV v3 v_fma(v3 b, float s, v3 a) { return V(f_fma(b.x, s, a.x), f_fma(b.y, s, a.y), f_fma(b.z, s, a.z)); }
PT_DEV float v_dot(v3 a, v3 b) { return f_fma(a.z, b.z, f_fma(a.y, b.y, a.x * b.x)); }
PT_DEV v3 v_normalize(v3 a) { return v_scale(a, f_rsqrt(v_dot(a, a))); }
PT_DEV v3 v_mix(v3 x, v3 y, float a)
{
    float ia = 1.0f - a;
    return V(f_fma(y.x, a, x.x * ia), f_fma(y.y, a, x.y * ia), f_fma(y.z, a, x.z * ia));
}

// sin/cos on the small arguments the integrator produces ([0, 2*pi]): Cody-Waite by pi/2 (two fused steps),
// single-precision minimax polynomials on [-pi/4, pi/4], quadrant fix-up.
PT_DEV void pt_sincos(float a, float &sn, float &cs)
{
    float k = __builtin_rintf(a * 0.636619772f);
    float r = f_fma(k, -1.57079637050628662109375f, a);
    r = f_fma(k, 4.37113900018624283e-8f, r);
    float z = r * r;
    float ps = f_fma(f_fma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = f_fma(ps * z, r, r);
    float pc = f_fma(f_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float c = f_fma(pc * z, z, f_fma(-0.5f, z, 1.0f));
    int q = (int)k & 3;
    float s_out = (q & 1) ? c : s;
    float c_out = (q & 1) ? s : c;
    if (q == 1 || q == 2) c_out = -c_out;
    if (q >= 2) s_out = -s_out;
    sn = s_out;
    cs = c_out;
}

// e^x: n = rint(x*log2 e), two-step fused reduction, degree-6 polynomial, 2^n as two exact power-of-two factors.
template <bool SELECTED = false>
PT_DEV float pt_exp(float x)
{
    // SELECTED: the three special cases are selected at the end instead of branched around (same value for every input).  A divergent
    // early return costs four scalar instructions and a branch (tools/ubench3.hip: a scalar instruction takes 0.8 of a vector
    // instruction's issue time), and the polynomial runs for some lane of the wavefront anyway; but the selected form keeps three
    // polynomial chains in flight at once and costs registers, so only the kernel that has them to spare asks for it.
    if (!SELECTED) {
        if (x != x) return x;
        if (x > 88.72283935546875f) return __builtin_inff();
        if (x < -104.0f) return 0.0f;
    }
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = f_fma(n, -0.693145751953125f, x);
    r = f_fma(n, -1.428606765330187045e-06f, r);
    float p = f_fma(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = f_fma(p, r, 8.3334519073e-3f);
    p = f_fma(p, r, 4.1665795894e-2f);
    p = f_fma(p, r, 1.6666665459e-1f);
    p = f_fma(p, r, 5.0000001201e-1f);
    float y = f_fma(p, r * r, r) + 1.0f;
    int ni = SELECTED ? (int)__builtin_fminf(__builtin_fmaxf(n, -200.0f), 200.0f) : (int)n; // (in range wherever the special cases do not take over)
    int n1 = ni >> 1, n2 = ni - n1;
    y = y * __uint_as_float((uint32_t)(n1 + 127) << 23);
    y = y * __uint_as_float((uint32_t)(n2 + 127) << 23);
    if (SELECTED) {
        y = x < -104.0f ? 0.0f : y;
        y = x > 88.72283935546875f ? __builtin_inff() : y;
        y = x != x ? x : y;
    }
    return y;
}

PT_DEV float pt_pow5(float x)
{
    float x2 = x * x;
    return x * (x2 * x2);
}

// log(x), x > 0 normal: exponent split + degree-9 polynomial in (m - 1), m in [sqrt(1/2), sqrt(2))  (~1 ulp)
PT_DEV float pt_log(float x)
{
    uint32_t u = __float_as_uint(x);
    int e = (int)(u >> 23) - 127;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237f) { m *= 0.5f; e += 1; }
    float t = m - 1.0f, z = t * t;
    float y = f_fma(7.0376836292e-2f, t, -1.1514610310e-1f);
    y = f_fma(y, t, 1.1676998740e-1f);
    y = f_fma(y, t, -1.2420140846e-1f);
    y = f_fma(y, t, 1.4249322787e-1f);
    y = f_fma(y, t, -1.6668057665e-1f);
    y = f_fma(y, t, 2.0000714765e-1f);
    y = f_fma(y, t, -2.4999993993e-1f);
    y = f_fma(y, t, 3.3333331174e-1f);
    y = y * t * z;
    float fe = (float)e;
    y = f_fma(-2.12194440e-4f, fe, y);
    y = f_fma(-0.5f, z, y);
    return f_fma(0.693359375f, fe, t + y);
}

PT_DEV float f_clamp01(float x) { return f_min(f_max(x, 0.0f), 1.0f); }

// PostProcessing/fragment.glsl:35-43 ACESFilm, per channel
PT_DEV float aces_film(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    float num = x * f_fma(a, x, b), den = f_fma(x, f_fma(c, x, d), e);
    return f_clamp01(num * f_rcp(den));
}

// PostProcessing/fragment.glsl:28-32 LinearToInverseGamma, per channel; pow(x,y) = exp(y * log(x))
PT_DEV float linear_to_inverse_gamma(float v, float gamma)
{
    if (v < 0.0031308f) return v * 12.92f;
    return f_fma(pt_exp(f_rcp(gamma) * pt_log(v)), 1.055f, -0.055f);
}

PT_DEV unsigned char to_unorm8(float v) { return (unsigned char)(int)(f_clamp01(v) * 255.0f + 0.5f); }

// compute.glsl:334-344 — PCG hash RNG; uint -> float conversion is round-to-nearest-even, /2^32 is exact.
PT_DEV uint32_t pcg_hash(uint32_t &seed)
{
    seed = seed * 747796405u + 2891336453u;
    uint32_t word = ((seed >> ((seed >> 28u) + 4u)) ^ seed) * 277803737u;
    return (word >> 22u) ^ word;
}
PT_DEV float rand01(uint32_t &seed) { return (float)pcg_hash(seed) * 2.3283064365386962890625e-10f; }

} // namespace pt
