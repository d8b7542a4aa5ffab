// pt_atmosphere.hpp — device functions of the atmosphere environment precompute (included by pt_helper_kernels.hip only).
#pragma once
#include "pt_device.hpp"

namespace pt {

// ---------------------------------------------------------------------------------------------- atmosphere
// /root/reference/OpenTK-PathTracer/res/shaders/AtmosphericScattering/compute.glsl:30-171
// (algorithm credited there to github.com/wwwtyro/glsl-atmosphere); one thread per cube texel.
PT_DEV void atmo_rsi(v3 r0, v3 rd, float sr, float &x, float &y) // :58-71
{
    float a = v_dot(rd, rd);
    float b = 2.0f * v_dot(rd, r0);
    float c = f_fma(-sr, sr, v_dot(r0, r0));
    float d = f_fma(b, b, -(4.0f * a * c));
    if (d < 0.0f) { x = 1e5f; y = -1e5f; return; }
    // (the contract's pt_sqrt and ONE pt-f32 reciprocal: the correctly rounded sqrt and two divisions were 138 issue cycles of this
    // function, which runs 53 times per texel)
    float sq = pt_sqrt(d), rden = f_rcp(2.0f * a);
    x = (-b - sq) * rden;
    y = (-b + sq) * rden;
}

PT_DEV v3 atmosphere(v3 r, v3 r0, v3 pSun, float iSun, float rPlanet, float rAtmos, v3 kRlh, float kMie, float shRlh,
                     float shMie, float g, int iSteps, int jSteps) // :73-159
{
    pSun = v_normalize(pSun);
    r = v_normalize(r);
    float px, py, qx, qy;
    atmo_rsi(r0, r, rAtmos, px, py);
    if (px > py) return V(0.0f, 0.0f, 0.0f);
    atmo_rsi(r0, r, rPlanet, qx, qy);
    py = f_min(py, qx);
    float iStepSize = (py - px) / (float)iSteps;
    float iTime = 0.0f;
    v3 totalRlh = V(0.0f, 0.0f, 0.0f), totalMie = V(0.0f, 0.0f, 0.0f);
    float iOdRlh = 0.0f, iOdMie = 0.0f;
    float mu = v_dot(r, pSun), mumu = mu * mu, gg = g * g;
    float pRlh = 3.0f / (16.0f * PI) * (1.0f + mumu);
    float base = 1.0f + gg - 2.0f * mu * g;
    float pMie = 3.0f / (8.0f * PI) * ((1.0f - gg) * (mumu + 1.0f)) / ((base * f_sqrt(base)) * (2.0f + gg));
    float invShRlh = -1.0f / shRlh, invShMie = -1.0f / shMie;
    const float invJSteps = f_div_ieee(1.0f, (float)jSteps); // uniform
    for (int i = 0; i < iSteps; i++) {
        v3 iPos = v_fma(r, f_fma(iStepSize, 0.5f, iTime), r0);
        float iHeight = pt_sqrt(v_dot(iPos, iPos)) - rPlanet;
        float odStepRlh = pt_exp(iHeight * invShRlh) * iStepSize;
        float odStepMie = pt_exp(iHeight * invShMie) * iStepSize;
        iOdRlh += odStepRlh;
        iOdMie += odStepMie;
        float sx, sy;
        atmo_rsi(iPos, pSun, rAtmos, sx, sy);
        float jStepSize = sy * invJSteps;
        float jTime = 0.0f, jOdRlh = 0.0f, jOdMie = 0.0f;
        for (int j = 0; j < jSteps; j++) {
            v3 jPos = v_fma(pSun, f_fma(jStepSize, 0.5f, jTime), iPos);
            float jHeight = pt_sqrt(v_dot(jPos, jPos)) - rPlanet;
            jOdRlh = f_fma(pt_exp(jHeight * invShRlh), jStepSize, jOdRlh);
            jOdMie = f_fma(pt_exp(jHeight * invShMie), jStepSize, jOdMie);
            jTime += jStepSize;
        }
        float mieTerm = kMie * (iOdMie + jOdMie), rl = iOdRlh + jOdRlh;
        v3 attn = V(pt_exp(-f_fma(kRlh.x, rl, mieTerm)), pt_exp(-f_fma(kRlh.y, rl, mieTerm)),
                    pt_exp(-f_fma(kRlh.z, rl, mieTerm)));
        totalRlh = v_fma(attn, odStepRlh, totalRlh);
        totalMie = v_fma(attn, odStepMie, totalMie);
        iTime += iStepSize;
    }
    float pm = pMie * kMie;
    return V(iSun * f_fma(pRlh * kRlh.x, totalRlh.x, pm * totalMie.x),
             iSun * f_fma(pRlh * kRlh.y, totalRlh.y, pm * totalMie.y),
             iSun * f_fma(pRlh * kRlh.z, totalRlh.z, pm * totalMie.z));
}

} // namespace pt
