// pt_device.hpp — device functions of the path-tracing integrator (included by the integrator kernels through pt_kernel_common.hpp, and by pt_helper_kernels.hip): environment
// sampling, scene traversal (+ the per-tile sphere culling of the tile pass), sampling / BSDF, one bounce, camera
// rays, pixel resolve.  Each function cites the lines of
//   /root/reference/OpenTK-PathTracer/res/shaders/PathTracing/compute.glsl
// it implements; the arithmetic is the "pt-f32" contract of pt_math.hpp (bit-identical to oracle/pt_oracle.c).
#pragma once
#include "pt_debug_hooks.hpp"
#include "pt_kernels.hpp"
#include "pt_math.hpp"

namespace pt {

struct Material { // std140 Material, compute.glsl:13-26
    v3 albedo;
    float specularChance;
    v3 emissiv;
    float specularRoughness;
    v3 absorbance;
    float refractionChance;
    float refractionRoughness, ior;
};

struct Hit { // compute.glsl:44-51 HitInfo
    float T;
    bool fromInside;
    v3 nearHitPos, normal;
    Material m;
};

struct SceneLds {
    const float4 *sph;  // [numSpheres]  (centre.xyz, radius)
    const float4 *cmin; // [numCuboids]
    const float4 *cmax; // [numCuboids]
    const float4 *mat;  // [(numSpheres + numCuboids) * 4]
    const float *invr;  // [numSpheres rounded up to 4] 1 / radius (IEEE quotient, computed once per workgroup)
    const float *lut;   // [256] sRGB8 -> linear (only staged for SRGB8_A8 environments)
    const float4 *objects; // the std140 GameObjectsUBO in device memory (materials of large scenes are read from here)
    const unsigned short *gridStarts; // sphere grid (large scenes, FrameArgs::grid) staged in LDS, or nullptr
    const unsigned char *gridRefs;
};

// Geometry (16 B per sphere + 4 B 1/radius, 32 B per cuboid) is read by every ray and always lives in LDS.  The
// 64-byte materials are read once per hit, by the winner only: they are staged too while the workgroup still fits
// 5-per-CU, and stay in device memory (L2-resident, 26 KB) for large scenes, where they would cost a resident workgroup.
__host__ __device__ inline size_t scene_lds_bytes(int ns, int nc, int envFormat, bool matInLds, int gridLdsBytes = 0)
{
    return (size_t)(ns + 2 * nc + (matInLds ? 4 * (ns + nc) : 0)) * 16 + (size_t)((ns + 3) & ~3) * 4 + (envFormat == 1 ? 1024 : 0) +
           (size_t)gridLdsBytes;
}

// ---------------------------------------------------------------------------------------------- environment
// texture(SamplerEnvironment, dir) (compute.glsl:177): LOD 0, LINEAR magnification, seamless cube edges
// (src/MainWindow.cs:168,178).  Face selection per OpenGL 4.5 table 8.19 (ties: Z, then X, then Y).
typedef const __attribute__((address_space(3))) float *LdsFloats; // keeps LUT reads as ds_read (a generic pointer would make them flat loads)
struct EnvRef {
    const void *data;
    LdsFloats lut;
    int size, format;
};

// Rarely needed launch parameters are re-read from the kernarg segment where they are used (scalar loads, always
// cached) instead of being kept live in SGPRs for the whole kernel: the bounce loop needs its SGPRs for exec-mask
// nesting, and every spilled SGPR costs v_writelane / v_readlane VALU slots.  FrameArgs is the kernel's first argument.
// The pointer stays in the CONSTANT address space (4), so every access is a scalar s_load (uniform, cached), not a
// per-lane flat load.
typedef const __attribute__((address_space(4))) FrameArgs *ColdArgs;
typedef const __attribute__((address_space(4))) float *ColdFloats;
PT_DEV ColdArgs cold_args()
{
    ColdArgs p = (ColdArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

PT_DEV v3 env_texel(const EnvRef &e, int face, int x, int y)
{
    size_t idx = ((size_t)face * e.size + (size_t)y) * e.size + (size_t)x;
    if (e.format == 0) {
        float4 t = ((const float4 *)e.data)[idx];
        return V(t.x, t.y, t.z);
    }
    uchar4 t = ((const uchar4 *)e.data)[idx];
    return V(e.lut[t.x], e.lut[t.y], e.lut[t.z]);
}

PT_DEV void face_to_dir(int face, float sc, float tc, float &x, float &y, float &z)
{
    switch (face) {
    case 0: x = 1.0f; y = -tc; z = -sc; break;
    case 1: x = -1.0f; y = -tc; z = sc; break;
    case 2: x = sc; y = 1.0f; z = tc; break;
    case 3: x = sc; y = -1.0f; z = -tc; break;
    case 4: x = sc; y = -tc; z = 1.0f; break;
    default: x = -sc; y = -tc; z = -1.0f; break;
    }
}

PT_DEV void dir_to_face(float x, float y, float z, int &face, float &sc, float &tc, float &ma)
{
    float ax = f_abs(x), ay = f_abs(y), az = f_abs(z);
    if (az >= f_max(ax, ay)) {
        face = z < 0.0f ? 5 : 4; ma = az; sc = z < 0.0f ? -x : x; tc = -y;
    } else if (ax >= ay) {
        face = x < 0.0f ? 1 : 0; ma = ax; sc = x < 0.0f ? z : -z; tc = -y;
    } else {
        face = y < 0.0f ? 3 : 2; ma = ay; sc = x; tc = y < 0.0f ? -z : z;
    }
}

// texel (ix,iy), possibly one step outside `face` in one direction -> the texel across the seam
PT_DEV v3 env_texel_wrapped(const EnvRef &e, int face, int ix, int iy)
{
    int S = e.size;
    if (ix >= 0 && ix < S && iy >= 0 && iy < S) return env_texel(e, face, ix, iy);
    float fs = (float)S;
    float rfs = f_div_ieee(1.0f, fs); // uniform
    float sc = ((float)ix + 0.5f) * rfs * 2.0f - 1.0f;
    float tc = ((float)iy + 0.5f) * rfs * 2.0f - 1.0f;
    float x, y, z, ma, nsc, ntc;
    int nface;
    face_to_dir(face, sc, tc, x, y, z);
    dir_to_face(x, y, z, nface, nsc, ntc, ma);
    float rma = f_rcp(ma);
    float u = (nsc * rma * 0.5f + 0.5f) * fs;
    float v = (ntc * rma * 0.5f + 0.5f) * fs;
    int nx = (int)__builtin_floorf(u), ny = (int)__builtin_floorf(v);
    nx = nx < 0 ? 0 : (nx > S - 1 ? S - 1 : nx);
    ny = ny < 0 ? 0 : (ny > S - 1 ? S - 1 : ny);
    return env_texel(e, nface, nx, ny);
}

PT_DEV v3 sample_env(const EnvRef &e, v3 d)
{
    int S = e.size, face;
    float sc, tc, ma;
    dir_to_face(d.x, d.y, d.z, face, sc, tc, ma);
    float ima = 0.5f * f_rcp(ma);
    float fs = (float)S;
    float u = f_fma(sc, ima, 0.5f) * fs - 0.5f;
    float v = f_fma(tc, ima, 0.5f) * fs - 0.5f;
    u = f_min(f_max(u, -1.0f), fs); // NaN / inf directions: defined, identical clamp on CPU and GPU
    v = f_min(f_max(v, -1.0f), fs);
    float fu = __builtin_floorf(u), fv = __builtin_floorf(v);
    float wu = u - fu, wv = v - fv;
    int x0 = (int)fu, y0 = (int)fv, x1 = x0 + 1, y1 = y0 + 1;
    bool offx0 = x0 < 0, offx1 = x1 >= S, offy0 = y0 < 0, offy1 = y1 >= S;
    float w00 = (1.0f - wu) * (1.0f - wv), w10 = wu * (1.0f - wv), w01 = (1.0f - wu) * wv, w11 = wu * wv;
    v3 t00, t10, t01, t11;
    if (!(offx0 || offx1 || offy0 || offy1)) { // interior: the overwhelmingly common case
        // one base index, the other three taps at +1, +S, +S+1 (32-bit index math, a single 64-bit address)
        const unsigned base = ((unsigned)face * (unsigned)S + (unsigned)y0) * (unsigned)S + (unsigned)x0;
        if (e.format == 0) {
            const float4 *p = (const float4 *)e.data + base;
            float4 a = p[0], b = p[1], c = p[S], d = p[S + 1];
            t00 = V(a.x, a.y, a.z); t10 = V(b.x, b.y, b.z); t01 = V(c.x, c.y, c.z); t11 = V(d.x, d.y, d.z);
        } else {
            const uchar4 *p = (const uchar4 *)e.data + base;
            uchar4 a = p[0], b = p[1], c = p[S], d = p[S + 1];
            t00 = V(e.lut[a.x], e.lut[a.y], e.lut[a.z]); t10 = V(e.lut[b.x], e.lut[b.y], e.lut[b.z]);
            t01 = V(e.lut[c.x], e.lut[c.y], e.lut[c.z]); t11 = V(e.lut[d.x], e.lut[d.y], e.lut[d.z]);
        }
    } else {
        bool miss00 = offx0 && offy0, miss10 = offx1 && offy0, miss01 = offx0 && offy1, miss11 = offx1 && offy1;
        v3 zero = V(0.0f, 0.0f, 0.0f);
        t00 = miss00 ? zero : env_texel_wrapped(e, face, x0, y0);
        t10 = miss10 ? zero : env_texel_wrapped(e, face, x1, y0);
        t01 = miss01 ? zero : env_texel_wrapped(e, face, x0, y1);
        t11 = miss11 ? zero : env_texel_wrapped(e, face, x1, y1);
        if (miss00 || miss10 || miss01 || miss11) {
            // cube corner: the missing tap's weight is shared equally by the three existing taps
            float a = (miss00 ? w00 : miss10 ? w10 : miss01 ? w01 : w11) * 0.333333343f;
            w00 = miss00 ? 0.0f : w00 + a;
            w10 = miss10 ? 0.0f : w10 + a;
            w01 = miss01 ? 0.0f : w01 + a;
            w11 = miss11 ? 0.0f : w11 + a;
        }
    }
    v3 o;
    o.x = f_fma(t11.x, w11, f_fma(t01.x, w01, f_fma(t10.x, w10, t00.x * w00)));
    o.y = f_fma(t11.y, w11, f_fma(t01.y, w01, f_fma(t10.y, w10, t00.y * w00)));
    o.z = f_fma(t11.z, w11, f_fma(t01.z, w01, f_fma(t10.z, w10, t00.z * w00)));
    return o;
}

// ---------------------------------------------------------------------------------------------- traversal
PT_DEV float f_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
PT_DEV float f_step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }

// compute.glsl:322-332 GetNormal(Cuboid)
PT_DEV v3 cuboid_normal(v3 mn, v3 mx, v3 p)
{
    v3 half = v_scale(v_sub(mx, mn), 0.5f);
    v3 cs = v_sub(p, v_scale(v_add(mx, mn), 0.5f));
    v3 n;
    n.x = f_sign(cs.x) * f_step(f_abs(f_abs(cs.x) - half.x), EPSILON);
    n.y = f_sign(cs.y) * f_step(f_abs(f_abs(cs.y) - half.y), EPSILON);
    n.z = f_sign(cs.z) * f_step(f_abs(f_abs(cs.z) - half.z), EPSILON);
    return v_normalize(n);
}

PT_DEV Material load_material(const float4 *m)
{
    float4 a = m[0], b = m[1], c = m[2], d = m[3];
    Material r;
    r.albedo = V(a.x, a.y, a.z);     r.specularChance = a.w;
    r.emissiv = V(b.x, b.y, b.z);    r.specularRoughness = b.w;
    r.absorbance = V(c.x, c.y, c.z); r.refractionChance = c.w;
    r.refractionRoughness = d.x;     r.ior = d.y;
    return r;
}

// compute.glsl:226-258 RayTrace (+ :261-294 intersections, :316-332 normals).
// Acceptance uses the ENTRY distance t1 against the stored GetSmallestPositive (compute.glsl:234,247,347-350);
// objects are visited in reference order; material + normal are evaluated once for the surviving candidate.
// MASKED: only the spheres whose bit is set in the wave-uniform masks[0..3] are visited (still in index order).  The
// masks come from cull_spheres(): a sphere outside them fails its own `t2 > 0` / discriminant test for EVERY ray of
// the wavefront, and a sphere that fails its own test never changes T — skipping it cannot change the result.
//
// GRID (large scenes, generic bounce): instead of all ns spheres a ray visits the spheres listed in the cells of a uniform
// grid along its way (3D-DDA, every lane on its own), and stops once the accepted distance lies before the exit of the
// current cell.  The result is the one the in-order loop produces, by this argument.  Call a sphere VALID for the ray when
// it passes `disc >= 0, t1 <= t2, t2 > 0`, INSIDE when in addition t1 < 0 (the origin is in it), OUTSIDE otherwise.  The
// in-order loop accepts every inside sphere (t1 < 0 < T always) and an outside sphere only if t1 < T, so with L = the
// highest-index inside sphere (none: L = -1, T = FLT_MAX) its result is: the outside sphere of index > L with the smallest
// t1 below t2(L), the lowest index among equal t1 — or L itself.  That is a minimum over a set and may be taken in any
// order given the two tie rules.  Inside spheres contain the origin, so all of them are listed in the FIRST cell, whose
// list is ascending: processed with the reference's own rule it yields L; later cells apply the set rule (index > L;
// smaller t1, or equal t1 and lower index than an outside winner).  A sphere's hit point lies in its bounding box, which is
// inflated at build time by more than the rounding error of t1 for origins within `gridReach` of the grid (pt_sphere_grid.hpp),
// so it is listed in the cell that contains the hit point and has been tested when the walk reaches that cell.  Lanes the
// argument does not cover — origin out of reach, non-finite or non-unit direction, an inside sphere met after the first
// cell — take the in-order loop afterwards (needBrute).
// SHARE (the default kernel's generic bounce): the in-order sphere loop shares what consecutive spheres with equal centre x and z
// have in common (sphere runs, below).
// WALK SLICES (GRID): the lanes of a wavefront walk in lock step, one cell per round, so a wavefront pays for its longest walk
// (10.9 cells for the slowest of 64 rays of the 256-sphere scene, 3.3 on average: tools/grid_walk_model.py).  A call therefore walks at
// most PT_WALK_ROUNDS cells; a lane that is not done by then returns with walkFrom = the parameter at which its ray leaves the last
// cell it processed, takes no part in the rest of the bounce, and the caller traces the SAME ray again in its next iteration, entering
// the grid at walkFrom — beside lanes that meanwhile took fresh paths, so every round runs with (nearly) all lanes.  Nothing found
// before walkFrom needs to be remembered: an unfinished walk means T > the exit of every processed cell, so the accepted sphere's hit
// point, and every better one, lies in a cell from walkFrom on and is listed there (the set rule above does not depend on the order);
// a sphere that contains the origin and reaches past walkFrom is met again with first == false and sends the lane through the
// in-order loop, as an inside sphere after the first cell always did.
#ifndef PT_WALK_ROUNDS
#define PT_WALK_ROUNDS 6
#endif

template <bool MASKED, bool MATLDS, bool GRID = false, bool SHARE = false>
PT_DEV bool ray_trace_t(const SceneLds &sc, int ns, int nc, v3 o, v3 d, Hit &h, const unsigned long long *masks, float &walkFrom PROF_PARAM)
{
    PROF_BEGIN
    float T = FLOAT_MAX, wt2 = 0.0f;
    int winner = -1;
    bool needBrute = true;
    float wf = -1.0f; // GRID: >= 0 when this call leaves the walk unfinished (see WALK SLICES below)
    if (GRID && !MASKED && sc.gridStarts != nullptr) {
        ColdArgs ca = cold_args();
        const v3 rel = V(o.x - ca->gridCenter[0], o.y - ca->gridCenter[1], o.z - ca->gridCenter[2]);
        const float dd = v_dot(d, d);
        // (the intersection formulas assume a unit direction, compute.glsl:261-277; every direction the integrator produces is
        // normalised to ~1e-7, and the margins of the build cover 1e-5: anything else — and NaN anywhere — takes the in-order loop)
        needBrute = !(v_dot(rel, rel) <= ca->gridReach2 && __builtin_fabsf(dd - 1.0f) < 1e-5f);
        if (!needBrute) {
            // reciprocal direction, clamped: no infinities / NaNs in the walk (an axis the ray is parallel to is never chosen)
            const float ix_ = __builtin_fabsf(d.x) > 1e-18f ? __builtin_amdgcn_rcpf(d.x) : (d.x < 0.0f ? -1e18f : 1e18f);
            const float iy_ = __builtin_fabsf(d.y) > 1e-18f ? __builtin_amdgcn_rcpf(d.y) : (d.y < 0.0f ? -1e18f : 1e18f);
            const float iz_ = __builtin_fabsf(d.z) > 1e-18f ? __builtin_amdgcn_rcpf(d.z) : (d.z < 0.0f ? -1e18f : 1e18f);
            const float ax0 = (ca->gridLo[0] - o.x) * ix_, ax1 = (ca->gridHi[0] - o.x) * ix_;
            const float ay0 = (ca->gridLo[1] - o.y) * iy_, ay1 = (ca->gridHi[1] - o.y) * iy_;
            const float az0 = (ca->gridLo[2] - o.z) * iz_, az1 = (ca->gridHi[2] - o.z) * iz_;
            float tn = f_max(0.0f, f_max(f_min(ax0, ax1), f_max(f_min(ay0, ay1), f_min(az0, az1))));
            const bool resumed = walkFrom >= 0.0f;
            tn = resumed ? f_max(tn, walkFrom) : tn;
            const float tf = f_min(f_max(ax0, ax1), f_min(f_max(ay0, ay1), f_max(az0, az1)));
            if (tn <= tf) { // (else: the ray misses the box that holds every sphere)
                const int nx = ca->gridDims[0], ny = ca->gridDims[1], nz = ca->gridDims[2];
                const v3 p = v_fma(d, tn, o);
                int cx = (int)((p.x - ca->gridLo[0]) * ca->gridInvCell[0]), cy = (int)((p.y - ca->gridLo[1]) * ca->gridInvCell[1]),
                    cz = (int)((p.z - ca->gridLo[2]) * ca->gridInvCell[2]);
                cx = cx < 0 ? 0 : (cx >= nx ? nx - 1 : cx);
                cy = cy < 0 ? 0 : (cy >= ny ? ny - 1 : cy);
                cz = cz < 0 ? 0 : (cz >= nz ? nz - 1 : cz);
                const bool px = !(d.x < 0.0f), py = !(d.y < 0.0f), pz = !(d.z < 0.0f);
                // parameter at which the ray leaves the current cell along each axis, its increment per cell, cells left to the box's face
                float mx = (ca->gridLo[0] + (float)(cx + (px ? 1 : 0)) * ca->gridCell[0] - o.x) * ix_;
                float my = (ca->gridLo[1] + (float)(cy + (py ? 1 : 0)) * ca->gridCell[1] - o.y) * iy_;
                float mz = (ca->gridLo[2] + (float)(cz + (pz ? 1 : 0)) * ca->gridCell[2] - o.z) * iz_;
                const float dx = __builtin_fabsf(ca->gridCell[0] * ix_), dy = __builtin_fabsf(ca->gridCell[1] * iy_),
                            dz = __builtin_fabsf(ca->gridCell[2] * iz_);
                int rx = px ? nx - 1 - cx : cx, ry = py ? ny - 1 - cy : cy, rz = pz ? nz - 1 - cz : cz;
                int cell = (cz * ny + cy) * nx + cx;
                const int sx = px ? 1 : -1, sy = py ? nx : -nx, sz = pz ? nx * ny : -(nx * ny);
                int L = -1, viol = 0; // (flags carried through divergent loops live in VGPRs: a bool would cost mask bookkeeping per round)
                bool first = !resumed; // (a resumed walk is past its first cell)
                int k = sc.gridStarts[cell], kEnd = sc.gridStarts[cell + 1];
                for (int round = 0;; round++) {
                    // Where the ray goes next does not depend on this cell's tests: decide it first and fetch the next cell's list
                    // bounds now, so that their LDS latency is covered by the tests (only the stop criterion needs T).
                    const bool xm = mx <= my && mx <= mz, ym = !xm && my <= mz;
                    const float texit = xm ? mx : (ym ? my : mz);
                    const bool more = (xm ? rx : (ym ? ry : rz)) > 0; // cells left along the axis of the step
                    const int ncell = more ? cell + (xm ? sx : (ym ? sy : sz)) : cell;
                    const int nk = sc.gridStarts[ncell], nkEnd = sc.gridStarts[ncell + 1];
                    // two-deep software pipeline over the list: the next sphere's geometry and the index after it are in flight during
                    // a test (reads past the end of a list fetch some other byte / sphere of the staged scene: unused)
                    int j = sc.gridRefs[k], jn = sc.gridRefs[k + 1];
                    float4 s = sc.sph[j];
                    while (k < kEnd) {
                        const float4 sn = sc.sph[jn];
                        const int jnn = sc.gridRefs[k + 2];
                        k++;
                        const v3 oc = V(o.x - s.x, o.y - s.y, o.z - s.z);
                        const float b = v_dot(d, oc);
                        const float c = f_fma(-s.w, s.w, v_dot(oc, oc));
                        const float disc = f_fma(b, b, -c);
                        if (!(disc < 0.0f) && !(c > 0.0f && b > 1e-10f)) {
                            const float sq = pt_sqrt(disc);
                            const float t1 = -b - sq, t2 = -b + sq;
                            // branch-free, bit-wise on purpose: `&&` / `||` would come back as nested exec-mask regions
                            const bool valid = (t1 <= t2) & (t2 > 0.0f);
                            const bool inside = valid & (t1 < 0.0f); // the reference accepts it whatever T is (t1 < 0 < T)
                            const bool nearer = (t1 < T) | ((t1 == T) & (winner != L) & (j < winner));
                            const bool accOut = valid & !inside & (j > L) & nearer;
                            const bool accIn = inside & first;
                            viol |= (inside & !first) ? 1 : 0; // (cannot happen within the build's margins; never trust it silently)
                            const bool acc = accIn | accOut;
                            T = acc ? (inside ? t2 : t1) : T;
                            wt2 = acc ? t2 : wt2;
                            winner = acc ? j : winner;
                            L = accIn ? j : L;
                        }
                        j = jn;
                        jn = jnn;
                        s = sn;
                    }
                    first = false;
                    if (T <= texit || !more) break; // nothing listed only in later cells can be nearer / left the box
                    if (round == PT_WALK_ROUNDS - 1) { wf = texit; break; } // WALK SLICES: the rest of this ray's walk runs in the next bounce iteration
                    mx = xm ? mx + dx : mx;
                    my = ym ? my + dy : my;
                    mz = (xm || ym) ? mz : mz + dz;
                    rx = xm ? rx - 1 : rx;
                    ry = ym ? ry - 1 : ry;
                    rz = (xm || ym) ? rz : rz - 1;
                    cell = ncell;
                    k = nk;
                    kEnd = nkEnd;
                }
                needBrute = viol != 0;
            }
        }
        if (needBrute) { T = FLOAT_MAX; wt2 = 0.0f; winner = -1; wf = -1.0f; }
    }
    // Sphere pass, 4 spheres per step: the four discriminants are computed branch-free from four broadcast LDS
    // reads issued together (ILP instead of one exposed LDS latency per sphere); only lanes with a real
    // forward candidate enter the exact sqrt path, and candidates are accepted strictly in index order.
    // A sphere entirely behind the origin (c > 0: origin outside, b > 0: pointing away) can never pass `t2 > 0`
    // (sqrt(b*b - c) <= b when c > 0), so it is rejected before the square root — an exact shortcut.
    auto candidate = [&](int i, float b, float c, float disc) {
        if (!(disc < 0.0f) && !(c > 0.0f && b > 1e-10f)) {
            float sq = pt_sqrt(disc);
            float t1 = -b - sq, t2 = -b + sq;
            if (t1 <= t2 && t2 > 0.0f && t1 < T) {
                T = t1 < 0.0f ? t2 : t1;
                wt2 = t2;
                winner = i;
            }
        }
    };
    int i = 0;
    if (MASKED) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w * 64 >= ns) break;
            unsigned long long mk = masks[w];
            while (mk != 0ull) {
                int k = w * 64 + (int)__builtin_ctzll(mk);
                mk &= mk - 1ull;
                float4 s = sc.sph[k];
                v3 oc = V(o.x - s.x, o.y - s.y, o.z - s.z);
                float b = v_dot(d, oc);
                float c = f_fma(-s.w, s.w, v_dot(oc, oc));
                candidate(k, b, c, f_fma(b, b, -c));
            }
        }
        i = ns;
    }
    if (!GRID || needBrute) { // (GRID: only the lanes the grid could not serve)
    // Runs (FrameArgs::sphereRunStart): a sphere whose centre has the same x and z BITS as its predecessor's reuses o.x - c.x,
    // o.z - c.z and the products d.x * oc.x, oc.x * oc.x that start the two dot-product chains of v_dot — identical binary32 values,
    // 7 instead of 11 operations for that sphere.  The run bits are wave-uniform (scalar load, scalar branches).  The two fused
    // multiply-adds that CONSUME the shared products are written as instructions: the compiler's two-address v_fmac would overwrite
    // the product (and then copy it for the next sphere: three moves per sphere); `volatile` keeps them in program order between the
    // run headers.  Four spheres per step as before: four broadcast LDS reads issued together, one exposed LDS latency per step.
    if constexpr (!SHARE) {
    // the plain loop, four spheres per step: four broadcast LDS reads issued together, one wave-level branch per sphere
        for (; i + 4 <= ns; i += 4) {
            const float4 sv[4] = {sc.sph[i], sc.sph[i + 1], sc.sph[i + 2], sc.sph[i + 3]};
            float b[4], c[4], disc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v3 oc = V(o.x - sv[k].x, o.y - sv[k].y, o.z - sv[k].z);
                b[k] = v_dot(d, oc);
                c[k] = f_fma(-sv[k].w, sv[k].w, v_dot(oc, oc));
                disc[k] = f_fma(b[k], b[k], -c[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) // (`any`: see the run loop below)
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(disc[k] < 0.0f)) != 0ull, 0)) candidate(i + k, b[k], c[k], disc[k]);
        }
        for (; i < ns; i++) {
            float4 s = sc.sph[i];
            v3 oc = V(o.x - s.x, o.y - s.y, o.z - s.z);
            float b = v_dot(d, oc);
            float c = f_fma(-s.w, s.w, v_dot(oc, oc));
            candidate(i, b, c, f_fma(b, b, -c));
        }
    } else if (i < ns) {
        // Issue slots, not arithmetic, are what this loop costs (tools/ubench3.hip: a scalar instruction takes an issue slot like a
        // vector one): per sphere ONE asm block (the compiler pads every asm statement with an s_nop), the run bit tested straight
        // from the 32-bit word, and a sphere that no lane can hit costs one compare and one scalar branch (`any`, below).
        ColdArgs rca = cold_args();
        float ocz = 0.0f, bx = 0.0f, cx = 0.0f;
#define PT_RUN_SPHERE(S, K)                                                                                              \
        if (runs & (1u << (K)))                                                                                          \
            asm volatile("v_sub_f32 %0, %4, %6\n v_sub_f32 %1, %5, %7\n v_mul_f32 %2, %8, %0\n v_mul_f32 %3, %0, %0"     \
                         : "=&v"(ocx), "=&v"(ocz), "=&v"(bx), "=&v"(cx)                                                  \
                         : "v"(o.x), "v"(o.z), "v"(S.x), "v"(S.z), "v"(d.x)); /* oc.x, oc.z, d.x * oc.x, oc.x * oc.x */  \
        asm volatile("v_sub_f32 %2, %3, %4\n"       /* oc.y */                                                           \
                     "v_fma_f32 %0, %5, %2, %6\n"   /* fma(d.y, oc.y, d.x * oc.x) */                                      \
                     "v_fma_f32 %1, %2, %2, %7\n"   /* fma(oc.y, oc.y, oc.x * oc.x) */                                    \
                     "v_fma_f32 %0, %8, %9, %0\n"   /* == v_dot(d, oc) */                                                 \
                     "v_fma_f32 %1, %9, %9, %1"     /* == v_dot(oc, oc) */                                                \
                     : "=&v"(b[K]), "=&v"(c[K]), "=&v"(ocy)                                                              \
                     : "v"(o.y), "v"(S.y), "v"(d.y), "v"(bx), "v"(cx), "v"(d.z), "v"(ocz));                              \
        c[K] = f_fma(-S.w, S.w, c[K]);              /* ... - r * r */                                                    \
        disc[K] = f_fma(b[K], b[K], -c[K]);
        const int ns4 = ns & ~3;
        while (i < ns4) { // (i is a multiple of 32 here)
            unsigned int runs = ((const __attribute__((address_space(4))) unsigned int *)rca->sphereRunStart)[i >> 5];
            const int end = ns4 < i + 32 ? ns4 : i + 32;
            for (; i < end; i += 4) {
                const float4 s0 = sc.sph[i], s1 = sc.sph[i + 1], s2 = sc.sph[i + 2], s3 = sc.sph[i + 3];
                float b[4], c[4], disc[4], ocx, ocy;
                asm volatile("" ::"v"(s0.x), "v"(s0.z), "v"(s1.x), "v"(s1.z), "v"(s2.x), "v"(s2.z), "v"(s3.x), "v"(s3.z)); // (whole float4 reads, all four at once)
                PT_RUN_SPHERE(s0, 0)
                PT_RUN_SPHERE(s1, 1)
                PT_RUN_SPHERE(s2, 2)
                PT_RUN_SPHERE(s3, 3)
                runs >>= 4;
                asm volatile("" : "+v"(disc[0]), "+v"(disc[1]), "+v"(disc[2]), "+v"(disc[3])); // (the four tests below stay below the four blocks above)
#pragma unroll
                for (int k = 0; k < 4; k++) // `any`: a sphere that no lane's line meets costs one compare and one scalar branch (no exec save / restore)
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(disc[k] < 0.0f)) != 0ull, 0)) candidate(i + k, b[k], c[k], disc[k]);
            }
        }
#undef PT_RUN_SPHERE
        for (; i < ns; i++) { // (ns % 4 spheres: evaluated on their own)
            float4 s = sc.sph[i];
            v3 oc = V(o.x - s.x, o.y - s.y, o.z - s.z);
            float b = v_dot(d, oc);
            float c = f_fma(-s.w, s.w, v_dot(oc, oc));
            candidate(i, b, c, f_fma(b, b, -c));
        }
    }
    }
    PROF_MARK(1) // sphere pass
    if constexpr (GRID) {
        walkFrom = wf;
        if (wf >= 0.0f) return false; // (unfinished walk: the caller re-traces this bounce from walkFrom; nothing else of the bounce has happened)
    }
    v3 invd = V(f_rcp(d.x), f_rcp(d.y), f_rcp(d.z)); // slab test by reciprocal (pt-f32 contract)
    // MASKED: masks[4] = the cuboids some ray of the wavefront's bundle can reach (cone_cuboid_mask; all ones = not culled: the plain
    // loop); a cuboid outside it fails `t1 <= t2 && t2 > 0` for every lane and never changes T
    auto cuboid = [&](int i) {
        float4 mn = sc.cmin[i], mx = sc.cmax[i];
        v3 t0s = V((mn.x - o.x) * invd.x, (mn.y - o.y) * invd.y, (mn.z - o.z) * invd.z);
        v3 t1s = V((mx.x - o.x) * invd.x, (mx.y - o.y) * invd.y, (mx.z - o.z) * invd.z);
        v3 sm = V(f_min(t0s.x, t1s.x), f_min(t0s.y, t1s.y), f_min(t0s.z, t1s.z));
        v3 bg = V(f_max(t0s.x, t1s.x), f_max(t0s.y, t1s.y), f_max(t0s.z, t1s.z));
        float t1 = f_max(FLOAT_MIN, f_max(sm.x, f_max(sm.y, sm.z)));
        float t2 = f_min(FLOAT_MAX, f_min(bg.x, f_min(bg.y, bg.z)));
        if (t1 <= t2 && t2 > 0.0f && t1 < T) {
            T = t1 < 0.0f ? t2 : t1;
            wt2 = t2;
            winner = 256 + i;
        }
    };
    if (MASKED && masks[4] != ~0ull) {
        for (unsigned long long cm = masks[4]; cm != 0ull; cm &= cm - 1ull) {
            const int i = (int)__builtin_ctzll(cm);
            if (i >= nc) break;
            cuboid(i);
        }
    } else {
        for (int i = 0; i < nc; i++) cuboid(i);
    }
    PROF_MARK(2) // cuboid pass
    if (winner < 0 || !(T != FLOAT_MAX)) return false; // compute.glsl:257
    h.T = T;
    h.fromInside = (T == wt2);
    h.nearHitPos = v_fma(d, T, o);
    if (winner < 256) {
        float4 s = sc.sph[winner];
        if (MATLDS) h.m = load_material(sc.mat + 4 * winner);
        else h.m = load_material(sc.objects + 5 * winner + 1); // std140 Sphere = geometry + 4 x float4 material
        v3 pc = V(h.nearHitPos.x - s.x, h.nearHitPos.y - s.y, h.nearHitPos.z - s.z);
        h.normal = v_scale(pc, sc.invr[winner]); // compute.glsl:316-319; 1/radius = IEEE quotient staged in LDS
    } else {
        int ci = winner - 256;
        float4 mn = sc.cmin[ci], mx = sc.cmax[ci];
        if (MATLDS) h.m = load_material(sc.mat + 4 * (ns + ci));
        else h.m = load_material(sc.objects + 1280 + 6 * ci + 2); // Cuboids[] start at float4 index 1280: min, max, material
        h.normal = cuboid_normal(V(mn.x, mn.y, mn.z), V(mx.x, mx.y, mx.z), h.nearHitPos);
    }
    PROF_MARK(3) // winner: material + normal
    return true;
}
PT_DEV bool ray_trace(const SceneLds &sc, int ns, int nc, v3 o, v3 d, Hit &h PROF_PARAM)
{
    float fresh = -1.0f;
    return ray_trace_t<false, true>(sc, ns, nc, o, d, h, nullptr, fresh PROF_PASS);
}

// ---- per-tile sphere culling for a wavefront of (nearly) coherent rays
// Wavefront-wide maximum of a non-negative value.  row_shr:1/2/4/8 inside each row of 16 lanes (a lane shifted in
// from outside the row contributes 0, the identity here), then row_bcast:15 / row_bcast:31 carry the row results
// across the rows; lane 63 ends up with the maximum, which is broadcast.  Must be called with all 64 lanes active.
PT_DEV float wave_max_nonneg(float x)
{
#define PT_DPP_MAX(ctrl, rowmask) \
    x = __builtin_fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rowmask, 0xf, false)))
    PT_DPP_MAX(0x111, 0xf);
    PT_DPP_MAX(0x112, 0xf);
    PT_DPP_MAX(0x114, 0xf);
    PT_DPP_MAX(0x118, 0xf);
    PT_DPP_MAX(0x142, 0xa);
    PT_DPP_MAX(0x143, 0xc);
#undef PT_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// The 64 primary rays of one 8x8 tile leave (almost) one point in (almost) one direction.  They are bounded by a cone:
// apex region = ball of radius rho around one reference ray's origin O, axis A = that ray's direction, half-angle theta =
// the largest angle any ray of the wavefront makes with A (both measured from the actual rays, so any camera matrix,
// aperture or jitter is covered).  A ray (o, d) of the bundle can only reach sphere (c, r) with t > 0 if the ray (O, d)
// reaches the sphere (c, R = r + rho), i.e. if |c - O| <= R or angle(A, c - O) <= theta + asin(R / |c - O|).  Lane j
// tests sphere j (+64, +128, +192): masks[] = spheres that pass.  Every quantity is padded (0.1 % and absolute slack far
// above float rounding), and the test only ever REMOVES spheres that no ray of the wavefront can hit — those would fail
// `discriminant >= 0 && t2 > 0` (compute.glsl:261-277) for every lane and leave T untouched — so the traced result is
// bit-identical to visiting all spheres.  Hardware sqrt/rcp approximations are fine here: they only move the padding.
PT_DEV void cone_sphere_masks(const SceneLds &sc, int ns, v3 O, v3 A, float rho, float ct, unsigned long long masks[4]);
PT_DEV void cull_spheres(const SceneLds &sc, int ns, bool valid, v3 o, v3 d, unsigned long long masks[5])
{
    const int lane = threadIdx.x & 63;
    masks[0] = masks[1] = masks[2] = masks[3] = 0ull;
    masks[4] = ~0ull;
    const unsigned long long vm = __ballot(valid);
    if (vm == 0ull) return;
    const int ref = ((vm >> 36) & 1ull) ? 36 : (int)__builtin_ctzll(vm); // pixel (4,4) of the tile, else the first valid lane
    const v3 O = V(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.x), ref)),
                   __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.y), ref)),
                   __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.z), ref)));
    const v3 A = V(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(d.x), ref)),
                   __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d.y), ref)),
                   __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d.z), ref)));
    const v3 dv = v_sub(o, O);
    // a NaN ray hits nothing (every comparison of its intersection tests is false): fmax drops it from the bounds
    float dev2 = wave_max_nonneg(__builtin_fmaxf(valid ? v_dot(dv, dv) : 0.0f, 0.0f));
    float spread = wave_max_nonneg(__builtin_fmaxf(valid ? 1.0f - v_dot(A, d) : 0.0f, 0.0f)); // 1 - cos(angle to A)
    const float rho = __builtin_amdgcn_sqrtf(dev2) * 1.001f;
    const float ct = 1.0f - spread * 1.01f - 1e-5f;          // padded cos(theta)
    cone_sphere_masks(sc, ns, O, A, rho, ct, masks);
    masks[4] = ~0ull; // (cuboids are only culled by the cached tile masks)
}

// Lane j tests sphere j (+64, +128, +192) against the cone (apex ball of radius rho around O, axis A, cos of the half-angle ct).
PT_DEV void cone_sphere_masks(const SceneLds &sc, int ns, v3 O, v3 A, float rho, float ct, unsigned long long masks[4])
{
    const int lane = threadIdx.x & 63;
    masks[0] = masks[1] = masks[2] = masks[3] = 0ull;
    const bool coneUsable = ct > 0.05f;                       // a bundle wider than ~87 degrees is not culled at all (NaN: neither)
    const float st = __builtin_amdgcn_sqrtf(__builtin_fmaxf(f_fma(-ct, ct, 1.0f), 0.0f));
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (w * 64 >= ns) break;
        const int i = w * 64 + lane;
        const bool in = i < ns;
        const float4 s = sc.sph[in ? i : 0];
        const v3 v = V(s.x - O.x, s.y - O.y, s.z - O.z);
        const float L = __builtin_amdgcn_sqrtf(v_dot(v, v));
        const float R = f_fma(f_abs(s.w) + rho, 1.001f, f_fma(L, 1e-4f, 1e-3f));
        bool outside = false;
        if (coneUsable && L > R) { // NaN / inf anywhere -> comparisons false -> the sphere is kept
            const float sb = R * __builtin_amdgcn_rcpf(L);
            const float cb = __builtin_amdgcn_sqrtf(__builtin_fmaxf(f_fma(-sb, sb, 1.0f), 0.0f));
            const float cosSum = f_fma(ct, cb, -(st * sb)) - 1e-4f; // padded cos(theta + beta)
            outside = v_dot(v, A) < L * cosSum;
        }
        masks[w] = __ballot(in && !outside);
    }
}

// The same cone for EVERY primary ray an 8x8 tile can ever cast (any sub-pixel jitter, any lens sample; compute.glsl:113-121), so that
// the masks can be computed once per camera / scene and reused by every frame (pt_tile_masks_kernel).  The un-normalised pinhole
// direction wd = InvView * (InvProj * (ndc, -1, 0)).xy(-1)(0) is an AFFINE function of ndc, so over the tile's rectangle of pixel
// coordinates [x0, x0 + 8] x [gy0, gy0 + 8] it stays inside the convex hull of its four corner values: every pinhole direction lies
// within the largest corner angle of the axis (a circular cone of less than 90 degrees is convex).  The thin lens (origin = InvView *
// (ox, oy, 0, 1) with |(ox, oy)| <= aperture / 2, direction towards ViewPos + dir * focalLength) moves the origin by at most
// rho = aperture / 2 * ||InvView[:, 0:2]||_F from InvView's translation O, and turns the direction by at most asin(e / (f - e)),
// e = rho + |ViewPos - O|, f = focalLength.  Anything degenerate (NaN, f <= 2 e, a corner behind the axis) makes the cone unusable:
// ct = 0 keeps every sphere.
template <typename FP>
PT_DEV void mat_vec(FP m, float x, float y, float z, float w, float *out);
template <typename FP>
PT_DEV void tile_cone(FP cam, float invW, float invH, int x0, int gy0, v3 &O, v3 &A, float &rho, float &ct, v3 dirs[4], float &lensSin)
{
    FP invProj = cam, invView = cam + 16;
    v3 sum = V(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float ndcx = f_fma((float)(x0 + ((c & 1) ? 8 : 0)) * invW, 2.0f, -1.0f);
        const float ndcy = f_fma((float)(gy0 + ((c & 2) ? 8 : 0)) * invH, 2.0f, -1.0f);
        float eye[4], wd[4];
        mat_vec(invProj, ndcx, ndcy, -1.0f, 0.0f, eye);
        mat_vec(invView, eye[0], eye[1], -1.0f, 0.0f, wd);
        const float l2 = f_fma(wd[2], wd[2], f_fma(wd[1], wd[1], wd[0] * wd[0]));
        const float il = __builtin_amdgcn_rsqf(l2);
        dirs[c] = V(wd[0] * il, wd[1] * il, wd[2] * il);
        sum = V(sum.x + dirs[c].x, sum.y + dirs[c].y, sum.z + dirs[c].z);
    }
    const float sl = __builtin_amdgcn_rsqf(f_fma(sum.z, sum.z, f_fma(sum.y, sum.y, sum.x * sum.x)));
    A = V(sum.x * sl, sum.y * sl, sum.z * sl);
    float cp = 1.0f; // cos of the largest corner angle
#pragma unroll
    for (int c = 0; c < 4; c++) cp = __builtin_fminf(cp, f_fma(A.z, dirs[c].z, f_fma(A.y, dirs[c].y, A.x * dirs[c].x)));
    O = V(invView[12], invView[13], invView[14]);
    const float half_ap = __builtin_fabsf(cam[36]) * 0.5f;
    const float frob = __builtin_amdgcn_sqrtf(invView[0] * invView[0] + invView[1] * invView[1] + invView[2] * invView[2] +
                                              invView[4] * invView[4] + invView[5] * invView[5] + invView[6] * invView[6]);
    rho = f_fma(half_ap * frob, 1.002f, 1e-6f);
    const v3 c0 = V(cam[32] - O.x, cam[33] - O.y, cam[34] - O.z);
    const float e = rho + __builtin_amdgcn_sqrtf(f_fma(c0.z, c0.z, f_fma(c0.y, c0.y, c0.x * c0.x))) * 1.002f;
    const float f = cam[35];
    const bool ok = f > 2.0f * e && cp > 0.0f; // (NaN anywhere: false)
    const float q = e * __builtin_amdgcn_rcpf(f - e) * 1.002f;                          // sin of the lens' turn, padded
    const float cq = __builtin_amdgcn_sqrtf(__builtin_fmaxf(f_fma(-q, q, 1.0f), 0.0f));
    const float sp = __builtin_amdgcn_sqrtf(__builtin_fmaxf(f_fma(-cp, cp, 1.0f), 0.0f)) * 1.002f + 1e-4f; // sin of the corner angle, padded
    const float ctot = f_fma(cp, cq, -(sp * q));                                         // cos(corner angle + lens turn)
    const float spread = 1.0f - ctot;
    ct = ok ? 1.0f - spread * 1.02f - 2e-4f : 0.0f;
    rho = ok ? rho : 0.0f;
    lensSin = q;
}

// The cuboids a tile's rays can reach (bit i = cuboid i; all of them when the cone is unusable).  The rays leave the ball (O, rho) in
// directions within asin(lensSin) of the pinhole directions, which lie in the pyramid spanned by the four corner directions.  For a
// side plane of that pyramid with inward unit normal n, every point p of such a ray satisfies
//     n . (p - O) >= -rho - (|p - O| + rho) * lensSin,
// and g(p) = n . (p - O) + (|p - O| + rho) * lensSin + rho is convex in p, so a box whose eight corners all have g < 0 for ONE plane is
// reached by no ray of the tile: its slab test fails `t1 <= t2 && t2 > 0` (compute.glsl:279-294) for every lane.  Padded by 0.2 % of
// the coordinates' magnitude, far above the rounding of the slab test.  Lane j tests cuboid j.
PT_DEV unsigned long long cone_cuboid_mask(const SceneLds &sc, int nc, v3 O, float rho, float ct, const v3 dirs[4], float lensSin)
{
    const int lane = threadIdx.x & 63;
    const bool usable = ct > 0.05f;
    const bool in = lane < nc;
    const float4 mn = sc.cmin[in ? lane : 0], mx = sc.cmax[in ? lane : 0];
    const float scale = 1.0f + f_max(f_max(f_abs(O.x), f_max(f_abs(O.y), f_abs(O.z))),
                                     f_max(f_max(f_abs(mn.x), f_max(f_abs(mn.y), f_abs(mn.z))), f_max(f_abs(mx.x), f_max(f_abs(mx.y), f_abs(mx.z)))));
    const float pad = f_fma(scale, 2e-3f, rho * 1.01f);
    const float q = f_fma(lensSin, 1.01f, 1e-3f);
    bool outside = false;
    const int order[5] = {0, 1, 3, 2, 0}; // the tile's corners in cyclic order
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const v3 a = dirs[order[e]], b = dirs[order[e + 1]];
        v3 n = V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
        const v3 mid = V(dirs[0].x + dirs[1].x + dirs[2].x + dirs[3].x, dirs[0].y + dirs[1].y + dirs[2].y + dirs[3].y, dirs[0].z + dirs[1].z + dirs[2].z + dirs[3].z);
        const float il = __builtin_amdgcn_rsqf(f_fma(n.z, n.z, f_fma(n.y, n.y, n.x * n.x)));
        const float sgn = (n.x * mid.x + n.y * mid.y + n.z * mid.z) < 0.0f ? -il : il; // inward: the pyramid's axis is on the positive side
        n = V(n.x * sgn, n.y * sgn, n.z * sgn);
        bool allOut = true;
#pragma unroll
        for (int v = 0; v < 8; v++) {
            const v3 p = V(((v & 1) ? mx.x : mn.x) - O.x, ((v & 2) ? mx.y : mn.y) - O.y, ((v & 4) ? mx.z : mn.z) - O.z);
            const float len = __builtin_amdgcn_sqrtf(f_fma(p.z, p.z, f_fma(p.y, p.y, p.x * p.x)));
            const float g = f_fma(n.x, p.x, f_fma(n.y, p.y, n.z * p.z)) + f_fma(len + rho, q, pad);
            allOut = allOut && (g < 0.0f); // (NaN: false -> the cuboid is kept)
        }
        outside = outside || allOut;
    }
    const unsigned long long keep = __ballot(in && !(usable && outside));
    return keep;
}

// ---------------------------------------------------------------------------------------------- sampling / BSDF
// compute.glsl:297-307
PT_DEV v3 cosine_sample_hemisphere(v3 n, uint32_t &seed)
{
    float z = f_fma(rand01(seed), 2.0f, -1.0f);
    float a = rand01(seed) * 2.0f * PI;
    float r = pt_sqrt(f_fma(-z, z, 1.0f));
    float sn, cs;
    pt_sincos(a, sn, cs);
    return v_normalize(v_add(n, V(r * cs, r * sn, z)));
}

// compute.glsl:359-364
PT_DEV float fresnel_schlick(float cosTheta, float n1, float n2)
{
    float r0 = (n1 - n2) * f_rcp(n1 + n2);
    r0 *= r0;
    return f_fma(1.0f - r0, pt_pow5(1.0f - cosTheta), r0);
}

PT_DEV v3 f_reflect(v3 i, v3 n) { return v_fma(n, -(2.0f * v_dot(n, i)), i); }

PT_DEV v3 f_refract(v3 i, v3 n, float eta)
{
    float ni = v_dot(n, i);
    float k = f_fma(-(eta * eta), f_fma(-ni, ni, 1.0f), 1.0f);
    if (k < 0.0f) return V(0.0f, 0.0f, 0.0f);
    float f = f_fma(eta, ni, pt_sqrt(k));
    return V(f_fma(eta, i.x, -(f * n.x)), f_fma(eta, i.y, -(f * n.y)), f_fma(eta, i.z, -(f * n.z)));
}

// compute.glsl:184-224 BSDF: picks the next ray, returns its probability
PT_DEV float bsdf(v3 &ro, v3 &rd, const Hit &h, bool &isRefractive, uint32_t &seed)
{
    isRefractive = false;
    float spec = h.m.specularChance, refr = h.m.refractionChance;
    if (spec > 0.0f) {
        float n1 = h.fromInside ? h.m.ior : 1.0f, n2 = !h.fromInside ? h.m.ior : 1.0f;
        spec = f_mix(spec, 1.0f, fresnel_schlick(v_dot(v_neg(rd), h.normal), n1, n2));
        float diffuse = 1.0f - spec - refr;
        refr = 1.0f - spec - diffuse;
    }
    v3 diffuseRay = cosine_sample_hemisphere(h.normal, seed);
    float prob;
    float roll = rand01(seed);
    // the specular and the refractive lobe both end in normalize(mix(...)); the mix is evaluated per lobe and the
    // normalisation once for whichever lobe the lane took (same arithmetic per lane, one code instance per wave)
    v3 raw = diffuseRay;
    bool lobe = false;
    if (spec > roll) {
        v3 refl = f_reflect(rd, h.normal);
        raw = v_mix(refl, diffuseRay, h.m.specularRoughness * h.m.specularRoughness);
        prob = spec;
        lobe = true;
    } else if (spec + refr > roll) {
        v3 rf = f_refract(rd, h.normal, h.fromInside ? h.m.ior : f_rcp(h.m.ior));
        v3 rough = cosine_sample_hemisphere(v_neg(h.normal), seed);
        raw = v_mix(rf, rough, h.m.refractionRoughness * h.m.refractionRoughness);
        prob = refr;
        isRefractive = true;
        lobe = true;
    } else {
        prob = 1.0f - spec - refr;
    }
    rd = lobe ? v_normalize(raw) : raw;
    ro = v_fma(rd, EPSILON, h.nearHitPos);
    return f_max(prob, EPSILON);
}

// One iteration of Radiance's bounce loop (compute.glsl:140-180) for one path.  Returns true when the path
// continues (hit, survived Russian roulette), false when it ended (miss -> environment, or roulette kill).
template <bool MASKED, bool MATLDS, bool GRID = false, bool SHARE = false>
PT_DEV bool bounce_step_t(const SceneLds &sc, int ns, int nc, const EnvRef &env, v3 &ro, v3 &rd, v3 &throughput, v3 &rad,
                          uint32_t &seed, const unsigned long long *masks, float &walkFrom PROF_PARAM)
{
    Hit h;
    const bool hit = ray_trace_t<MASKED, MATLDS, GRID, SHARE>(sc, ns, nc, ro, rd, h, masks, walkFrom PROF_PASS);
    if constexpr (GRID) {
        if (walkFrom >= 0.0f) return true; // the walk was cut short (WALK SLICES): the path is unchanged, the caller repeats this bounce
    }
    if (hit) {
        PROF_BEGIN
        if (h.fromInside) { // Beer's law, compute.glsl:145-149
            h.normal = v_neg(h.normal);
            throughput.x *= pt_exp<SHARE>(-h.m.absorbance.x * h.T);
            throughput.y *= pt_exp<SHARE>(-h.m.absorbance.y * h.T);
            throughput.z *= pt_exp<SHARE>(-h.m.absorbance.z * h.T);
        }
        PROF_MARK(4) // Beer
        bool isRefractive;
        float prob = bsdf(ro, rd, h, isRefractive, seed);
        PROF_MARK(5) // BSDF
        rad = V(f_fma(h.m.emissiv.x, throughput.x, rad.x), f_fma(h.m.emissiv.y, throughput.y, rad.y),
                f_fma(h.m.emissiv.z, throughput.z, rad.z));
        if (!isRefractive) throughput = v_mul(throughput, h.m.albedo);
        throughput = v_scale(throughput, f_rcp(prob));
        float p = f_max(throughput.x, f_max(throughput.y, throughput.z)); // Russian roulette, :167-173
        if (rand01(seed) > p) return false;
        throughput = v_scale(throughput, f_rcp(p));
        return true;
    }
    PROF_BEGIN
    v3 e;
    if (env.data == nullptr) { // persistent kernel: fetch the environment descriptor where it is needed (see cold_args)
        ColdArgs ca = cold_args();
        EnvRef cold{ca->env, env.lut, ca->envSize, ca->envFormat};
        e = sample_env(cold, rd); // compute.glsl:177
    } else {
        e = sample_env(env, rd);
    }
    rad = V(f_fma(e.x, throughput.x, rad.x), f_fma(e.y, throughput.y, rad.y), f_fma(e.z, throughput.z, rad.z));
    PROF_MARK(6) // miss shading
    return false;
}
PT_DEV bool bounce_step(const SceneLds &sc, int ns, int nc, const EnvRef &env, v3 &ro, v3 &rd, v3 &throughput, v3 &rad,
                        uint32_t &seed PROF_PARAM)
{
    float fresh = -1.0f;
    return bounce_step_t<false, true>(sc, ns, nc, env, ro, rd, throughput, rad, seed, nullptr, fresh PROF_PASS);
}

// compute.glsl:132-182 Radiance
PT_DEV v3 radiance(const FrameArgs &a, const SceneLds &sc, const EnvRef &env, v3 ro, v3 rd, uint32_t &seed)
{
#ifdef PT_PROFILE
    unsigned long long prof_dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    v3 throughput = V(1.0f, 1.0f, 1.0f), rad = V(0.0f, 0.0f, 0.0f);
    for (int i = 0; i < a.rayDepth; i++)
        if (!bounce_step(sc, a.numSpheres, a.numCuboids, env, ro, rd, throughput, rad, seed PROF_DUMMY)) break;
    return rad;
}

// GLSL mat4 * vec4 on the column-major view of the UBO bytes: m[4c + r]
template <typename FP>
PT_DEV void mat_vec(FP m, float x, float y, float z, float w, float *out)
{
#pragma unroll
    for (int r = 0; r < 4; r++) out[r] = f_fma(m[12 + r], w, f_fma(m[8 + r], z, f_fma(m[4 + r], y, m[r] * x)));
}

// compute.glsl:113-121: sub-pixel jitter, GetWorldSpaceRay (:352-357), thin lens (UniformSampleUnitCircle :309-314).
// Consumes 4 RNG draws.
// The camera block (invProj[16], invView[16], viewPos[3], focalLength, apertureDiameter = the first 37 floats of
// FrameArgs) is read through `cam`.  The persistent kernel passes a pointer into its kernarg segment that is made
// opaque once per ring refill, so these 37 scalars are s_load-ed where they are used instead of being kept live in
// SGPRs across the whole bounce loop (which spilled SGPRs into VGPR lanes).
template <typename FP>
PT_DEV void primary_ray_cam(FP cam, float invW, float invH, int px, int py, uint32_t &seed, v3 &ro, v3 &rd)
{
    FP invProj = cam, invView = cam + 16;
    v3 viewPos = V(cam[32], cam[33], cam[34]);
    float u0 = rand01(seed), u1 = rand01(seed); // :113
    float ndcx = f_fma(((float)px + u0) * invW, 2.0f, -1.0f); // uniform 1/W, 1/H (IEEE quotients)
    float ndcy = f_fma(((float)py + u1) * invH, 2.0f, -1.0f);
    float eye[4], wd[4];
    mat_vec(invProj, ndcx, ndcy, -1.0f, 0.0f, eye);
    mat_vec(invView, eye[0], eye[1], -1.0f, 0.0f, wd);
    v3 dir = v_normalize(V(wd[0], wd[1], wd[2]));
    v3 focal = v_fma(dir, cam[35], viewPos); // :117
    float angle = rand01(seed) * 2.0f * PI;
    float rr = pt_sqrt(rand01(seed));
    float sn, cs;
    pt_sincos(angle, sn, cs);
    float half_ap = cam[36] * 0.5f;
    float ox = half_ap * (cs * rr), oy = half_ap * (sn * rr);
    float org[4];
    mat_vec(invView, ox, oy, 0.0f, 1.0f, org); // :120
    ro = V(org[0], org[1], org[2]);
    rd = v_normalize(v_sub(focal, ro));
}

PT_DEV void primary_ray(const FrameArgs &a, int px, int py, uint32_t &seed, v3 &ro, v3 &rd)
{
    primary_ray_cam<const float *>(a.invProj, a.invW, a.invH, px, py, seed, ro, rd);
}

// image row of local row `ly` of this launch (contiguous row block, or block-cyclic bands across GPUs)
PT_DEV int global_row_v(int bandRows, int bandWorld, int bandRank, int localRow0, int y0, int ly)
{
    if (bandRows == 0) return y0 + ly;
    int l = localRow0 + ly;
    int band = l / bandRows;
    return (band * bandWorld + bandRank) * bandRows + (l - band * bandRows);
}
PT_DEV int global_row(const FrameArgs &a, int ly) { return global_row_v(a.bandRows, a.bandWorld, a.bandRank, a.localRow0, a.y0, ly); }

PT_DEV uint32_t pixel_seed(int px, int py, int frame)
{
    return ((uint32_t)px * 1973u + (uint32_t)py * 9277u + (uint32_t)frame * 2699u) | 1u; // compute.glsl:106
}

// compute.glsl:125-129: irradiance /= SPP; running mean with the previous accumulation value; alpha = 1
PT_DEV float4 resolve_pixel(const FrameArgs &a, v3 irr, float4 last)
{
    irr = v_scale(irr, f_div_ieee(1.0f, (float)a.spp)); // uniform reciprocal
    float w = f_div_ieee(1.0f, (float)(a.frame + 1));
    return make_float4(f_mix(last.x, irr.x, w), f_mix(last.y, irr.y, w), f_mix(last.z, irr.z, w), 1.0f);
}

// compute.glsl:101-130 main for one pixel: returns the new accumulation value
PT_DEV float4 shade_pixel(const FrameArgs &a, const SceneLds &sc, const EnvRef &env, int px, int py, float4 last)
{
    uint32_t seed = pixel_seed(px, py, a.frame);
    v3 irr = V(0.0f, 0.0f, 0.0f);
    for (int s = 0; s < a.spp; s++) {
        v3 ro, rd;
        primary_ray(a, px, py, seed, ro, rd);
        irr = v_add(irr, radiance(a, sc, env, ro, rd, seed));
    }
    return resolve_pixel(a, irr, last);
}

} // namespace pt
