// pt_sphere_grid.hpp — host-side build of the uniform sphere grid that the generic bounce of large scenes walks
// (ray_trace_t<GRID> in pt_device.hpp has the traversal and the argument why its result equals the reference's in-order loop,
// compute.glsl:226-258).  The reference has no acceleration structure: every ray tests every sphere; at 256 spheres that is
// 2,816 of the ~4,000 vector instructions of a bounce.  The grid only decides WHICH spheres a ray tests, never how.
//
// Built on the host from a shadow of the std140 GameObjectsUBO whenever the scene changed (<= 256 spheres: microseconds),
// uploaded stream-ordered like the scene itself.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pt_tuning.hpp"

namespace ptgrid {

constexpr int kMinSpheres = 64;   // below this the in-order loop is as fast (default scene: 48 spheres)
constexpr int kMaxCells = 256;    // uint16 starts: 514 bytes of LDS
constexpr int kMaxRefs = 1024;    // uint8 refs: LDS budget (scene + rings + parked list leave ~1.5 KB at 6 workgroups per CU)

struct SphereGrid {
    bool valid = false;
    int dims[3] = {1, 1, 1};
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, cell[3] = {0, 0, 0}, invCell[3] = {0, 0, 0}, center[3] = {0, 0, 0};
    float reach2 = 0.0f;
    std::vector<unsigned char> packed; // uint16 starts[cells + 1], uint8 refs[], padded to a multiple of 4 bytes
    int numRefs = 0;
};

// Sphere runs of the in-order loop (FrameArgs::sphereRunStart): bit i = sphere i starts a new run, i.e. its centre's x or z differs
// BITWISE from sphere i - 1's (a scene laid out on a lattice, like the reference's LoadScene, MainWindow.cs:208-244, repeats
// coordinates in consecutive spheres).  Equal bits give equal differences and products, so sharing them changes no result.
inline void sphere_runs(const float *objects, int ns, unsigned long long out[4])
{
    for (int w = 0; w < 4; w++) out[w] = ~0ull; // (spheres beyond ns are never visited)
    for (int i = 1; i < ns && i < 256; i++) {
        uint32_t a[3], b[3];
        std::memcpy(a, objects + 20 * (i - 1), 12);
        std::memcpy(b, objects + 20 * i, 12);
        if (a[0] == b[0] && a[2] == b[2]) out[i >> 6] &= ~(1ull << (i & 63));
    }
}

// objects: the 26,624-byte std140 block (sphere i: centre.xyz, radius at float 20 * i).
inline SphereGrid build(const float *objects, int ns)
{
    SphereGrid g;
    const int minSpheres = pt::tuning().gridMinSpheres; // (kMinSpheres unless a tuning run changed it)
    if (ns < minSpheres || ns < 2 || ns > 256) return g;
    double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300}, maxAbs = 0.0;
    std::vector<double> rad(ns);
    for (int i = 0; i < ns; i++) {
        const float *s = objects + 20 * i;
        for (int k = 0; k < 4; k++)
            if (!std::isfinite(s[k])) return g; // (a sphere that can never be hit would be fine to skip, but keep the rule simple)
        rad[i] = std::fabs((double)s[3]); // the intersection only uses radius^2
        for (int k = 0; k < 3; k++) {
            blo[k] = std::min(blo[k], (double)s[k] - rad[i]);
            bhi[k] = std::max(bhi[k], (double)s[k] + rad[i]);
            maxAbs = std::max(maxAbs, std::fabs((double)s[k]) + rad[i]);
        }
    }
    double diag2 = 0.0;
    for (int k = 0; k < 3; k++) diag2 += (bhi[k] - blo[k]) * (bhi[k] - blo[k]);
    const double Rg = 0.5 * std::sqrt(diag2); // half diagonal of the spheres' bounding box
    if (!(Rg > 0.0) || !(Rg < 1e15) || !(maxAbs < 1e15)) return g;
    // A ray may use the grid when its origin is within `reach` of the box centre: then |o - c| <= D = reach + Rg for every
    // sphere.  How far can the fp32 hit point P = o + d * t1 of such a ray lie from the sphere it belongs to?  With
    // q(t) = |o + d t - c|^2 - r^2 = (t - t1*)(t - t2*), the computed root satisfies |q(t1)| <= 2 err, where err bounds the
    // rounding error of the discriminant b*b - c: <= 1.2e-6 D^2 (three-term dot products, the fused b*b - c, the 2-ulp
    // square root) + 2.5e-7 D M from forming o - c at coordinate magnitude M.  So | |P - c| - r | <= min(2 err / r, sqrt(2 err)):
    // every sphere's box is inflated by twice that (and the same bound covers an origin that is classified as inside a
    // sphere it touches from outside), plus a slack for the fp32 cell arithmetic of the walk itself.
    const double reach = 2.0 * Rg, D = reach + Rg, M = maxAbs + reach;
    const double err = 1.2e-6 * D * D + 2.5e-7 * D * M;
    const double walkSlack = 1e-5 * (D + M);
    std::vector<double> margin(ns);
    double maxMargin = 0.0;
    for (int i = 0; i < ns; i++) {
        const double radial = rad[i] > 0.0 ? std::min(2.0 * err / rad[i], std::sqrt(2.0 * err)) : std::sqrt(2.0 * err);
        margin[i] = 2.0 * radial + walkSlack;
        maxMargin = std::max(maxMargin, margin[i]);
    }
    for (int k = 0; k < 3; k++) { // the box holds every inflated sphere box
        blo[k] = 1e300;
        bhi[k] = -1e300;
    }
    for (int i = 0; i < ns; i++) {
        const float *s = objects + 20 * i;
        for (int k = 0; k < 3; k++) {
            blo[k] = std::min(blo[k], (double)s[k] - rad[i] - margin[i]);
            bhi[k] = std::max(bhi[k], (double)s[k] + rad[i] + margin[i]);
        }
    }
    double ext[3], vol = 1.0;
    for (int k = 0; k < 3; k++) {
        g.lo[k] = (float)(blo[k] - walkSlack); // (fp32 rounding may move a face by half an ulp: inside the slack)
        g.hi[k] = (float)(bhi[k] + walkSlack);
        ext[k] = (double)g.hi[k] - (double)g.lo[k];
        if (!(ext[k] > 0.0)) return g;
        vol *= ext[k];
        g.center[k] = (float)(0.5 * (blo[k] + bhi[k])); // (of the inflated box: within maxMargin of the spheres' own centre)
    }
    const double reachIn = std::max(0.0, reach - maxMargin) * 0.99; // (compared against an fp32 distance^2: stay inside the analysed range)
    g.reach2 = (float)(reachIn * reachIn);
    const int cellTarget = pt::tuning().gridCells; // (kMaxCells unless a tuning run changed it)
    const int maxCells = std::max(1, std::min(cellTarget, kMaxCells));
    // about 5 cells per 8 spheres (measured on the 256-sphere scene: 8 x 5 x 4 cells beat both coarser and finer grids), cells as
    // cubic as the box allows
    const double side = std::cbrt(vol / std::max(1, std::min(ns * 5 / 8, maxCells)));
    for (int k = 0; k < 3; k++) g.dims[k] = std::max(1, std::min(32, (int)(ext[k] / side + 0.5)));
    while (g.dims[0] * g.dims[1] * g.dims[2] > maxCells) {
        int big = 0;
        for (int k = 1; k < 3; k++)
            if (g.dims[k] > g.dims[big]) big = k;
        g.dims[big]--;
    }
    if (const int *td = pt::tuning().gridDims; td[0] > 0 && td[1] > 0 && td[2] > 0 && td[0] * td[1] * td[2] <= kMaxCells) { // tuning runs
        g.dims[0] = td[0]; g.dims[1] = td[1]; g.dims[2] = td[2];
    }
    for (int k = 0; k < 3; k++) {
        g.cell[k] = (float)(ext[k] / g.dims[k]);
        g.invCell[k] = (float)(g.dims[k] / ext[k]);
    }
    const int cells = g.dims[0] * g.dims[1] * g.dims[2];
    // cell range of every sphere's inflated box
    std::vector<int> range(6 * ns);
    std::vector<int> count(cells + 1, 0);
    for (int i = 0; i < ns; i++) {
        const float *s = objects + 20 * i;
        for (int k = 0; k < 3; k++) {
            const double a = ((double)s[k] - rad[i] - margin[i] - (double)g.lo[k]) * g.dims[k] / ext[k];
            const double b = ((double)s[k] + rad[i] + margin[i] - (double)g.lo[k]) * g.dims[k] / ext[k];
            range[6 * i + 2 * k] = std::max(0, std::min(g.dims[k] - 1, (int)std::floor(a)));
            range[6 * i + 2 * k + 1] = std::max(0, std::min(g.dims[k] - 1, (int)std::floor(b)));
        }
        for (int z = range[6 * i + 4]; z <= range[6 * i + 5]; z++)
            for (int y = range[6 * i + 2]; y <= range[6 * i + 3]; y++)
                for (int x = range[6 * i]; x <= range[6 * i + 1]; x++) count[(z * g.dims[1] + y) * g.dims[0] + x + 1]++;
    }
    for (int c = 0; c < cells; c++) count[c + 1] += count[c];
    g.numRefs = count[cells];
    if (g.numRefs > kMaxRefs) return g; // big overlapping spheres: every cell lists everything, the in-order loop is the better plan
    const size_t bytes = ((size_t)(cells + 1) * 2 + (size_t)g.numRefs + 3) & ~(size_t)3;
    g.packed.assign(bytes, 0);
    uint16_t *starts = (uint16_t *)g.packed.data();
    unsigned char *refs = g.packed.data() + (size_t)(cells + 1) * 2;
    for (int c = 0; c <= cells; c++) starts[c] = (uint16_t)count[c];
    std::vector<int> fill(count.begin(), count.end() - 1);
    for (int i = 0; i < ns; i++) // ascending sphere index inside every cell
        for (int z = range[6 * i + 4]; z <= range[6 * i + 5]; z++)
            for (int y = range[6 * i + 2]; y <= range[6 * i + 3]; y++)
                for (int x = range[6 * i]; x <= range[6 * i + 1]; x++) refs[fill[(z * g.dims[1] + y) * g.dims[0] + x]++] = (unsigned char)i;
    g.valid = true;
    return g;
}

} // namespace ptgrid
