// pt_kernels.hip — hand-written HIP kernels for gfx950 (CDNA4): the path-tracing integrator and the
// atmosphere environment precompute.  Written from the algorithm, not transpiled: the control structure, data
// staging and thread mapping are designed for wave64 / LDS / 8 XCDs.
//
// What it computes (per pixel, per frame) is the reference's compute shader
//   /root/reference/OpenTK-PathTracer/res/shaders/PathTracing/compute.glsl:101-369
// dispatched by src/Render/PathTracer.cs:114-123; each device function cites the GLSL lines it implements.
//
// Mapping
//   * one wavefront (64 lanes) = one 8x8 pixel tile (the reference's 8x8 workgroup, compute.glsl:8);
//     a 256-thread workgroup = 4 tiles, so the scene is staged into LDS once per 256 pixels;
//   * the scene (std140 GameObjectsUBO) is re-packed while staging: sphere geometry as one float4 (c.xyz, r),
//     cuboid geometry as two float4, materials as 4 x float4 — the traversal loop index is wave-uniform, so
//     every LDS read in the hot loop is a conflict-free broadcast ds_read_b128;
//   * camera + parameters travel in the kernel argument (SGPRs);
//   * workgroup -> tile mapping is XCD-aware: consecutive workgroup ids are dealt round-robin to the 8 XCDs by
//     the dispatcher, so id b is remapped to a contiguous band of tiles per XCD (neighbouring tiles share
//     environment-map texels in that XCD's L2);
//   * the accumulation image is read and written as one float4 per lane: a wave covers 8 rows x 128 B segments.
//
// Arithmetic: the "pt-f32" contract of pt_math.hpp (bit-identical to oracle/pt_oracle.c).
// Build flags (see __graft_entry__.build): -O3 -ffp-contract=off -fno-fast-math --offload-arch=gfx950
#include "pt_atmosphere.hpp"
#include "pt_device.hpp"
#include "pt_kernels.hpp"
#include "pt_math.hpp"
#include "pt_tuning.hpp"

namespace pt {

// ---------------------------------------------------------------------------------------------- kernels
extern __shared__ float4 g_lds[];

// ---- hand-over audit and chaos injection (tools/handover_stress.cpp; compiled out of the product library).
// PT_AUDIT: every read-modify-write of an accumulation pixel (compute.glsl:126-129) is mirrored by ONE device-scope atomic
// exchange on a 64-bit side word per pixel: (frames folded so far) << 32 | hash(colour stored).  The exchange returns what the
// previous resolve of that pixel left there, so a resolve that ran out of order (frame f before f-1, or twice), or that folded
// into a colour other than the one the previous resolve stored (a stale or torn 16-byte read), is caught the moment it happens,
// independently of the alpha tags the product protocol relies on.  All-ones = history unknown (after a clear / reset / restore).
// PT_CHAOS: pseudo-random s_sleep delays (0.4 us ... 100 us) at the protocol's decision points, to widen every race window.
#ifdef PT_AUDIT
PT_DEV uint32_t audit_hash(float x, float y, float z)
{
    uint32_t a = __float_as_uint(x), b = __float_as_uint(y), c = __float_as_uint(z);
    uint32_t h = a * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + b * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + c * 0xC2B2AE3Du;
    return h ^ (h >> 16);
}
// `p`: pixel index relative to a.accum; F: absolute frame being folded; `last`: the value that was loaded; `next`: the value
// about to be stored; site: which resolve site of which kernel (for the log)
PT_DEV void audit_resolve(const FrameArgs &a, size_t p, int F, float4 last, float4 next, int site)
{
    if (!a.audit) return;
    const unsigned long long now = ((unsigned long long)(uint32_t)(F + 1) << 32) | audit_hash(next.x, next.y, next.z);
    const unsigned long long old = atomicExch(a.audit + p, now);
    if (old == ~0ull) return;
    const uint32_t oldFrames = (uint32_t)(old >> 32), oldHash = (uint32_t)old, lastHash = audit_hash(last.x, last.y, last.z);
    if (oldFrames == (uint32_t)F && (F == 0 || oldHash == lastHash)) return;
    const unsigned int slot = atomicAdd(a.auditLog, 1u);
    if (slot >= (unsigned int)kAuditLogRecords) return;
    unsigned int *r = a.auditLog + 4 + slot * kAuditRecordWords;
    r[0] = (unsigned int)site | (oldFrames != (uint32_t)F ? 0x100u : 0u) | (oldHash != lastHash ? 0x200u : 0u);
    r[1] = (unsigned int)p;
    r[2] = (unsigned int)F;
    r[3] = oldFrames;
    r[4] = oldHash;
    r[5] = lastHash;
    r[6] = __float_as_uint(last.w);
    r[7] = a.launchSeq;
    r[8] = (unsigned int)a.frame | ((unsigned int)a.batchFrames << 24);
    r[9] = __float_as_uint(a.chainTag);
    r[10] = blockIdx.x;
    r[11] = (unsigned int)a.tagged | ((unsigned int)a.keepTags << 1) | ((unsigned int)a.variant << 8);
}
#define AUDIT_RESOLVE(a, p, F, last, next, site) audit_resolve(a, p, F, last, next, site)
#define AUDIT_SABOTAGED(a, pix, fj) ((a).auditSabotage > 0 && ((unsigned int)(pix) * 2654435761u + (unsigned int)(fj) * 40503u) % (unsigned int)(a).auditSabotage == 0u)
#else
#define AUDIT_RESOLVE(a, p, F, last, next, site)
#define AUDIT_SABOTAGED(a, pix, fj) false
#endif

#ifdef PT_CHAOS
// stateless: the wavefront's cycle counter hashed with the site; 3/4 of the calls do nothing, 3/16 sleep 0.4 - 3 us, 1/16 up to 100 us
PT_DEV void chaos_point(unsigned int site)
{
    uint32_t r = (uint32_t)__builtin_readcyclecounter();
    r = (r ^ (r >> 7)) * 0x9E3779B1u + site * 0x85EBCA6Bu;
    r ^= r >> 15;
    r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
    if ((r & 3u) != 0u) return;
    const bool longNap = ((r >> 2) & 3u) == 0u;
    const int n = (int)((r >> 4) & (longNap ? 31u : 7u)) + 1;
    for (int i = 0; i < n; i++) {
        if (longNap) __builtin_amdgcn_s_sleep(127);
        else __builtin_amdgcn_s_sleep(16);
    }
}
#define CHAOS(site) chaos_point(site)
#else
#define CHAOS(site)
#endif

// Stage + re-pack the scene into LDS (all 256 threads): std140 Sphere = 5 x float4 (geometry, 4 x material),
// Cuboid = 6 x float4.  Ends with a workgroup barrier.
PT_DEV SceneLds stage_scene(const FrameArgs &a)
{
    const int ns = a.numSpheres, nc = a.numCuboids;
    float4 *sph = g_lds;
    float4 *cmin = sph + ns;
    float4 *cmax = cmin + nc;
    float4 *mat = cmax + nc;
    const bool matInLds = a.materialsInLds != 0;
    float *invr = (float *)(mat + (matInLds ? 4 * (ns + nc) : 0));
    float *lut = invr + ((ns + 3) & ~3);
    const int tid = threadIdx.x;
    const float4 *obj = (const float4 *)a.objects;
    const int nthreads = blockDim.x;
    if (matInLds) {
        for (int i = tid; i < ns * 5; i += nthreads) {
            int s = i / 5, part = i - s * 5;
            float4 v = obj[i];
            if (part == 0) {
                sph[s] = v;
                invr[s] = f_div_ieee(1.0f, v.w);
            } else {
                mat[4 * s + part - 1] = v;
            }
        }
        for (int i = tid; i < nc * 6; i += nthreads) {
            int c = i / 6, part = i - c * 6;
            float4 v = obj[1280 + i]; // Cuboids[] start at byte 20480 = float4 index 1280
            if (part == 0) cmin[c] = v;
            else if (part == 1) cmax[c] = v;
            else mat[4 * (ns + c) + part - 2] = v;
        }
    } else { // geometry only
        for (int i = tid; i < ns; i += nthreads) {
            float4 v = obj[5 * i];
            sph[i] = v;
            invr[i] = f_div_ieee(1.0f, v.w);
        }
        for (int i = tid; i < nc * 2; i += nthreads) {
            int c = i >> 1;
            float4 v = obj[1280 + 6 * c + (i & 1)];
            if (i & 1) cmax[c] = v;
            else cmin[c] = v;
        }
    }
    if (a.envFormat == 1 && tid < 256) lut[tid] = a.srgbLut[tid];
    // sphere grid of large scenes (only when this launch traverses it): packed uint16 starts + uint8 refs, copied word by word
    unsigned int *grid = (unsigned int *)(lut + (a.envFormat == 1 ? 256 : 0));
    const unsigned short *gridStarts = nullptr;
    const unsigned char *gridRefs = nullptr;
    if (a.gridLdsBytes > 0) {
        const unsigned int *src = (const unsigned int *)a.grid;
        for (int i = tid; i < (a.gridBytes + 3) / 4; i += nthreads) grid[i] = src[i];
        gridStarts = (const unsigned short *)grid;
        gridRefs = (const unsigned char *)(gridStarts + a.gridDims[0] * a.gridDims[1] * a.gridDims[2] + 1);
    }
    __syncthreads();
    return SceneLds{sph, cmin, cmax, mat, invr, lut, obj, gridStarts, gridRefs};
}

// XCD-aware workgroup id: the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, so id b is
// remapped to a contiguous band of work per XCD (the tail nwg & 7 keeps its identity mapping).
PT_DEV int xcd_band_id(int b, int nwg)
{
    int per = nwg >> 3;
    return b < per * 8 ? (b & 7) * per + (b >> 3) : b;
}

// ---- variant 1: one wavefront = one 8x8 tile, one pixel per lane, the wave runs until its longest path ends
__global__ __launch_bounds__(256) void pt_integrate_kernel(const FrameArgs a)
{
    SceneLds sc = stage_scene(a);
    EnvRef env{a.env, (LdsFloats)sc.lut, a.envSize, a.envFormat};
    const int tid = threadIdx.x;
    const int b = xcd_band_id(blockIdx.x, gridDim.x);
    const int wave = tid >> 6, lane = tid & 63;
    const int tile = b * 4 + wave;
    if (tile >= a.tilesX * a.tilesY) return;
    const int tx = tile % a.tilesX, ty = tile / a.tilesX;
    const int px = tx * 8 + (lane & 7);
    const int ly = ty * 8 + (lane >> 3); // row inside this GPU's row block
    if (px >= a.width || ly >= a.rows) return;
    const size_t idx = (size_t)ly * a.width + px;
    float4 last = a.accum[idx];                                  // imageLoad  (compute.glsl:126)
    const float4 next = shade_pixel(a, sc, env, px, global_row(a, ly), last);
    AUDIT_RESOLVE(a, idx, a.frame, last, next, 1);
    a.accum[idx] = next;                                         // imageStore (compute.glsl:129)
}

// ---- variants 2..6: wave-level pixel pool with path regeneration.
// Russian roulette and environment misses end paths after very different numbers of bounces (mean 2.7 of 8 in the
// default scene), so a wave that keeps one pixel per lane idles most lanes most of the time.  Here a wavefront
// owns a pool of POOL consecutive 8x8 tiles; whenever a lane's pixel is finished it takes the next pixel of the
// pool (ballot + prefix count, no atomics), so the traversal loops run with (almost) all 64 lanes busy.  Every
// pixel still owns its RNG stream (seeded by its global coordinate, compute.glsl:106) and the samples of a pixel
// stay on one lane in order, so the image is bit-identical to variant 1.
__global__ __launch_bounds__(256) void pt_integrate_pool_kernel(const FrameArgs a, const int poolTiles)
{
    SceneLds sc = stage_scene(a);
    EnvRef env{a.env, (LdsFloats)sc.lut, a.envSize, a.envFormat};
    const int tid = threadIdx.x;
    const int b = xcd_band_id(blockIdx.x, gridDim.x);
    const int wave = tid >> 6;
    const int numTiles = a.tilesX * a.tilesY;
    const int pool = b * 4 + wave;
    int next = pool * poolTiles * 64;                               // wave-uniform cursor into the pool
    int tileEnd = (pool + 1) * poolTiles;
    const int poolEnd = (tileEnd < numTiles ? tileEnd : numTiles) * 64;
    if (next >= poolEnd) return;

#ifdef PT_PROFILE
    unsigned long long prof_dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // per-lane path state
    int pix = -1;           // linear index into accum, -1 = lane has no pixel
    int px = 0, py = 0, sample = 0, bounce = 0;
    bool needRay = false;
    uint32_t seed = 0;
    v3 ro = V(0, 0, 0), rd = V(0, 0, 1), throughput = V(1, 1, 1), rad = V(0, 0, 0), irr = V(0, 0, 0);

    for (;;) {
        // ---- refill idle lanes from the pool
        bool idle = pix < 0;
        unsigned long long m = __ballot(idle);
        if (m != 0ull && next < poolEnd) {
            int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            int cand = next + rank;
            if (idle && cand < poolEnd) {
                int tile = cand >> 6, q = cand & 63;
                int tx = tile % a.tilesX, ty = tile / a.tilesX;
                int x = tx * 8 + (q & 7), ly = ty * 8 + (q >> 3);
                if (x < a.width && ly < a.rows) { // ragged right/bottom tiles: skip the pixel, stay idle
                    px = x;
                    py = global_row(a, ly);
                    pix = ly * a.width + x;
                    seed = pixel_seed(px, py, a.frame);
                    sample = 0;
                    irr = V(0.0f, 0.0f, 0.0f);
                    needRay = true;
                }
            }
            next += __builtin_popcountll(m);
        }
        bool active = pix >= 0;
        if (!__any(active)) {
            if (next >= poolEnd) break;
            continue;
        }
        // ---- (re)generate the primary ray of the lane's current sample
        if (active && needRay) {
            primary_ray(a, px, py, seed, ro, rd);
            throughput = V(1.0f, 1.0f, 1.0f);
            rad = V(0.0f, 0.0f, 0.0f);
            bounce = 0;
            needRay = false;
        }
        // ---- one bounce for every active lane
        if (active) {
            bool cont = false;
            if (bounce < a.rayDepth) cont = bounce_step(sc, a.numSpheres, a.numCuboids, env, ro, rd, throughput, rad, seed PROF_DUMMY);
            bounce++;
            if (!cont || bounce >= a.rayDepth) {
                irr = v_add(irr, rad);
                sample++;
                if (sample < a.spp) {
                    needRay = true;
                } else {
                    float4 last = a.accum[pix];
                    const float4 next = resolve_pixel(a, irr, last);
                    AUDIT_RESOLVE(a, (size_t)pix, a.frame, last, next, 2);
                    a.accum[pix] = next;
                    pix = -1;
                }
            }
        }
    }
}

// ---- frame pipelining (persistent spp = 1 kernels).  One launch can render a BATCH of consecutive frames: its tile queue
// runs over (frame, tile) pairs, frame-major, so wavefronts only drain once per batch instead of once per frame (the
// drain tail is ~75 us of a ~215 us frame at 1080p).  The only dependency between frames is per pixel: the running
// mean of frame f+1 needs the pixel's value after frame f (compute.glsl:126-129).  It is carried IN the pixel: inside a
// batch, frame j of the batch stores alpha = FRAME_TAG + j instead of 1 (the last frame of the batch stores the 1 the
// reference stores), and the resolve of frame j only proceeds when it reads the tag of frame j-1.  Pixels are written
// with ONE 16-byte device-scope (sc1) store and read with ONE 16-byte sc1 load — single-copy atomic and coherent
// across the 8 XCD L2s — so colour and tag always belong together.  A resolve that finds its predecessor missing is
// simply retried in the wavefront's next iteration (never a spin loop: the predecessor may live in another lane of
// the same wavefront); after FRAME_RETRY_LIMIT attempts it proceeds anyway and raises the launch's error word.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float FRAME_TAG = 2.0f;
// the tag of (absolute) frame f: distinct for any two frames that can be in flight together, exact in binary32.  Launches
// may CHAIN: the first frame of a tagged launch waits for the tag of the previous launch's last frame (FrameArgs::chainTag), so
// two launches on different streams overlap like the frames inside one launch do (the second fills the wavefront slots the
// first one's drain frees) — the host restores alpha = 1 before anything can observe the image (pt_set_alpha_kernel).
PT_DEV float frame_tag(int absFrame) { return FRAME_TAG + (float)(absFrame & 1023); }
constexpr int FRAME_RETRY_LIMIT = 1 << 22;
constexpr int MAX_BATCH_FRAMES = 256; // (one workgroup fills the weight table: <= its 256 threads; tags cover 1,024 frames)

PT_DEV float4 load_pixel_sc1(const float4 *p)
{
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
PT_DEV void store_pixel_sc1(float4 *p, float4 c)
{
    f32x4 v = {c.x, c.y, c.z, c.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

// ---- variant 0 (default) and >= 10: persistent wavefronts + two-level tile queue + per-wavefront LDS ring.
// The grid is sized to the machine (blocksPerCU x CUs), not to the image.  Each wavefront repeatedly
//   1. takes an 8x8 tile — or, in a pipelined batch, a (frame, tile) pair — from its workgroup's queue: an LDS
//      (cursor,end) pair advanced with one 64-bit LDS atomic per tile; when the pair runs dry ONE wavefront of the
//      workgroup refills it with a chunk of a.queueChunk tiles from the global counter (one device atomic per
//      queueChunk x 64 pixels; with one frame per launch the workgroup's first chunk is static),
//   2. spp = 1: runs the TILE PASS — the tile's 64 primary rays and their whole first bounce with all lanes, spheres
//      culled against the tile's ray bundle — and parks the surviving paths in its LDS ring (PathEntry);
//      spp > 1: generates the tile's 64 primary rays with all lanes into the ring (RingEntry),
//   3. runs bounce iterations in which every lane whose path ended resolves its pixel and pops the next ring entry.
// So the traversal loops always run (nearly) full, the camera code is never executed divergently (for spp = 1),
// work is balanced dynamically across the chip, and the only tail is the drain at the end of the launch (once per
// batch of frames, see "frame pipelining" above).  Pixels keep their own RNG streams -> bit-identical to every variant.
struct RingEntry { // 40 bytes (spp > 1)
    int pix;       // linear index into accum, -1 = pixel outside the image (ragged tile)
    int pxy;       // px | py << 16 (global coordinates)
    uint32_t seed; // RNG state after the primary-ray draws
    float ox, oy, oz, dx, dy, dz;
    int pad;       // frame of the batch (frame pipelining)
};

// spp == 1 kernels: the ring holds paths AFTER their first bounce (see the tile pass in the kernel), 60 bytes each
struct PathEntry {
    int pix;       // linear index into accum
    int bounce;    // bounces done so far | frame of the batch << 16 | bit 30: `last` holds the pixel's current value
    uint32_t seed; // RNG state
    float ro[3], rd[3], thr[3], rad[3];
#ifdef PT_CARRY_LAST
    // The pixel's accumulation value, read by the TILE PASS with all 64 lanes (8 rows x 128 B: full lines) while the first bounce
    // computes, and carried with the path: its resolve then needs no load (and no memory round trip) — nobody else writes the
    // pixel between frame f-1's resolve and frame f's.  Only valid when the tile pass already saw the previous frame's tag.
    float last[3];
#endif
};
constexpr int PATH_HAS_LAST = 1 << 30;
#ifdef PT_CARRY_LAST
constexpr size_t kLaneLastBytes = 12; // per lane: the pixel value read by the tile pass (LDS slot, see the kernel)
#else
constexpr size_t kLaneLastBytes = 0;
#endif

// spp = 1, frame pipelining: a finished path whose pixel still holds an older frame used to keep its lane until the
// previous frame's resolve arrived.  It now PARKS the result (pixel, frame of the batch, radiance: 20 bytes) in its
// wavefront's LDS list and frees the lane; the list is retried by the wavefront's first lanes once per iteration.  It
// matters when a GPU owns few tiles per frame (a 1/8 share of a 1080p image has 4,050 tiles for 6,144 wavefronts, so
// consecutive frames of one tile are in flight together all the time).
struct ParkedResolve {
    int pix, fj;
    float irr[3];
};
constexpr int PARKED_MAX = 64; // upper bound; FrameArgs::parkedMax is what a launch uses
// LDS bytes of the per-launch table of running-mean weights (spp = 1 persistent kernels), 16-byte aligned
__host__ __device__ constexpr size_t frame_weight_bytes(int batchFrames) { return (size_t)((batchFrames + 63) & ~63) * 4; }

struct BlockQueue {            // one per workgroup, in static LDS
    unsigned long long pair;   // (end << 32) | cursor : absolute tile indices of the current chunk
    unsigned int lock;         // refill lock
    unsigned int done;         // global queue exhausted
};


// relaxed workgroup-scope loads/stores of LDS control words (compile to ds_read / ds_write, never cached in registers)
PT_DEV unsigned int lds_load(const unsigned int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PT_DEV unsigned long long lds_load64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PT_DEV void lds_store(unsigned int *p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Next tile for this wavefront, or -1 when the frame's tiles are all handed out.  Wave-uniform result.
PT_DEV int queue_pop_tile(BlockQueue *q)
{
    const bool leader = (threadIdx.x & 63) == 0;
    for (;;) {
        unsigned long long old = 0;
        if (leader) old = atomicAdd(&q->pair, 1ull);
        unsigned int cursor = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)old);
        unsigned int end = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(old >> 32));
        if (cursor < end) return (int)cursor;
        if (__builtin_amdgcn_readfirstlane((int)lds_load(&q->done))) return -1;
        unsigned int got = 1;
        if (leader) got = atomicCAS(&q->lock, 0u, 1u);
        if (__builtin_amdgcn_readfirstlane((int)got) == 0) { // this wavefront refills
            // re-check under the lock: another wavefront may have refilled or hit the end meanwhile (a workgroup
            // must draw exactly ONE failing ticket per launch — the host's queueBase accounting relies on it)
            unsigned long long cur = lds_load64(&q->pair);
            unsigned int isDone = lds_load(&q->done);
            if (!isDone && (unsigned int)cur >= (unsigned int)(cur >> 32)) {
                unsigned int ticket = 0;
                ColdArgs ca = cold_args();
                const int numTiles = ca->tilesX * ca->tilesY * ca->batchFrames, chunk = ca->queueChunk; // (frame, tile) pairs, frame-major
                CHAOS(2);
                if (leader) ticket = atomicAdd(ca->queue, 1u) - ca->queueBase;
                ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
                const long long first = ((ca->tagged ? 0ll : (long long)gridDim.x) + ticket) * chunk; // tagged launches have no static chunks
                const long long last = first + chunk < numTiles ? first + chunk : numTiles;
                if (first >= numTiles) {
                    if (leader) lds_store(&q->done, 1u);
                } else {
                    if (leader) atomicExch(&q->pair, ((unsigned long long)last << 32) | (unsigned long long)first);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (leader) atomicExch(&q->lock, 0u);
        } else {
            __builtin_amdgcn_s_sleep(2);
        }
    }
}

// ---- drain compaction.  When the frame's tiles are all handed out, every wavefront still holds up to 64 paths
// of very different remaining length, and would spend ~5 more iterations mostly empty.  Instead, a draining
// wavefront that is at most half full DONATES its live paths to a per-workgroup LDS pool and exits, and the
// draining wavefronts that stay pull from that pool into their idle lanes — four thin wavefronts collapse into one
// or two full ones.  A path is a self-contained record (pixel, RNG state, ray, throughput, radiance, sample /
// bounce counters), so moving it to another lane changes nothing in its arithmetic: still bit-identical.
struct PathState { // 80 bytes
    int pix, pxy, counters; // counters = sample | bounce << 12 | needRay << 24
    uint32_t seed;
    float ro[3], rd[3], thr[3], rad[3], irr[3];
    int pad;
};
constexpr int DONATE_MAX = 32;

__host__ __device__ constexpr int pool_slots(int waves) { return (waves - 1) * DONATE_MAX; } // <= (waves-1) donors x DONATE_MAX

struct DrainControl {        // static LDS, one per workgroup
    unsigned int pushed;     // pool entries [0, pushed) are published
    unsigned int taken;      // pool entries [0, taken) are consumed
    unsigned int alive;      // wavefronts that have neither exited nor committed to donate-and-exit
    unsigned int pushing;    // donors between commit and publication
    unsigned int lock;       // serialises donors
    unsigned int pad[3];
};

//
// SPP1 (one sample per pixel per frame, the usual case) adds the TILE PASS: the wavefront that refills its ring runs the
// whole first bounce of the tile's 64 primary rays right there, all lanes together.  Primary rays of one tile are
// coherent, so the spheres are first culled against the tile's ray bundle (cull_spheres: typically 0-5 of them survive)
// and the sphere pass of the first bounce — 37 % of all rays cast at 2.7 bounces per path — shrinks from numSpheres to
// that handful; material / BSDF / environment code runs on coherent lanes too.  Paths that end at the first bounce are
// resolved immediately, the survivors go to the ring as PathEntry records and are picked up by idle lanes of the
// generic bounce loop.  Per path the arithmetic is unchanged (same tests in the same order, same RNG draws).
template <int NWAVES, int MIN_WAVES_PER_SIMD, bool TIMELINE, bool SPP1, bool MATLDS, bool GRID = false>
__global__ __launch_bounds__(NWAVES * 64, MIN_WAVES_PER_SIMD) void pt_integrate_persistent_kernel(const FrameArgs a)
{
    __shared__ __attribute__((aligned(16))) BlockQueue queue; // 16 B: keeps the dynamic-LDS base 16-byte aligned
    __shared__ __attribute__((aligned(16))) DrainControl drain; // 32 B
    const int numTilesFrame = a.tilesX * a.tilesY;
    const int numTiles = numTilesFrame * a.batchFrames;         // (frame, tile) pairs, frame-major
    // 1 / (frame + j + 1): running-mean weight of the batch's frame j.  In DYNAMIC LDS between the scene and the rings, sized by the
    // launch (64 entries; 256 only for the long batches of small shares): as a static 1 KB table it cost the 256-sphere scene its
    // sixth workgroup per CU (27.9 instead of 27.2 KB)
    // (the table's address is re-derived from the kernarg segment where it is read — see cold_args — instead of living in a register
    // across the bounce loop)
    auto frame_weights = [&]() -> float * {
        ColdArgs ca = cold_args();
        return (float *)((char *)g_lds + scene_lds_bytes(ca->numSpheres, ca->numCuboids, ca->envFormat, ca->materialsInLds != 0, ca->gridLdsBytes));
    };
    if (SPP1) {
        float *fw = frame_weights();
        for (int j = (int)threadIdx.x; j < a.batchFrames; j += (int)blockDim.x) fw[j] = f_div_ieee(1.0f, (float)(a.frame + j + 1));
    }
    if (threadIdx.x == 0) {
        // One frame per launch: the workgroup's first chunk is static (chunk index = workgroup index).  A pipelined batch
        // hands out EVERY chunk through the global counter instead: frames depend on each other per pixel, and a workgroup
        // that is not resident yet (more workgroups launched than fit, or another process on the GPU) must not own early
        // work that resident workgroups are waiting for — tickets are only ever held by workgroups that are running.
        long long first = (long long)blockIdx.x * a.queueChunk;
        long long last = first + a.queueChunk < numTiles ? first + a.queueChunk : numTiles;
        if (first >= numTiles || a.tagged) { first = 0; last = 0; }
        queue.pair = ((unsigned long long)last << 32) | (unsigned long long)first;
        queue.lock = 0u;
        queue.done = 0u;
        drain.pushed = 0u;
        drain.taken = 0u;
        drain.alive = (unsigned int)NWAVES;
        drain.pushing = 0u;
        drain.lock = 0u;
        if (a.startedFlags) // "this workgroup is resident" (launch chaining): a system-scope store, the host polls the word
            __hip_atomic_store(a.startedFlags + blockIdx.x, a.launchSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    CHAOS(1);
    SceneLds sc = stage_scene(a); // ends with __syncthreads()
    EnvRef env{nullptr, (LdsFloats)sc.lut, 0, 0}; // descriptor is cold-loaded at the miss-shading site (bounce_step)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // the ring lives behind the staged scene in dynamic LDS
    constexpr int ENTRY_BYTES = SPP1 ? (int)sizeof(PathEntry) : (int)sizeof(RingEntry);
    char *ringBase = (char *)g_lds + scene_lds_bytes(a.numSpheres, a.numCuboids, a.envFormat, a.materialsInLds != 0, a.gridLdsBytes) +
                     (SPP1 ? frame_weight_bytes(a.batchFrames) : 0);
    RingEntry *ring = (RingEntry *)(ringBase + wave * 64 * ENTRY_BYTES);  // !SPP1: primary rays
    PathEntry *pring = (PathEntry *)(ringBase + wave * 64 * ENTRY_BYTES); //  SPP1: paths after their first bounce
#ifdef PT_CARRY_LAST
    // per-lane slots for the pixel value the tile pass read (PathEntry::last travels here when a lane pops the path): three planes of
    // 64 floats per wavefront, so that keeping it costs no registers across the bounce loop
    constexpr int LANE_LAST_BYTES = SPP1 ? NWAVES * 3 * 64 * 4 : 0;
    float *laneLast = (float *)(ringBase + NWAVES * 64 * ENTRY_BYTES) + wave * 3 * 64 + lane;
#else
    constexpr int LANE_LAST_BYTES = 0;
#endif
    PathState *pool = (PathState *)(ringBase + NWAVES * 64 * ENTRY_BYTES + LANE_LAST_BYTES);
    const bool compaction = a.drainCompaction != 0;
    // parked resolves of this wavefront (pipelined spp = 1 launches only; behind the rings — such launches have no drain pool)
    ParkedResolve *parkedList = (ParkedResolve *)(ringBase + NWAVES * 64 * ENTRY_BYTES + LANE_LAST_BYTES) + wave * a.parkedMax;
    const bool parking = SPP1 && a.tagged && !compaction && a.parkedMax > 0;
    int nparked = 0, parkSpins = 0; // wave-uniform
    const int donateMax = a.drainCompaction < DONATE_MAX ? a.drainCompaction : DONATE_MAX; // a wavefront this thin donates
    const bool leader = lane == 0;

    int avail = 0;           // wave-uniform: ring entries [0, avail) are unconsumed
    bool exhausted = false;
    bool lastAlive = false;  // this wavefront found itself the last one of its workgroup: it can neither donate nor leave early
#ifdef PT_PROFILE
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_util[3] = {0, 0, 0};
    unsigned long long prof_t = __builtin_readcyclecounter();
#endif
    unsigned long long tStart = 0, tExhausted = 0, nIter = 0;
    if (TIMELINE) tStart = wall_clock64();

    int pix = -1, px = 0, py = 0, sample = 0, bounce = 0;
    bool needRay = false;
    uint32_t seed = 0;
    v3 ro = V(0, 0, 0), rd = V(0, 0, 1), throughput = V(1, 1, 1), rad = V(0, 0, 0), irr = V(0, 0, 0);
    // SPP1 / frame pipelining: index of the path's frame inside the batch, "ended, waiting for its pixel's previous
    // frame" flag, and the number of failed resolve attempts
    int fj = 0;
    bool pending = false;
    // One register, two lives.  While the lane's path is being traced (!pending): GRID kernels keep the parameter from which the grid walk
    // of the current bounce continues (>= 0 while it is unfinished, else -1: pt_device.hpp, WALK SLICES).  Once the path has ended and
    // waits for its pixel (pending): the bits count the failed resolve attempts.
    float walkFrom = -1.0f, walkFresh = -1.0f;
    auto retries = [&]() -> int { return __float_as_int(walkFrom); };
    // (PT_CARRY_LAST: bit 14 of fj = the lane's slot of laneLast holds the pixel's accumulation value as the tile pass read it)

    // compute.glsl:125-129 for one finished path of frame `rfj` of the batch.  False = the pixel still holds an older
    // frame (only possible inside a batch): try again in the next iteration.
    // compute.glsl:125-129 for frame `rfj` of the batch: irradiance / SPP folded into the running mean, alpha = 1 (or the
    // frame tag inside a batch).  The uniform inputs are re-read from the kernarg segment here (see cold_args).
    auto fold = [&](float4 last, v3 rirr, int rfj) -> float4 {
        ColdArgs ca = cold_args();
        float w;
        if constexpr (SPP1) { // irradiance / 1 is exact (x * 1.0f == x bit for bit): skipped; weight from the per-batch table
            w = frame_weights()[rfj];
        } else {
            rirr = v_scale(rirr, f_div_ieee(1.0f, (float)ca->spp));
            w = f_div_ieee(1.0f, (float)(ca->frame + rfj + 1));
        }
        const float alpha = (rfj == ca->batchFrames - 1 && !ca->keepTags) ? 1.0f : frame_tag(ca->frame + rfj);
        return make_float4(f_mix(last.x, rirr.x, w), f_mix(last.y, rirr.y, w), f_mix(last.z, rirr.z, w), alpha);
    };
#ifdef PT_CARRY_LAST
    // compute.glsl:126-129 with the pixel's value already in hand (read by the tile pass, which also checked the tag): no load
    auto commit_resolve = [&](int rpix, int rfj, v3 rirr, v3 rlast) -> void {
        const float4 last = make_float4(rlast.x, rlast.y, rlast.z, 0.0f);
        const float4 next = fold(last, rirr, rfj);
        AUDIT_RESOLVE(a, (size_t)rpix, a.frame + rfj, last, next, 7);
        if (!a.tagged) a.accum[rpix] = next;
        else store_pixel_sc1(a.accum + rpix, next);
        if (rfj == cold_args()->batchFrames - 1)
            if (float4 *snap = cold_args()->snapshot) snap[rpix] = make_float4(next.x, next.y, next.z, 1.0f);
    };
#endif
    // False = the pixel still holds an older frame (only possible inside a batch): try again in the next iteration.
    auto try_resolve = [&](int rpix, int rfj, v3 rirr, bool force) -> bool {
        float4 *ptr = a.accum + rpix;
        if (!a.tagged) {
            float4 last = *ptr;
            const float4 next = fold(last, rirr, 0);
            AUDIT_RESOLVE(a, (size_t)rpix, a.frame, last, next, 3);
            *ptr = next;
            if (float4 *snap = cold_args()->snapshot) snap[rpix] = next; // (one frame per launch: it is the last)
            return true;
        }
        CHAOS(10);
        float4 last = load_pixel_sc1(ptr);
        const float expected = rfj > 0 ? frame_tag(a.frame + rfj - 1) : a.chainTag; // (0 = the launch's first frame has no predecessor in flight)
        if (expected != 0.0f && !force && last.w != expected) return false;
        if (AUDIT_SABOTAGED(a, rpix, rfj)) last.x += 1.0f; // (audit build + PT_AUDIT_SABOTAGE only: a simulated stale / torn read)
        CHAOS(11);
        const float4 next = fold(last, rirr, rfj);
        AUDIT_RESOLVE(a, (size_t)rpix, a.frame + rfj, last, next, 4);
        store_pixel_sc1(ptr, next);
        if (rfj == cold_args()->batchFrames - 1) // the launch's last frame: the present snapshot (plain store, read after the launch)
            if (float4 *snap = cold_args()->snapshot) snap[rpix] = make_float4(next.x, next.y, next.z, 1.0f);
        CHAOS(12);
        return true;
    };

    // park the results of the lanes with `want` (as many as fit); returns true for the lanes that were parked
    auto park_resolves = [&](bool want, int rpix, int rfj, v3 rirr) -> bool {
        const unsigned long long wm = __ballot(want);
        if (wm == 0ull) return false;
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(wm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wm, 0u));
        const int room = a.parkedMax - nparked;
        const bool fits = want && rank < room;
        if (fits) {
            ParkedResolve e;
            e.pix = rpix; e.fj = rfj; e.irr[0] = rirr.x; e.irr[1] = rirr.y; e.irr[2] = rirr.z;
            parkedList[nparked + rank] = e;
        }
        const int n = __builtin_popcountll(wm);
        nparked += n < room ? n : room;
        __builtin_amdgcn_wave_barrier();
        return fits;
    };
    // retry the parked resolves: lane l takes entry l; the ones that still have to wait are compacted to the front
    auto service_parked = [&](bool force) -> void {
        if (nparked == 0) return;
        CHAOS(4);
        const bool mine = lane < nparked;
        ParkedResolve e = {0, 0, {0.0f, 0.0f, 0.0f}};
        bool keep = false;
        if (mine) {
            e = parkedList[lane];
            keep = !try_resolve(e.pix, e.fj, V(e.irr[0], e.irr[1], e.irr[2]), force);
        }
        const unsigned long long km = __ballot(keep);
        __builtin_amdgcn_wave_barrier(); // every entry has been read before the survivors are written back
        if (keep) parkedList[__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u))] = e;
        const int left = __builtin_popcountll(km);
        parkSpins = left == nparked ? parkSpins + 1 : 0;
        nparked = left;
        if (force && lane == 0) atomicOr(cold_args()->errorWord, 1u);
        __builtin_amdgcn_wave_barrier();
    };

    for (;;) {
        // ---- feed idle lanes: pop from the ring; if the ring runs dry while lanes are still idle, refill it with the next
        // tile and pop again in the SAME iteration (a lane never idles through a bounce iteration because the ring happened
        // to hold fewer rays than there were idle lanes)
        bool idle = pix < 0;
        unsigned long long m = __ballot(idle);
        for (int pass = 0; (SPP1 ? pass < 16 : pass < 2) && m != 0ull; pass++) {
            if (avail == 0) {
                if (exhausted) break;
                    // ---- refill the ring: one tile, every lane generates one primary ray
                    int tile = queue_pop_tile(&queue);
                    if (tile < 0) {
                        exhausted = true;
                        if (TIMELINE) tExhausted = wall_clock64();
                    } else if constexpr (SPP1) {
                        // ---- tile pass: primary rays + the whole first bounce, all 64 lanes together
#ifdef PT_PROFILE
                        const unsigned long long prof_tile0 = __builtin_readcyclecounter();
#endif
                        ColdArgs ca = cold_args();
                        ColdFloats cam = (ColdFloats)ca;
                        const int width = ca->width, tilesX = ca->tilesX;
                        const float invW = f_div_ieee(1.0f, (float)width), invH = f_div_ieee(1.0f, (float)ca->height);
                        const int tfj = tile / numTilesFrame; // frame of the batch this (frame, tile) ticket belongs to
                        tile -= tfj * numTilesFrame;
                        int tx = (int)(tile % tilesX), ty = (int)(tile / tilesX);
                        int x = tx * 8 + (lane & 7), ly = ty * 8 + (lane >> 3);
                        const bool valid = x < width && ly < ca->rows;
                        v3 to = V(0.0f, 0.0f, 0.0f), td = V(0.0f, 0.0f, 1.0f), tthr = V(1.0f, 1.0f, 1.0f), trad = V(0.0f, 0.0f, 0.0f);
                        uint32_t tseed = 0;
                        int tpix = 0;
                        if (valid) {
                            int gy = global_row_v(ca->bandRows, ca->bandWorld, ca->bandRank, ca->localRow0, ca->y0, ly);
                            tseed = pixel_seed(x, gy, ca->frame + tfj);
                            primary_ray_cam(cam, invW, invH, x, gy, tseed, to, td);
                            tpix = ly * width + x;
                        }
                        unsigned long long masks[4];
                        cull_spheres(sc, a.numSpheres, valid, to, td, masks);
                        bool tcont = false, tkeep = false; // tkeep: the path goes to the ring (it continues, or its resolve must wait)
#ifdef PT_CARRY_LAST
                        bool plastOk = false;
                        float4 plast = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#endif
                        if (valid) {
                            if (0 < a.rayDepth)
                                tcont = bounce_step_t<true, MATLDS>(sc, a.numSpheres, a.numCuboids, env, to, td, tthr, trad, tseed, masks, walkFresh PROF_DUMMY);
                            if (1 >= a.rayDepth) tcont = false;
                            tkeep = tcont;
#ifdef PT_CARRY_LAST
                            // imageLoad (compute.glsl:126) for the whole tile, all lanes together: 8 rows x 128 B = full lines, one memory
                            // round trip per TILE instead of one per bounce iteration.  (After the first bounce: held across it, the four
                            // registers spill.)  A plain load: a stale cached copy can only show an OLDER tag, and then the pixel takes the
                            // coherent load of try_resolve when its path ends.
                            plast = a.accum[tpix];
                            {
                                const float expected = tfj > 0 ? frame_tag(a.frame + tfj - 1) : a.chainTag;
                                plastOk = !a.tagged || expected == 0.0f || plast.w == expected;
                            }
                            if (!tcont && plastOk) { // ended at its first bounce, previous frame already there: fold and store, no second load
                                commit_resolve(tpix, tfj, v_add(V(0.0f, 0.0f, 0.0f), trad), V(plast.x, plast.y, plast.z));
                            } else
#endif
                            if (!tcont) { // the path ended at its first bounce: compute.glsl:125-129 right away
                                v3 tirr = v_add(V(0.0f, 0.0f, 0.0f), trad);
                                tkeep = !try_resolve(tpix, tfj, tirr, false);
                            }
                        }
                        if (parking) { // resolves that have to wait for the previous frame: parked (else through the ring)
                            const bool twait = valid && !tcont && tkeep;
                            if (park_resolves(twait, tpix, tfj, v_add(V(0.0f, 0.0f, 0.0f), trad))) tkeep = false;
                        }
                        const unsigned long long cm = __ballot(tkeep);
                        if (tkeep) {
                            int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
                            PathEntry e;
                            // a path whose resolve has to wait re-enters the bounce loop "at full depth": it is resolved there
                            e.pix = tpix; e.bounce = (tcont ? 1 : a.rayDepth) | (tfj << 16); e.seed = tseed;
                            e.ro[0] = to.x; e.ro[1] = to.y; e.ro[2] = to.z;
                            e.rd[0] = td.x; e.rd[1] = td.y; e.rd[2] = td.z;
                            e.thr[0] = tthr.x; e.thr[1] = tthr.y; e.thr[2] = tthr.z;
                            e.rad[0] = trad.x; e.rad[1] = trad.y; e.rad[2] = trad.z;
#ifdef PT_CARRY_LAST
                            if (plastOk) e.bounce |= PATH_HAS_LAST;
                            e.last[0] = plast.x; e.last[1] = plast.y; e.last[2] = plast.z;
#endif
                            pring[slot] = e;
                        }
                        __builtin_amdgcn_wave_barrier(); // ring entries are read by other lanes of this wave below
                        avail = __builtin_popcountll(cm);
#ifdef PT_PROFILE
                        { // slot 6 = the tile pass (taken out of the feed slot)
                            const unsigned long long d_ = __builtin_readcyclecounter() - prof_tile0;
                            prof[6] += d_;
                            prof_t += d_;
                        }
#endif
                    } else {
                        // camera block: FrameArgs is the kernel's first argument, so it starts the kernarg segment
                        ColdArgs ca = cold_args(); // opaque: load the camera here, do not keep it live across the loop
                        ColdFloats cam = (ColdFloats)ca;
                        const int width = ca->width, tilesX = ca->tilesX;
                        const float invW = f_div_ieee(1.0f, (float)width), invH = f_div_ieee(1.0f, (float)ca->height);
                        const int tfj = tile / numTilesFrame; // frame of the batch this (frame, tile) ticket belongs to
                        tile -= tfj * numTilesFrame;
                        int tx = (int)(tile % tilesX), ty = (int)(tile / tilesX);
                        int x = tx * 8 + (lane & 7), ly = ty * 8 + (lane >> 3);
                        RingEntry e;
                        e.pix = -1;
                        e.pxy = 0; e.seed = 0; e.ox = e.oy = e.oz = e.dx = e.dy = e.dz = 0.0f; e.pad = tfj;
                        if (x < width && ly < ca->rows) {
                            int gy = global_row_v(ca->bandRows, ca->bandWorld, ca->bandRank, ca->localRow0, ca->y0, ly);
                            uint32_t sd = pixel_seed(x, gy, ca->frame + tfj);
                            v3 o, d;
                            primary_ray_cam(cam, invW, invH, x, gy, sd, o, d);
                            e.pix = ly * width + x;
                            e.pxy = x | (gy << 16);
                            e.seed = sd;
                            e.ox = o.x; e.oy = o.y; e.oz = o.z; e.dx = d.x; e.dy = d.y; e.dz = d.z;
                        }
                        ring[lane] = e;
                        __builtin_amdgcn_wave_barrier(); // ring entries are read by other lanes of this wave below
                        avail = 64;
                    }

                if (exhausted) break;
                if (SPP1 && avail == 0) continue; // every path of that tile ended at its first bounce: next tile
            }
            CHAOS(3);
            if constexpr (SPP1) {
                // ---- idle lanes pop paths (top down)
                int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (idle && rank < avail) {
                    PathEntry e = pring[avail - 1 - rank];
                    pix = e.pix;
                    bounce = e.bounce & 0xffff;
                    fj = (e.bounce >> 16) & 0x3fff;
#ifdef PT_CARRY_LAST
                    if (e.bounce & PATH_HAS_LAST) fj |= 0x4000;
                    laneLast[0] = e.last[0]; laneLast[64] = e.last[1]; laneLast[128] = e.last[2];
#endif
                    pending = false;
                    walkFrom = -1.0f;
                    seed = e.seed;
                    ro = V(e.ro[0], e.ro[1], e.ro[2]);
                    rd = V(e.rd[0], e.rd[1], e.rd[2]);
                    throughput = V(e.thr[0], e.thr[1], e.thr[2]);
                    rad = V(e.rad[0], e.rad[1], e.rad[2]);
                }
                int n = __builtin_popcountll(m);
                avail = n < avail ? avail - n : 0;
            } else {
                // ---- idle lanes pop ring entries (top down); a popped out-of-image entry leaves the lane idle
                int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (idle && rank < avail) {
                    RingEntry e = ring[avail - 1 - rank];
                    if (e.pix >= 0) {
                        pix = e.pix;
                        px = e.pxy & 0xffff;
                        py = e.pxy >> 16;
                        seed = e.seed;
                        ro = V(e.ox, e.oy, e.oz);
                        rd = V(e.dx, e.dy, e.dz);
                        throughput = V(1.0f, 1.0f, 1.0f);
                        rad = V(0.0f, 0.0f, 0.0f);
                        irr = V(0.0f, 0.0f, 0.0f);
                        sample = 0;
                        fj = e.pad;
                        pending = false;
                        walkFrom = -1.0f;
                        bounce = 0;
                        needRay = false;
                    }
                }
                int n = __builtin_popcountll(m);
                avail = n < avail ? avail - n : 0;
            }
            idle = pix < 0;
            m = __ballot(idle);
        }
        if (m != 0ull) {
            if (avail > 0) {
                // (ring entries left: the remaining idle lanes popped out-of-image entries of a ragged tile)
            } else if (exhausted && compaction) {
                // ---- drain: idle lanes adopt donated paths from the workgroup's pool
                unsigned int pushed = lds_load(&drain.pushed), taken = lds_load(&drain.taken);
                pushed = (unsigned int)__builtin_amdgcn_readfirstlane((int)pushed);
                taken = (unsigned int)__builtin_amdgcn_readfirstlane((int)taken);
                if (pushed > taken) {
                    unsigned int want = (unsigned int)__builtin_popcountll(m), have = pushed - taken;
                    unsigned int k = want < have ? want : have;
                    unsigned int got = ~0u;
                    if (leader) got = atomicCAS(&drain.taken, taken, taken + k);
                    got = (unsigned int)__builtin_amdgcn_readfirstlane((int)got);
                    if (got == taken) { // claimed entries [taken, taken + k)
                        int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (idle && (unsigned int)rank < k) {
                            PathState st = pool[taken + rank];
                            pix = st.pix;
                            bounce = (st.counters >> 12) & 0xfff;
                            pending = (st.counters >> 25) & 1;
                            fj = (st.counters >> 26) & 0x3f;
                            walkFrom = pending ? 0.0f : -1.0f;
#ifdef PT_CARRY_LAST
                            if constexpr (SPP1) {
                                if (st.counters & 1) fj |= 0x4000;
                                laneLast[0] = st.irr[0]; laneLast[64] = st.irr[1]; laneLast[128] = st.irr[2];
                            }
#endif
                            seed = st.seed;
                            ro = V(st.ro[0], st.ro[1], st.ro[2]);
                            rd = V(st.rd[0], st.rd[1], st.rd[2]);
                            throughput = V(st.thr[0], st.thr[1], st.thr[2]);
                            rad = V(st.rad[0], st.rad[1], st.rad[2]);
                            if constexpr (!SPP1) {
                                px = st.pxy & 0xffff;
                                py = st.pxy >> 16;
                                sample = st.counters & 0xfff;
                                needRay = (st.counters >> 24) & 1;
                                irr = V(st.irr[0], st.irr[1], st.irr[2]);
                            }
                        }
                    }
                }
            }
        }
        bool active = pix >= 0;
        const unsigned long long am = __ballot(active);
        if (am == 0ull) {
            if (parking && nparked > 0) { // nothing to trace: look after the parked resolves (they must be gone before leaving)
                service_parked(parkSpins > FRAME_RETRY_LIMIT);
                if (nparked > 0 && exhausted && avail == 0) __builtin_amdgcn_s_sleep(8);
            }
            if (!(exhausted && avail == 0 && nparked == 0)) continue;
            if (!compaction) break;
            // ---- leaving: the last wavefront of the workgroup must outlive every donor and empty the pool
            unsigned int old = 0;
            if (leader) old = atomicSub(&drain.alive, 1u);
            old = (unsigned int)__builtin_amdgcn_readfirstlane((int)old);
            if (old > 1u) break;
            lastAlive = true;
            unsigned int pushing = lds_load(&drain.pushing), pushed = lds_load(&drain.pushed), taken = lds_load(&drain.taken);
            bool pending = __builtin_amdgcn_readfirstlane((int)(pushing != 0u || pushed != taken)) != 0;
            if (!pending) break;
            if (leader) atomicAdd(&drain.alive, 1u);
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        if (compaction && exhausted && avail == 0 && !lastAlive && __builtin_popcountll(am) <= donateMax) {
            // ---- donate: commit (pushing++, alive--), publish the live paths under the donor lock, exit
            unsigned int old = 0, base = 0;
            if (leader) {
                atomicAdd(&drain.pushing, 1u);
                old = atomicSub(&drain.alive, 1u);
            }
            old = (unsigned int)__builtin_amdgcn_readfirstlane((int)old);
            bool committed = old > 1u;
            if (committed) {
                if (leader) {
                    while (atomicCAS(&drain.lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
                    base = lds_load(&drain.pushed);
                }
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                unsigned int n = (unsigned int)__builtin_popcountll(am);
                if (base + n <= (unsigned int)pool_slots(NWAVES)) {
                    int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u));
                    if (active) {
                        PathState st;
                        st.pix = pix;
                        st.pxy = SPP1 ? 0 : (px | (py << 16));
                        st.counters = (SPP1 ? 0 : (sample | ((needRay ? 1 : 0) << 24))) | (bounce << 12) | ((pending ? 1 : 0) << 25) | ((fj & 0x3f) << 26);
#ifdef PT_CARRY_LAST
                        if (SPP1 && (fj & 0x4000)) st.counters |= 1; // (the sample field is unused with one sample per pixel)
#endif
                        st.seed = seed;
                        st.ro[0] = ro.x; st.ro[1] = ro.y; st.ro[2] = ro.z;
                        st.rd[0] = rd.x; st.rd[1] = rd.y; st.rd[2] = rd.z;
                        st.thr[0] = throughput.x; st.thr[1] = throughput.y; st.thr[2] = throughput.z;
                        st.rad[0] = rad.x; st.rad[1] = rad.y; st.rad[2] = rad.z;
                        st.irr[0] = SPP1 ? 0.0f : irr.x; st.irr[1] = SPP1 ? 0.0f : irr.y; st.irr[2] = SPP1 ? 0.0f : irr.z;
#ifdef PT_CARRY_LAST
                        if constexpr (SPP1) { st.irr[0] = laneLast[0]; st.irr[1] = laneLast[64]; st.irr[2] = laneLast[128]; } // (irr is unused with one sample per pixel)
#endif
                        st.pad = 0;
                        pool[base + rank] = st;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (leader) {
                        lds_store(&drain.pushed, base + n);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        atomicExch(&drain.lock, 0u);
                        atomicSub(&drain.pushing, 1u);
                    }
                    break; // this wavefront is done; its paths live on in the pool
                }
                // pool full (cannot happen with <= 3 donors x DONATE_MAX, kept for safety): withdraw the commit
                if (leader) atomicExch(&drain.lock, 0u);
            } else {
                lastAlive = true;
            }
            if (leader) {
                atomicAdd(&drain.alive, 1u);
                atomicSub(&drain.pushing, 1u);
            }
        }
        if (TIMELINE) nIter++;
        PROF_MARK(0) // feed: ring refill / pop / adopt / donate
#ifdef PT_PROFILE
        // lane utilisation of the generic bounce iteration: iterations, active lanes, lanes waiting for their pixel
        prof_util[0] += 1ull;
        prof_util[1] += (unsigned long long)__builtin_popcountll(am);
        prof_util[2] += (unsigned long long)__builtin_popcountll(__ballot(active && pending));
#endif
        if constexpr (SPP1) {
            if (active) {
                if (!pending) {
                    bool cont = false;
                    if (bounce < a.rayDepth) cont = bounce_step_t<false, MATLDS, GRID, (MATLDS && !GRID)>(sc, a.numSpheres, a.numCuboids, env, ro, rd, throughput, rad, seed, nullptr, walkFrom PROF_PASS);
                    if (!(GRID && walkFrom >= 0.0f)) { // (else: the grid walk of this bounce continues in the next iteration, pt_device.hpp WALK SLICES)
                        bounce++;
                        pending = !cont || bounce >= a.rayDepth;
                        if (pending) walkFrom = 0.0f; // (= no failed resolve attempt yet)
                    }
                }
#ifdef PT_PROFILE
                prof_t = __builtin_readcyclecounter();
#endif
#ifdef PT_CARRY_LAST
                if (pending && (fj & 0x4000)) { // the tile pass read the pixel (and saw the previous frame's tag): no load, cannot fail
                    commit_resolve(pix, fj & 0x3fff, v_add(V(0.0f, 0.0f, 0.0f), rad), V(laneLast[0], laneLast[64], laneLast[128]));
                    pix = -1;
                    pending = false;
                }
#endif
                if (pending) {
                    v3 firr = v_add(V(0.0f, 0.0f, 0.0f), rad);
                    const bool force = retries() > FRAME_RETRY_LIMIT;
                    if (try_resolve(pix, fj, firr, force)) {
                        if (force) atomicOr(cold_args()->errorWord, 1u); // host-visible error word
                        pix = -1;
                        pending = false;
                    } else {
                        walkFrom = __int_as_float(retries() + 1);
                    }
                }
            }
            if (parking) {
                // a resolve that has to wait gives its lane back: the result is parked, the wavefront's first lanes retry it
                const bool wait = pix >= 0 && pending;
                if (park_resolves(wait, pix, fj, v_add(V(0.0f, 0.0f, 0.0f), rad))) {
                    pix = -1;
                    pending = false;
                }
                service_parked(parkSpins > FRAME_RETRY_LIMIT);
            }
            // nothing but waiting paths left in this wavefront: do not hammer the pixel
            if ((__ballot(pix >= 0 && pending) != 0ull || nparked > 0) && __ballot(pix >= 0 && !pending) == 0ull && avail == 0) __builtin_amdgcn_s_sleep(8);
        } else {
        if (active && needRay) { // only for spp > 1: the next sample continues the pixel's RNG stream (compute.glsl:110)
            primary_ray(a, px, py, seed, ro, rd);
            throughput = V(1.0f, 1.0f, 1.0f);
            rad = V(0.0f, 0.0f, 0.0f);
            bounce = 0;
            needRay = false;
        }
        if (active) {
            if (!pending) {
                bool cont = false;
                if (bounce < a.rayDepth) cont = bounce_step_t<false, MATLDS, GRID>(sc, a.numSpheres, a.numCuboids, env, ro, rd, throughput, rad, seed, nullptr, walkFrom PROF_PASS);
                const bool sliced = GRID && walkFrom >= 0.0f; // (the grid walk of this bounce continues in the next iteration)
                if (!sliced) bounce++;
                if (!sliced && (!cont || bounce >= a.rayDepth)) {
                    irr = v_add(irr, rad);
                    sample++;
                    if (sample < a.spp) needRay = true;
                    else { pending = true; walkFrom = 0.0f; } // the pixel's last sample: fold into the accumulation image (no failed attempt yet)
                }
            }
#ifdef PT_PROFILE
            prof_t = __builtin_readcyclecounter();
#endif
            if (pending) {
                const bool force = retries() > FRAME_RETRY_LIMIT;
                if (try_resolve(pix, fj, irr, force)) {
                    if (force) atomicOr(cold_args()->errorWord, 1u);
                    pix = -1;
                    pending = false;
                } else {
                    walkFrom = __int_as_float(retries() + 1);
                }
            }
        }
        if (__ballot(active && pending) != 0ull && __ballot(active && !pending) == 0ull) __builtin_amdgcn_s_sleep(8);
        } // !SPP1
        PROF_MARK(7) // resolve
    }
#ifdef PT_PROFILE
    if (a.timeline && lane == 0)
        for (int k = 0; k < 8; k++) {
            atomicAdd(a.timeline + 200000 + k, prof[k]);
            if (k < 3) atomicAdd(a.timeline + 200008 + k, prof_util[k]);
        }
#endif
    if (TIMELINE && lane == 0) {
        unsigned long long *t = a.timeline + ((size_t)blockIdx.x * NWAVES + wave) * 4;
        t[0] = tStart; t[1] = tExhausted; t[2] = wall_clock64(); t[3] = nIter;
    }
}

// ---- spp > 1: the BATCH PASS kernel.
// With several samples per pixel per frame the samples of a pixel form a chain: sample s+1 starts from the RNG state sample s
// ended with (compute.glsl:106-124, one stream per pixel per frame), so a pixel's next primary ray can only be generated
// when its previous path has ended — after a different number of bounces for every pixel.  Generating it right there
// (the persistent kernel's spp > 1 path) runs the camera code and a full, unculled first bounce on a few lanes at a time.
// Here a lane that finishes a sample instead parks the pixel's continuation (pixel, RNG state, radiance so far, sample
// counter: 28 bytes) in its wavefront's LDS queue and takes other work; when the wavefront next runs out of ring
// entries it turns up to 64 parked continuations — or a fresh 8x8 tile for sample 0 — into a BATCH PASS: 64 primary rays
// and their whole first bounce with all lanes together, exactly like the spp = 1 tile pass.  The sphere culling needs no
// tile structure: cull_spheres() bounds whatever 64 rays the wavefront holds (a wavefront's tiles are neighbours, and all
// primary rays leave the lens), so every sample's first bounce — 1 / 2.7 of all rays cast — visits a handful of spheres
// instead of all of them.  A continuation that finds the queue full falls back to the divergent in-lane primary ray.
// Per pixel nothing changes: same samples in the same order on one RNG stream, irradiance summed in sample order -> the
// image is bit-identical to every other variant.  Frames are pipelined exactly as in the spp = 1 kernel (alpha tags).
struct PathEntryM { // 72 bytes: a path after its first bounce, plus what its pixel needs for the samples that follow
    int pix;        // x | local row << 16 of the pixel in this launch's accumulation rows (both < 32768: no division to unpack)
    int counters;   // bounces done | sample << 12 | frame of the batch << 24 ; bit 31: no ray yet (generate it in the lane)
    uint32_t seed;
    float ro[3], rd[3], thr[3], rad[3], irr[3];
};
struct ContEntry {  // 24 bytes: a pixel between two of its samples
    int pix;
    uint32_t seed;
    int sfj;        // sample | frame of the batch << 16
    float irr[3];
};

template <bool MATLDS, bool GRID = false>
__global__ __launch_bounds__(256, 5) void pt_integrate_multisample_kernel(const FrameArgs a)
{
    __shared__ __attribute__((aligned(16))) BlockQueue queue;
    constexpr int NWAVES = 4;
    const int numTilesFrame = a.tilesX * a.tilesY;
    const int numTiles = numTilesFrame * a.batchFrames; // (frame, tile) pairs, frame-major
    if (threadIdx.x == 0) {
        long long first = (long long)blockIdx.x * a.queueChunk;
        long long last = first + a.queueChunk < numTiles ? first + a.queueChunk : numTiles;
        if (first >= numTiles || a.tagged) { first = 0; last = 0; } // tagged launches draw every chunk from the global counter
        queue.pair = ((unsigned long long)last << 32) | (unsigned long long)first;
        queue.lock = 0u;
        queue.done = 0u;
        if (a.startedFlags) // "this workgroup is resident" (launch chaining): a system-scope store, the host polls the word
            __hip_atomic_store(a.startedFlags + blockIdx.x, a.launchSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    CHAOS(1);
    SceneLds sc = stage_scene(a); // ends with __syncthreads()
    EnvRef env{nullptr, (LdsFloats)sc.lut, 0, 0};
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    char *ringBase = (char *)g_lds + scene_lds_bytes(a.numSpheres, a.numCuboids, a.envFormat, a.materialsInLds != 0, a.gridLdsBytes);
    PathEntryM *ring = (PathEntryM *)ringBase + wave * 64;
    const int CONT_BATCH_MIN = a.contBatchMin; // parked continuations that make a batch pass worth its ~950 instructions
    const int parkCapacity = a.contCapacity; // per wavefront (whatever LDS is left next to scene and rings, see the launch)
    ContEntry *cq = (ContEntry *)(ringBase + NWAVES * 64 * (int)sizeof(PathEntryM)) + wave * parkCapacity;
    // image coordinates of accumulation pixel `p` of this launch: x | global row << 16
    auto pixel_xy = [&](int p) -> int { // p = x | local row << 16 (no division anywhere)
        ColdArgs ca = cold_args();
        const int ly = p >> 16, x = p & 0xffff;
        return x | (global_row_v(ca->bandRows, ca->bandWorld, ca->bandRank, ca->localRow0, ca->y0, ly) << 16);
    };

    int avail = 0, parked = 0, qhead = 0; // wave-uniform: ring entries [0, avail); `parked` continuations from slot qhead on (FIFO, circular)
    int stalled = 0;                      // wave-uniform: consecutive iterations in which no lane traced anything
    auto qslot = [&](int i) -> int { // slot of the i-th parked continuation
        int sl = qhead + i;
        return sl >= parkCapacity ? sl - parkCapacity : sl;
    };
    bool exhausted = false;
    int pix = -1, sample = 0, bounce = 0, fj = 0;
    bool needRay = false, pending = false;
    float walkFrom = -1.0f, walkFresh = -1.0f; // (WALK SLICES, as in the persistent kernel)
    uint32_t seed = 0;
    v3 ro = V(0, 0, 0), rd = V(0, 0, 1), throughput = V(1, 1, 1), rad = V(0, 0, 0), irr = V(0, 0, 0);

    auto fold = [&](float4 last, v3 rirr, int rfj) -> float4 { // compute.glsl:125-129
        ColdArgs ca = cold_args();
        rirr = v_scale(rirr, f_div_ieee(1.0f, (float)ca->spp));
        const float w = f_div_ieee(1.0f, (float)(ca->frame + rfj + 1));
        const float alpha = (rfj == ca->batchFrames - 1 && !ca->keepTags) ? 1.0f : frame_tag(ca->frame + rfj);
        return make_float4(f_mix(last.x, rirr.x, w), f_mix(last.y, rirr.y, w), f_mix(last.z, rirr.z, w), alpha);
    };
    auto try_resolve = [&](int rpix, int rfj, v3 rirr, bool force) -> bool {
        const size_t pidx = (size_t)((rpix >> 16) * cold_args()->width + (rpix & 0xffff));
        float4 *ptr = a.accum + pidx;
        if (!a.tagged) {
            const float4 last = *ptr, next = fold(last, rirr, 0);
            AUDIT_RESOLVE(a, pidx, a.frame, last, next, 5);
            *ptr = next;
            if (float4 *snap = cold_args()->snapshot) snap[pidx] = next;
            return true;
        }
        CHAOS(20);
        float4 last = load_pixel_sc1(ptr);
        const float expected = rfj > 0 ? frame_tag(a.frame + rfj - 1) : a.chainTag;
        if (expected != 0.0f && !force && last.w != expected) return false;
        if (AUDIT_SABOTAGED(a, rpix, rfj)) last.x += 1.0f; // (audit build + PT_AUDIT_SABOTAGE only: a simulated stale / torn read)
        CHAOS(21);
        const float4 next = fold(last, rirr, rfj);
        AUDIT_RESOLVE(a, pidx, a.frame + rfj, last, next, 6);
        store_pixel_sc1(ptr, next);
        if (rfj == cold_args()->batchFrames - 1)
            if (float4 *snap = cold_args()->snapshot) snap[pidx] = make_float4(next.x, next.y, next.z, 1.0f);
        CHAOS(22);
        return true;
    };

    // ---- rescue.  Inside a pipelined batch a pixel's last sample may have to wait for the pixel's previous frame.  If ALL
    // lanes of a wavefront wait like that, nothing pops its ring or runs a batch pass any more — and the work those lanes
    // wait for may be exactly what sits in this wavefront's ring or queue (small images: consecutive frames of one tile meet
    // in one wavefront).  So a wavefront whose lanes all wait moves the waiting results out of the lanes into free slots of
    // the continuation queue (a waiting result — pixel, frame, irradiance — is a continuation with sample == spp; batch passes
    // retry it, oldest first) and the freed lanes pop the ring as usual: no queued work depends on a lane that only waits,
    // and every pixel still runs the same samples in the same order on its own RNG stream.  Should the queue itself be full
    // of waiting results (more than 150 finished pixels of one wavefront all waiting for other frames), the per-wavefront
    // stall bound below ends the wait with the error word instead of hanging.  (Swapping waiting results with queued paths
    // was tried first: it needs the path state to be assignable at a second place, which costs 20 spilled VGPRs.)
    auto rescue = [&]() -> void {
        const int room = parkCapacity - parked;
        if (room > 0) {
            if (lane < room) { // (every lane waits, so lane l parks into the l-th free slot)
                ContEntry e;
                e.pix = pix; e.seed = seed; e.sfj = sample | (fj << 16); // sample == spp marks "last sample done, waiting"
                e.irr[0] = irr.x; e.irr[1] = irr.y; e.irr[2] = irr.z;
                cq[qslot(parked + lane)] = e;
                pix = -1;
                pending = false;
            }
            parked += room < 64 ? room : 64;
        }
        __builtin_amdgcn_wave_barrier();
    };

    for (;;) {
        bool idle = pix < 0;
        unsigned long long m = __ballot(idle);
        // ---- forced batch pass (the progress guarantee of the pipelining).  When EVERY lane holds a finished pixel that waits for its
        // previous frame, nothing pops the ring or runs a batch pass any more — and the work those lanes wait for may be parked in this
        // wavefront's own queue.  A batch pass needs no idle lane (it computes in its own registers and only needs ring slots), so
        // such a wavefront runs one over the oldest parked records anyway, appending the survivors to the ring.  Every record keeps
        // exactly one place (lane, ring slot or queue slot), a pass that finds real work advances it by a bounce or a sample and
        // leaves at least as much room in the queue as it put paths into the ring — so rescue() below can then free lanes for them.
        // Waiting records it meets are retried and rotate to the back of the FIFO.  With the tickets handed out frame-major, all
        // work of the oldest unfinished frame is therefore always executed by whichever wavefront holds it: no cycle of waits.
        bool forcePass = parked > 0 && avail < 64 && __ballot(!(pix >= 0 && pending && !needRay)) == 0ull;
        for (int pass = 0; pass < 16 && (m != 0ull || forcePass); pass++) {
            if (avail == 0 || forcePass) {
                // ---- batch pass: 64 parked continuations, or the next tile's 64 pixels (sample 0)
                const bool fromQueue = forcePass || parked >= CONT_BATCH_MIN || (exhausted && parked > 0);
                forcePass = false;
                const int base = avail; // ring entries already there (only a forced pass finds any)
                int tile = -1;
                if (!fromQueue) {
                    if (exhausted) break;
                    tile = queue_pop_tile(&queue);
                    if (tile < 0) {
                        exhausted = true;
                        continue; // (parked continuations, if any, are next)
                    }
                }
                ColdArgs ca = cold_args();
                ColdFloats cam = (ColdFloats)ca;
                const int width = ca->width;
                const float invW = f_div_ieee(1.0f, (float)width), invH = f_div_ieee(1.0f, (float)ca->height);
                bool valid = false;
                int tpix = 0, tpxy = 0, tsample = 0, tfj = 0;
                uint32_t tseed = 0;
                v3 tirr = V(0.0f, 0.0f, 0.0f);
                if (fromQueue) {
                    const int n = parked < 64 - base ? parked : 64 - base;
                    valid = lane < n;
                    if (valid) {
                        const ContEntry e = cq[qslot(lane)]; // oldest first: a parked pixel never waits behind younger ones
                        tpix = e.pix; tpxy = pixel_xy(e.pix); tseed = e.seed;
                        tsample = e.sfj & 0xffff; tfj = e.sfj >> 16;
                        tirr = V(e.irr[0], e.irr[1], e.irr[2]);
                    }
                    qhead = qslot(n);
                    parked -= n;
                    __builtin_amdgcn_wave_barrier(); // the entries are read before this pass parks new ones in their place
                } else {
                    const int tilesX = ca->tilesX;
                    tfj = tile / numTilesFrame;
                    tile -= tfj * numTilesFrame;
                    const int tx = tile % tilesX, ty = tile / tilesX;
                    const int x = tx * 8 + (lane & 7), ly = ty * 8 + (lane >> 3);
                    valid = x < width && ly < ca->rows;
                    if (valid) {
                        const int gy = global_row_v(ca->bandRows, ca->bandWorld, ca->bandRank, ca->localRow0, ca->y0, ly);
                        tpix = x | (ly << 16);
                        tpxy = x | (gy << 16);
                        tseed = pixel_seed(x, gy, ca->frame + tfj);
                    }
                }
                // (a parked record with sample == spp is a finished pixel that waited for its previous frame: retried below)
                const bool twaiting = valid && tsample >= a.spp;
                valid = valid && !twaiting;
                v3 to = V(0.0f, 0.0f, 0.0f), td = V(0.0f, 0.0f, 1.0f), tthr = V(1.0f, 1.0f, 1.0f), trad = V(0.0f, 0.0f, 0.0f);
                if (valid) primary_ray_cam(cam, invW, invH, tpxy & 0xffff, tpxy >> 16, tseed, to, td);
                unsigned long long masks[4];
                cull_spheres(sc, a.numSpheres, valid, to, td, masks);
                bool tcont = false;
                if (valid) {
                    if (0 < a.rayDepth) tcont = bounce_step_t<true, MATLDS>(sc, a.numSpheres, a.numCuboids, env, to, td, tthr, trad, tseed, masks, walkFresh);
                    if (1 >= a.rayDepth) tcont = false;
                }
                // 1. paths that continue go to the ring
                const unsigned long long cm = __ballot(tcont);
                if (tcont) {
                    const int slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
                    PathEntryM e;
                    e.pix = tpix; e.counters = 1 | (tsample << 12) | (tfj << 24); e.seed = tseed;
                    e.ro[0] = to.x; e.ro[1] = to.y; e.ro[2] = to.z;
                    e.rd[0] = td.x; e.rd[1] = td.y; e.rd[2] = td.z;
                    e.thr[0] = tthr.x; e.thr[1] = tthr.y; e.thr[2] = tthr.z;
                    e.rad[0] = trad.x; e.rad[1] = trad.y; e.rad[2] = trad.z;
                    e.irr[0] = tirr.x; e.irr[1] = tirr.y; e.irr[2] = tirr.z;
                    ring[slot] = e;
                }
                avail = base + __builtin_popcountll(cm);
                // 2. samples that ended at their first bounce: irradiance += Radiance (compute.glsl:122); more samples to go ->
                // park the continuation; the pixel's last sample -> compute.glsl:125-129
                const bool tfin = valid && !tcont;
                if (tfin) {
                    tirr = v_add(tirr, trad);
                    tsample++;
                }
                bool tmore = tfin && tsample < a.spp;
                if (twaiting) { // still waiting: back into the queue, as it was
                    const bool force = stalled > FRAME_RETRY_LIMIT;
                    if (!try_resolve(tpix, tfj, tirr, force)) tmore = true;
                    else if (force) atomicOr(cold_args()->errorWord, 1u);
                }
                const unsigned long long pm = __ballot(tmore);
                bool toRing = false; // overflow of the queue / a resolve that has to wait: through the ring, handled in the lane
                int ringCounters = 0;
                if (pm != 0ull) {
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pm, 0u));
                    const int room = parkCapacity - parked;
                    if (tmore && rank < room) {
                        ContEntry e;
                        e.pix = tpix; e.seed = tseed; e.sfj = tsample | (tfj << 16);
                        e.irr[0] = tirr.x; e.irr[1] = tirr.y; e.irr[2] = tirr.z;
                        cq[qslot(parked + rank)] = e;
                    } else if (tmore) {
                        toRing = true;
                        ringCounters = (tsample << 12) | (tfj << 24) | (int)0x80000000; // no ray yet
                    }
                    const int n = __builtin_popcountll(pm);
                    parked += n < room ? n : room;
                }
                if (tfin && tsample >= a.spp && !try_resolve(tpix, tfj, tirr, false)) {
                    toRing = true;
                    ringCounters = a.rayDepth | (tsample << 12) | (tfj << 24); // "at full depth": resolved in the bounce loop
                }
                const unsigned long long wm = __ballot(toRing);
                if (wm != 0ull) {
                    if (toRing) {
                        const int slot = avail + __builtin_amdgcn_mbcnt_hi((unsigned)(wm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wm, 0u));
                        PathEntryM e;
                        e.pix = tpix; e.counters = ringCounters; e.seed = tseed;
                        e.ro[0] = e.ro[1] = e.ro[2] = 0.0f; e.rd[0] = e.rd[1] = 0.0f; e.rd[2] = 1.0f;
                        e.thr[0] = e.thr[1] = e.thr[2] = 1.0f;
                        e.rad[0] = e.rad[1] = e.rad[2] = 0.0f;
                        e.irr[0] = tirr.x; e.irr[1] = tirr.y; e.irr[2] = tirr.z;
                        ring[slot] = e;
                    }
                    avail += __builtin_popcountll(wm);
                }
                __builtin_amdgcn_wave_barrier(); // ring / queue entries are read by other lanes of this wave below
                if (avail == 0) continue;
            }
            // ---- idle lanes pop paths (top down)
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (idle && rank < avail) {
                const PathEntryM e = ring[avail - 1 - rank];
                pix = e.pix;
                bounce = e.counters & 0xfff;
                sample = (e.counters >> 12) & 0xfff;
                fj = (e.counters >> 24) & 0x7f;
                needRay = e.counters < 0;
                pending = false;
                walkFrom = -1.0f; // (a fresh path: no unfinished grid walk)
                seed = e.seed;
                ro = V(e.ro[0], e.ro[1], e.ro[2]);
                rd = V(e.rd[0], e.rd[1], e.rd[2]);
                throughput = V(e.thr[0], e.thr[1], e.thr[2]);
                rad = V(e.rad[0], e.rad[1], e.rad[2]);
                irr = V(e.irr[0], e.irr[1], e.irr[2]);
                if (!needRay && bounce >= a.rayDepth && sample >= a.spp) pending = true; // a resolve that had to wait
            }
            const int n = __builtin_popcountll(m);
            avail = n < avail ? avail - n : 0;
            idle = pix < 0;
            m = __ballot(idle);
        }
        bool active = pix >= 0;
        if (__ballot(active) == 0ull) {
            if (exhausted && avail == 0 && parked == 0) break;
            stalled++; // (only waiting records left in the queue: they are retried by the batch passes above)
            if (parked > 0 && avail == 0) __builtin_amdgcn_s_sleep(8);
            continue;
        }
        if (__ballot(!(active && pending && !needRay)) == 0ull && (avail > 0 || parked > 0)) {
            rescue(); // every lane waits: see above
            // (round 4: the lanes rescue() has just emptied are NOT active any more.  They used to run one bounce of their dead path below —
            // harmless while a bounce left nothing behind in the lane, but a grid walk cut short (WALK SLICES) leaves walkFrom, and the
            // path the lane pops next would have resumed someone else's walk: found by tools/handover_stress --multisample)
            active = pix >= 0;
        }
        // (a wavefront that has done nothing but wait for FRAME_RETRY_LIMIT iterations in a row gives up the hand-over: waiting
        // records move between lanes, ring and queue, so the bound is kept per wavefront, not per lane)
        stalled = __ballot(pix >= 0 && !pending) == 0ull ? stalled + 1 : 0;
        if (active && needRay) { // fallback (queue was full): the next sample's primary ray, generated in the lane
            const int pxy = pixel_xy(pix);
            primary_ray(a, pxy & 0xffff, pxy >> 16, seed, ro, rd);
            throughput = V(1.0f, 1.0f, 1.0f);
            rad = V(0.0f, 0.0f, 0.0f);
            bounce = 0;
            needRay = false;
        }
        bool wantPark = false;
        if (active && !pending) {
            bool cont = false;
            if (bounce < a.rayDepth) cont = bounce_step_t<false, MATLDS, GRID>(sc, a.numSpheres, a.numCuboids, env, ro, rd, throughput, rad, seed, nullptr, walkFrom);
            const bool sliced = GRID && walkFrom >= 0.0f; // (the grid walk of this bounce continues in the next iteration: pt_device.hpp, WALK SLICES)
            if (!sliced) bounce++;
            if (!sliced && (!cont || bounce >= a.rayDepth)) {
                irr = v_add(irr, rad); // compute.glsl:122
                sample++;
                if (sample < a.spp) wantPark = true;
                else pending = true; // the pixel's last sample: fold into the accumulation image
            }
        }
        // ---- park the pixels whose sample ended; the lane is free for other work
        const unsigned long long pm = __ballot(wantPark);
        if (pm != 0ull) {
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pm, 0u));
            const int room = parkCapacity - parked;
            if (wantPark && rank < room) {
                ContEntry e;
                e.pix = pix; e.seed = seed; e.sfj = sample | (fj << 16);
                e.irr[0] = irr.x; e.irr[1] = irr.y; e.irr[2] = irr.z;
                cq[qslot(parked + rank)] = e;
                pix = -1;
            } else if (wantPark) {
                needRay = true;
            }
            const int n = __builtin_popcountll(pm);
            parked += n < room ? n : room;
            __builtin_amdgcn_wave_barrier();
        }
        if (pix >= 0 && pending) {
            const bool force = stalled > FRAME_RETRY_LIMIT; // (bounded per wavefront: waiting records move between lanes, ring and queue)
            if (try_resolve(pix, fj, irr, force)) {
                if (force) atomicOr(cold_args()->errorWord, 1u);
                pix = -1;
                pending = false;
            }
        }
        { // nothing but waiting paths left in this wavefront: do not hammer the pixel
            const bool act = pix >= 0;
            if (__ballot(act && pending) != 0ull && __ballot(act && !pending) == 0ull && parked < CONT_BATCH_MIN && avail == 0)
                __builtin_amdgcn_s_sleep(8);
        }
    }
}

// Kernel variants (pt_set_variant; every variant produces the same bits):
//   0        default = persistent queue kernel (5 workgroups per CU per stripe; 6 for a pipelined batch)
//   1        one wavefront per 8x8 tile, one pixel per lane (the reference's own mapping; simplest kernel)
//   2..6     wave-local pixel pools of 8 / 4 / 16 / 32 / 2 tiles with path regeneration
//   10 + k   persistent queue kernel with k + 1 workgroups per CU
static int pool_tiles_for_variant(int variant)
{
    switch (variant) {
    case 2: return 8;
    case 3: return 4;
    case 4: return 16;
    case 5: return 32;
    case 6: return 2;
    default: return 8;
    }
}

hipError_t launch_integrate(const FrameArgs &args, hipStream_t stream, unsigned int *ticketsConsumed, int *workgroups)
{
    FrameArgs a = args;
    a.materialsInLds = 1;
    a.gridLdsBytes = 0;
    *ticketsConsumed = 0;
    int tiles = a.tilesX * a.tilesY;
    size_t lds = scene_lds_bytes(a.numSpheres, a.numCuboids, a.envFormat, true);
    if (a.variant == 1) {
        int nwg = (tiles + 3) / 4;
        hipLaunchKernelGGL(pt_integrate_kernel, dim3(nwg), dim3(256), lds, stream, a);
    } else if (a.variant == 0 || a.variant >= 10) {
        // 10+k: k+1 workgroups (256 threads) per CU
        const int waves = 4;
        int k = a.variant == 0 ? 4 : a.variant - 10;
        if (k < 0 || k > 7) k = 4;
        int blocksPerCU = k + 1;
        if (a.spp != 1 && blocksPerCU > 5) blocksPerCU = 5; // the kernels without tile pass need 88 VGPRs: 5 wavefronts per SIMD
        int nwg = a.numCUs * blocksPerCU;
        if (a.batchFrames < 1 || a.batchFrames > MAX_BATCH_FRAMES) return hipErrorInvalidValue;
        if (a.batchFrames > 64) a.drainCompaction = 0; // (the drain pool's records keep the frame of the batch in 6 bits)
        int numChunks = (int)(((long long)tiles * a.batchFrames + a.queueChunk - 1) / a.queueChunk); // (frame, tile) pairs
        if (nwg > numChunks) nwg = numChunks;
        if (nwg < 1) nwg = 1;
        if (nwg > kStartedWords) a.startedFlags = nullptr; // (the roll call has kStartedWords words; the host then never chains on this launch)
        // Parked resolves (pipelined spp = 1 launches): what a GPU that owns few tiles per frame needs (consecutive frames of a tile in
        // flight together all the time: +20 % at a 1/8 share of 1080p) and a full image does not (+0.3 %).
#ifdef PT_CARRY_LAST
        a.parkedMax = (long long)tiles < 12000 ? 32 : 0; // (the lane slots of PT_CARRY_LAST take 3 KB of the workgroup's LDS budget)
#else
        a.parkedMax = PARKED_MAX;
#endif
        const Tuning &tune = tuning(); // (pt_tuning.hpp: A/B knobs, set through pt_debug_set only)
        if (tune.parkedMax >= 0) a.parkedMax = tune.parkedMax > PARKED_MAX ? PARKED_MAX : tune.parkedMax;
        const bool spp1 = a.spp == 1; // tile-pass kernels (the ring holds 60-byte paths instead of 40-byte primary rays)
        // spp > 1: the batch-pass kernel (every sample's first bounce coherent and culled), unless drain compaction is asked
        // for (single-launch frames of the A/B variants and of caller-owned streams keep the in-lane sample chain)
        const bool noBatchPass = tune.noBatchPass != 0; // A/B runs
        // ... and unless frames are pipelined over a SMALL image.  Inside a tagged launch a finished pixel may wait for its
        // previous frame; the batch-pass kernel keeps work outside the lanes (parked continuations), and when consecutive frames
        // of a tile meet in one wavefront — few tiles per frame for the ~5,000 resident wavefronts — every lane, and then the
        // whole queue, can fill up with results that wait for exactly that parked work.  Round 2 saw that end in the stall bound's
        // error code (225 tiles per frame, 32 frames: 2-4 % of launches).  Round 3: the kernel's FORCED BATCH PASS (a wavefront
        // whose lanes all wait runs a pass over its oldest parked records anyway) makes the parked work progress whatever the lanes
        // hold, so the cycle cannot form any more (tools/handover_stress --multisample with PT_BATCH_PASS_MIN_TILES=0 sends every
        // such launch through this kernel).  Small images still take the in-lane sample chain — for speed: when consecutive frames
        // of a tile meet in one wavefront all the time, the queue mostly rotates waiting records.  16,384 tiles per frame
        // (1024 x 1024) leave a wavefront 3-4 tiles per frame.
        const long long batchPassMinTiles = tune.batchPassMinTiles; // (16,384; 0: stress runs)
        const bool smallPipelined = a.tagged && (long long)a.tilesX * a.tilesY < batchPassMinTiles;
        const bool useBatchPass = !spp1 && a.drainCompaction == 0 && !noBatchPass && !smallPipelined;
        // the continuation queues take what a 5-per-CU workgroup has left next to the scene and the rings (<= 128 entries per wavefront)
        int park = 0;
        if (useBatchPass) {
            // (31 KB per workgroup: measured, a 32.3 KB workgroup no longer fits five times into the CU's 160 KB)
            const long long left = 31ll * 1024 - (long long)lds - (long long)waves * 64 * (long long)sizeof(PathEntryM);
            park = (int)(left / (long long)(waves * sizeof(ContEntry))) & ~7;
            if (tune.parkCapacity >= 0) park = tune.parkCapacity & ~7; // A/B runs
            if (park > 128) park = 128;
            if (park < 64) park = 64; // (then the scene's materials leave LDS below)
        }
        a.contCapacity = park;
        a.contBatchMin = tune.parkMin;
        const size_t queues = useBatchPass ? (size_t)waves * (64 * sizeof(PathEntryM) + (size_t)park * sizeof(ContEntry))
                              : (spp1 ? frame_weight_bytes(a.batchFrames) : 0) + (size_t)waves * 64 * (spp1 ? sizeof(PathEntry) + kLaneLastBytes : sizeof(RingEntry)) + (a.drainCompaction != 0 ? (size_t)pool_slots(waves) * sizeof(PathState) : 0) // no pool without drain compaction
                                + (spp1 && a.tagged && a.drainCompaction == 0 ? (size_t)waves * a.parkedMax * sizeof(ParkedResolve) : 0); // parked resolves of tagged launches
        // Large scenes: the generic bounce walks the sphere grid (ray_trace_t<GRID>); the grid rides in LDS next to the scene
        const bool noGrid = tune.noSphereGrid != 0; // A/B runs
        const bool useGrid = a.grid != nullptr && a.gridBytes > 0 && !noGrid && !a.timeline;
        a.gridLdsBytes = useGrid ? (a.gridBytes + 15) & ~15 : 0;
        lds += (size_t)a.gridLdsBytes;
        size_t ldsTotal = lds + queues;
        // materials leave LDS when they would cost a resident workgroup (160 KB per CU; 64 B of static LDS per workgroup)
        const size_t ldsPerCU = 160 * 1024, fixedLds = 64; // (static LDS of the persistent kernels: queue, drain control)
        const size_t ldsLean = scene_lds_bytes(a.numSpheres, a.numCuboids, a.envFormat, false, a.gridLdsBytes) + queues;
        size_t wgFull = ldsPerCU / (ldsTotal + fixedLds), wgLean = ldsPerCU / (ldsLean + fixedLds);
        if (wgFull > (size_t)blocksPerCU) wgFull = (size_t)blocksPerCU;
        if (wgLean > (size_t)blocksPerCU) wgLean = (size_t)blocksPerCU;
        const bool forceLean = tune.forceLeanLds != 0; // A/B runs: materials always from the UBO copy
        if (wgLean > wgFull || forceLean || useGrid) { // (the grid kernel is only instantiated for materials in device memory)
            a.materialsInLds = 0;
            ldsTotal = ldsLean;
        }
#ifndef PT_GRID_MIN_WAVES
#define PT_GRID_MIN_WAVES 6
#endif
#ifndef PT_SPP1_WAVES
#define PT_SPP1_WAVES 6
#endif
#define PT_LAUNCH_PERSISTENT(TL, S1, ML) \
    hipLaunchKernelGGL((pt_integrate_persistent_kernel<4, (S1 ? PT_SPP1_WAVES : 5), TL, S1, ML>), dim3(nwg), dim3(256), ldsTotal, stream, a)
        const bool matLds = a.materialsInLds != 0;
        if (a.timeline && spp1 && matLds) PT_LAUNCH_PERSISTENT(true, true, true); // per-wavefront timestamps (tools/timeline.py)
        else if (spp1 && matLds) PT_LAUNCH_PERSISTENT(false, true, true);
        else if (spp1 && useGrid) hipLaunchKernelGGL((pt_integrate_persistent_kernel<4, PT_GRID_MIN_WAVES, false, true, false, true>), dim3(nwg), dim3(256), ldsTotal, stream, a);
        else if (spp1) PT_LAUNCH_PERSISTENT(false, true, false);
        else if (useBatchPass && matLds) hipLaunchKernelGGL(pt_integrate_multisample_kernel<true>, dim3(nwg), dim3(256), ldsTotal, stream, a);
        else if (useBatchPass && useGrid) hipLaunchKernelGGL((pt_integrate_multisample_kernel<false, true>), dim3(nwg), dim3(256), ldsTotal, stream, a);
        else if (useBatchPass) hipLaunchKernelGGL(pt_integrate_multisample_kernel<false>, dim3(nwg), dim3(256), ldsTotal, stream, a);
        else if (matLds) PT_LAUNCH_PERSISTENT(false, false, true);
        else if (useGrid) hipLaunchKernelGGL((pt_integrate_persistent_kernel<4, 5, false, false, false, true>), dim3(nwg), dim3(256), ldsTotal, stream, a);
        else PT_LAUNCH_PERSISTENT(false, false, false);
#undef PT_LAUNCH_PERSISTENT
        // every workgroup draws tickets until its first failing one: (numChunks - nwg) successful + nwg failing
        // (a pipelined batch draws every chunk dynamically: numChunks successful + nwg failing)
        *ticketsConsumed = a.tagged ? (unsigned int)(numChunks + nwg) : (unsigned int)(numChunks > nwg ? numChunks : nwg);
        if (workgroups) *workgroups = nwg;
    } else {
        int poolTiles = pool_tiles_for_variant(a.variant);
        int pools = (tiles + poolTiles - 1) / poolTiles;
        int nwg = (pools + 3) / 4;
        hipLaunchKernelGGL(pt_integrate_pool_kernel, dim3(nwg), dim3(256), lds, stream, a, poolTiles);
    }
    return hipGetLastError();
}

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
