// pt_integrate_persistent.hip — hand-written HIP kernels for gfx950 (CDNA4): the path-tracing integrator (the simple
// one-tile-per-wavefront kernel, the wave-local pool kernel and the persistent default kernel) and the launch dispatch.
// (spp > 1 batch-pass kernel: pt_integrate_multisample.hip; clear / tone map / gather / atmosphere: pt_helper_kernels.hip.)  Written from the algorithm, not transpiled: the control structure, data
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
#include <cstdio>

#include "pt_kernel_common.hpp"
#include "pt_tuning.hpp"

namespace pt {

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

// ---- the repair pass of the hand-over bound (pt_kernel_common.hpp): what an ABANDONED tagged launch left undone.  Enqueued by the host
// behind a join of the handle's streams for every launch since the previous join (FrameArgs by value: the launch's own inputs), so it
// runs with nothing else of the handle in flight and plain loads and stores do; when no launch was abandoned — always, outside a
// contended device — every workgroup leaves after one load.  One lane per pixel, a small grid striding over the tiles.  The pixel's
// alpha says which frame it holds: tag(f) of this launch -> frames up to f are in; a tag of a later launch, or the 1 that the last
// launch of a chain stores last -> all of them are; anything else -> none (its predecessor's last tag, the 1 / 0 of an image no tagged
// launch has touched; FrameArgs::tileFlags tells the two meanings of 1 apart).  The missing frames are folded in frame order with the
// arithmetic every kernel uses (compute.glsl:101-130), so the image is bit for bit the one the undisturbed launch would have left.
// Launches are repaired in launch order (stream order of these kernels); a pixel is always handled by the same lane of the same
// workgroup.  ctl: [0] += (pixel, frame) pairs rendered, [1] += pixels whose tag fits nothing the launch sequence can have left.
__global__ __launch_bounds__(256) void pt_repair_kernel(const FrameArgs a, unsigned int *ctl)
{
    if (*(const volatile unsigned int *)a.abandonWord > a.launchSeq) return; // (uniform: nothing of this launch is missing)
    SceneLds sc = stage_scene(a);
    EnvRef env{a.env, (LdsFloats)sc.lut, a.envSize, a.envFormat};
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tiles = a.tilesX * a.tilesY, n = a.batchFrames;
#ifdef PT_PROFILE
    unsigned long long prof_dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int tile = (int)blockIdx.x * 4 + wave; tile < tiles; tile += (int)gridDim.x * 4) {
        const int tx = tile % a.tilesX, ty = tile / a.tilesX;
        const int px = tx * 8 + (lane & 7), ly = ty * 8 + (lane >> 3);
        if (px >= a.width || ly >= a.rows) continue;
        const int gy = global_row(a, ly);
        const size_t idx = (size_t)ly * a.width + px;
        float4 last = a.accum[idx];
        int done = 0, odd = 0;
        if (last.w >= FRAME_TAG) {
            const int rel = ((int)last.w - (int)FRAME_TAG - a.frame) & kFrameTagMask;
            if (rel < n) done = rel + 1;
            else if (rel <= kFrameTagMask / 2) done = n; // a later launch got this far: every frame of this one is in (window: pt_kernels.hpp)
            else if (last.w != a.chainTag) odd = 1; // older than the predecessor's last frame: the launches before were not complete
        } else if (last.w == 1.0f) {
            // 1 is what the last frame of a launch that does not keep its tags stores — this launch's, or a later one's of the same chain —
            // and what an image no tagged launch has touched holds.  A chained launch never meets the latter (its pixels start from the
            // predecessor's tag); a launch that starts a chain knows from its tile flags whether its first frame ran on the tile (then
            // the pixel held a tag at some point, and 1 means "finished")
            done = (a.chainTag != 0.0f || (a.tileFlags && a.tileFlags[tile] == a.launchSeq)) ? n : 0;
        } else if (a.chainTag != 0.0f) {
            odd = 1; // (a chained launch never meets an untagged unfinished pixel)
        }
        for (int j = done; j < n; j++) {
            uint32_t seed = pixel_seed(px, gy, a.frame + j); // compute.glsl:106
            v3 irr = V(0.0f, 0.0f, 0.0f);
            for (int sidx = 0; sidx < a.spp; sidx++) {
                v3 ro, rd;
                primary_ray(a, px, gy, seed, ro, rd);
                irr = v_add(irr, radiance(a, sc, env, ro, rd, seed));
            }
            irr = v_scale(irr, f_div_ieee(1.0f, (float)a.spp));
            const float w = f_div_ieee(1.0f, (float)(a.frame + j + 1));
            const float alpha = (j == n - 1 && !a.keepTags) ? 1.0f : frame_tag(a.frame + j);
            const float4 next = make_float4(f_mix(last.x, irr.x, w), f_mix(last.y, irr.y, w), f_mix(last.z, irr.z, w), alpha);
            AUDIT_RESOLVE(a, idx, a.frame + j, last, next, 8);
            last = next;
        }
        if (done < n) {
            a.accum[idx] = last;
            if (a.snapshot) a.snapshot[idx] = make_float4(last.x, last.y, last.z, 1.0f);
            atomicAdd(ctl, (unsigned int)(n - done));
        }
        if (odd) atomicAdd(ctl + 1, 1u);
    }
}

// ... and behind the repair passes of a join: the handle's abandon word and ticket counters as a fresh chain expects them (an abandoned
// launch draws fewer tickets than the host accounted for).  ctl[2] counts the joins that had something to repair.
__global__ void pt_repair_done_kernel(unsigned int *abandonWord, unsigned int *queueMain, unsigned int expectMain, unsigned int *queueChain,
                                      unsigned int expectChain, unsigned int *ctl)
{
    if (*abandonWord == ABANDON_NONE) return;
    *abandonWord = ABANDON_NONE;
    *queueMain = expectMain;
    *queueChain = expectChain;
    ctl[2] += 1u;
}

hipError_t launch_repair(const FrameArgs &args, unsigned int *ctl, hipStream_t stream)
{
    FrameArgs a = args;
    a.materialsInLds = 1;
    a.gridLdsBytes = 0;
    a.tileMasks = nullptr;
    a.timeline = nullptr;
    const int tiles = a.tilesX * a.tilesY;
    if (tiles <= 0 || a.abandonWord == nullptr) return hipSuccess;
    int nwg = 2 * a.numCUs;
    if (nwg > (tiles + 3) / 4) nwg = (tiles + 3) / 4;
    const size_t lds = scene_lds_bytes(a.numSpheres, a.numCuboids, a.envFormat, true);
    hipLaunchKernelGGL(pt_repair_kernel, dim3(nwg), dim3(256), lds, stream, a, ctl);
    return hipGetLastError();
}

hipError_t launch_repair_done(unsigned int *abandonWord, unsigned int *queueMain, unsigned int expectMain, unsigned int *queueChain,
                              unsigned int expectChain, unsigned int *ctl, hipStream_t stream)
{
    hipLaunchKernelGGL(pt_repair_done_kernel, dim3(1), dim3(1), 0, stream, abandonWord, queueMain, expectMain, queueChain, expectChain, ctl);
    return hipGetLastError();
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
// COMPACT = false: the donate / adopt code of the drain compaction is compiled out (pipelined launches never use it, and while it sits in
// the main loop the compiler keeps a second copy of the path state around it: "feed" was 20 % of the 256-sphere scene's wavefront time).
// FEED = true: a FRAME-FED launch (FrameArgs::feedHost, round 6) — the launch is started with room for kFeedCapacity frames and begins a
// frame when the host has published it (PathTracer.Render() of a host that shows every frame: the wavefronts stay resident between frames
// instead of draining and being launched again).  Differences, all behind `if constexpr (FEED)`: tickets come through queue_pop_tile_feed
// (a wavefront may be told "not yet"), a wavefront with nothing to trace and nothing published waits (bounded: feedIdleTicks, then the
// launch is abandoned with reason "idle"), resolved pixels are counted per frame (feedDone: the present's gate), and every frame may
// store a present snapshot (snapshots[]).  The arithmetic per pixel is untouched: bit-identical to every other kernel.
template <int NWAVES, int MIN_WAVES_PER_SIMD, bool TIMELINE, bool SPP1, bool MATLDS, bool GRID = false, bool CARRY = false, bool COMPACT = true, bool FEED = false>
__global__ __launch_bounds__(NWAVES * 64, MIN_WAVES_PER_SIMD) void pt_integrate_persistent_kernel(const FrameArgs a)
{
    static_assert(!(CARRY && COMPACT), "the carrying kernels are only launched without drain compaction");
    static_assert(!FEED || (SPP1 && !COMPACT && !TIMELINE), "frame-fed launches: spp = 1, no drain compaction");
    __shared__ __attribute__((aligned(16))) BlockQueue queue; // 16 B: keeps the dynamic-LDS base 16-byte aligned
    __shared__ __attribute__((aligned(16))) DrainControl drain; // 32 B
    __shared__ __attribute__((aligned(16))) FeedQueue feedq;    // 16 B (FEED kernels only; the others never touch it: not allocated)
    __shared__ unsigned int wgDone[8];                          // FEED: pixels resolved per frame slot by this workgroup whose stores are complete, since its last flush
    __shared__ unsigned int waveDone[FEED ? NWAVES * 8 : 1];    // FEED: ... per wavefront, not yet released (their stores may still be on their way)
    const int numTilesFrame = a.tilesX * a.tilesY;
    const int numTiles = numTilesFrame * a.batchFrames;         // (frame, tile) pairs, frame-major
    // 1 / (frame + j + 1): running-mean weight of the batch's frame j.  In DYNAMIC LDS between the scene and the rings, sized by the
    // launch (64 entries; 256 only for the long batches of small shares): as a static 1 KB table it cost the 256-sphere scene its
    // sixth workgroup per CU (27.9 instead of 27.2 KB)
    // (the table's address is re-derived from the kernarg segment where it is read — see cold_args — instead of living in a register
    // across the bounce loop)
    auto frame_weights = [&]() -> float2 * { return (float2 *)((char *)g_lds + cold_args()->sceneLdsBytes); };
    if (SPP1) {
        float2 *fw = frame_weights();
        for (int j = (int)threadIdx.x; j < a.batchFrames; j += (int)blockDim.x)
            fw[j] = make_float2(f_div_ieee(1.0f, (float)(a.frame + j + 1)), (j == a.batchFrames - 1 && !a.keepTags) ? 1.0f : frame_tag(a.frame + j));
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
        if constexpr (FEED) {
            feedq.stash = 0ull;
            feedq.limit = 0u; // (refreshed by the first refill)
            feedq.word = 0u;
            feedq.pollAt = 0u;
        }
        if (a.startedFlags) // "this workgroup is resident" (launch chaining): a system-scope store, the host polls the word
            __hip_atomic_store(a.startedFlags + blockIdx.x, a.launchSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef PT_AUDIT
        // (audit build: a tagged launch that STARTS with a ticket counter outside its own range of the host's base draws only failing tickets
        // and renders nothing — logged like a violation: site 91, "pix" = the counter, "frame" = the host's base)
        if (a.tagged && a.audit && a.auditLog && blockIdx.x == 0) {
            const unsigned int c = __hip_atomic_load((const unsigned int *)a.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned int)(c - a.queueBase) > (unsigned int)numTiles + 65536u) {
                const unsigned int slot = atomicAdd(a.auditLog, 1u);
                if (slot < (unsigned int)kAuditLogRecords) {
                    unsigned int *r = a.auditLog + 4 + slot * kAuditRecordWords;
                    r[0] = 91u; r[1] = c; r[2] = a.queueBase; r[3] = 0u; r[4] = 0u; r[5] = 0u; r[6] = 0u; r[7] = a.launchSeq;
                    r[8] = (unsigned int)a.frame | ((unsigned int)a.batchFrames << 24); r[9] = __float_as_uint(a.chainTag); r[10] = 0u;
                    r[11] = (unsigned int)a.tagged | ((unsigned int)a.keepTags << 1) | ((unsigned int)a.variant << 8);
                }
            }
        }
#endif
    }
    if constexpr (FEED) {
        if (threadIdx.x < 8) wgDone[threadIdx.x] = 0u;
        if (threadIdx.x < NWAVES * 8) waveDone[threadIdx.x] = 0u;
    }
    CHAOS(1);
    SceneLds sc = stage_scene(a); // ends with __syncthreads()
    EnvRef env{nullptr, (LdsFloats)sc.lut, 0, 0}; // descriptor is cold-loaded at the miss-shading site (bounce_step)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // FEED: a resolved pixel of frame j of the launch counts into slot j & 7 of its WAVEFRONT's LDS counters (one ds_add per resolving
    // lane).  release_resolved() — called where the wavefront has just waited for its memory operations anyway (the tile pass, behind the
    // tile's pixel load) — moves them to the WORKGROUP's counters: only pixels whose stores (image + display) are complete are ever
    // counted on.  flush_resolved() — the wavefront that starts a new ticket, or runs out of work — moves the workgroup's counters to
    // FrameArgs::feedDone: one device atomic per non-empty slot per ticket (~4,000 per 1080p frame; one per wavefront iteration was
    // 86,000, and the atomics of one address serialise at ~8 ns each: 0.69 ms per frame).
    [[maybe_unused]] unsigned int *const myDone = waveDone + (FEED ? wave * 8 : 0);
    [[maybe_unused]] auto count_resolved = [&](int rfj) -> void {
        if constexpr (FEED) atomicAdd(myDone + (rfj & 7), 1u);
    };
    [[maybe_unused]] auto release_resolved = [&]() -> void {
        if constexpr (FEED) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane < 8) {
                const unsigned int v = lds_load(myDone + lane);
                if (v != 0u) {
                    lds_store(myDone + lane, 0u);
                    atomicAdd(wgDone + lane, v);
                }
            }
        }
    };
    [[maybe_unused]] auto flush_resolved = [&]() -> void {
        if constexpr (FEED) {
            if (lane < 8) {
                const unsigned int v = atomicExch(wgDone + lane, 0u);
                if (v != 0u) __hip_atomic_fetch_add(cold_args()->feedDone + lane * kFeedDoneStride, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    // FEED, fused display (FrameArgs::displayImages): the present slot + 1 of frame rfj - 1 of the launch (0: that frame is not shown), from
    // the workgroup's copy of the feed word — whoever is handed a tile of frame rfj has seen a word whose count includes it, and the host
    // puts the slot of frame rfj - 1 into the word that publishes frame rfj.  Wave-uniform for a wave-uniform rfj.
    [[maybe_unused]] auto display_slot_before = [&](int rfj) -> unsigned int {
        if constexpr (FEED) {
            if (rfj == 0) return (unsigned int)cold_args()->displayPrev;
            return (lds_load(&feedq.word) >> (2 * ((rfj - 1) & 7))) & 3u;
        } else {
            return 0u;
        }
    };
    // PostProcessing/fragment.glsl:17-26 for one pixel: ACES + gamma 2.4 -> RGBA8 (the arithmetic of pt_postprocess_kernel)
    [[maybe_unused]] auto display_pixel = [&](unsigned int slot, int rpix, float r, float g, float b) -> void {
        if constexpr (FEED) {
            ColdArgs ca = cold_args();
            uchar4 *img = slot == 1u ? ca->displayImages[0] : (slot == 2u ? ca->displayImages[1] : ca->displayImages[2]);
            uchar4 o;
            o.x = to_unorm8(linear_to_inverse_gamma(aces_film(r), 2.4f));
            o.y = to_unorm8(linear_to_inverse_gamma(aces_film(g), 2.4f));
            o.z = to_unorm8(linear_to_inverse_gamma(aces_film(b), 2.4f));
            o.w = 255;
            unsigned int packed;
            __builtin_memcpy(&packed, &o, 4);
            asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" : : "v"(img + rpix), "v"(packed) : "memory"); // (write-through: read by the host's copy / GL while this launch runs)
        }
    };
    // the present snapshot of the launch's LAST frame, if the launch feeds a present (classic launches; a fed launch shows its frames
    // through the fused display instead)
    [[maybe_unused]] auto store_snapshot = [&](int rpix, int rfj, float4 next) -> void {
        if constexpr (!FEED) {
            if (float4 *snap = cold_args()->snapshot) // (wave-uniform, almost always null)
                if (rfj == cold_args()->batchFrames - 1) snap[rpix] = make_float4(next.x, next.y, next.z, 1.0f);
        }
    };

    // the ring lives behind the staged scene in dynamic LDS
    static_assert(!CARRY || SPP1, "the pixel travels with the path in the tile-pass kernels only");
    using PathRec = typename std::conditional<CARRY, PathEntryCarry, PathEntry>::type;
    constexpr int ENTRY_BYTES = SPP1 ? (int)sizeof(PathRec) : (int)sizeof(RingEntry);
    char *ringBase = (char *)g_lds + scene_lds_bytes(a.numSpheres, a.numCuboids, a.envFormat, a.materialsInLds != 0, a.gridLdsBytes) +
                     (SPP1 ? frame_weight_bytes(a.batchFrames) : 0);
    RingEntry *ring = (RingEntry *)(ringBase + wave * 64 * ENTRY_BYTES);  // !SPP1: primary rays
    PathRec *pring = (PathRec *)(ringBase + wave * 64 * ENTRY_BYTES); //  SPP1: paths after their first bounce
    // CARRY: per-lane slots for the pixel value the tile pass read (PathEntryCarry::last travels here when a lane pops the path): three
    // planes of 64 floats per wavefront, so that keeping it costs no registers across the bounce loop
    constexpr int LANE_LAST_BYTES = CARRY ? NWAVES * 3 * 64 * 4 : 0;
    float *laneLast = (float *)(ringBase + NWAVES * 64 * ENTRY_BYTES) + wave * 3 * 64 + lane; // (only touched by CARRY kernels)
    PathState *pool = (PathState *)(ringBase + NWAVES * 64 * ENTRY_BYTES + LANE_LAST_BYTES);
    const bool compaction = COMPACT && a.drainCompaction != 0;
    // parked resolves of this wavefront (pipelined spp = 1 launches only; behind the rings — such launches have no drain pool)
    ParkedResolve *parkedList = (ParkedResolve *)(ringBase + NWAVES * 64 * ENTRY_BYTES + LANE_LAST_BYTES) + wave * a.parkedMax;
    const bool parking = SPP1 && a.tagged && !compaction && a.parkedMax > 0;
    int nparked = 0; // wave-uniform
    bool parkProgress = false;         // ... the last service round resolved at least one parked entry
    unsigned int parkSince = 0u;       // ... wait_clock() | 1 since when the list has made no progress (0: empty, or it just did)
    HandoverBound bound;               // the hand-over's wall-clock bound (pt_kernel_common.hpp)
    bound.init();
    const int donateMax = a.drainCompaction < DONATE_MAX ? a.drainCompaction : DONATE_MAX; // a wavefront this thin donates
    const bool leader = lane == 0;

    if constexpr (FEED) {
        // ---- the MONITOR: the last wavefront of workgroup 0 takes no tiles.  It is the launch's only reader of the host word (PCIe), which
        // it broadcasts to the workgroups' slots (FrameArgs::feedBcast) whenever it changes, and it tells the host which frames are
        // complete (FrameArgs::feedHostDone).
        if (blockIdx.x == 0 && wave == NWAVES - 1) {
            ColdArgs ca = cold_args();
            unsigned int j = 0;      // frames [0, j) of the launch are complete
            unsigned int seen = 0u;  // the word broadcast last (0: none yet — the slots were zeroed before the launch)
            for (;;) {
                unsigned int w = __hip_atomic_load(ca->feedHost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w = (unsigned int)__builtin_amdgcn_readfirstlane((int)w);
                if (w != seen) { // (the word only ever grows: a newer count, then the closed bit)
#ifdef PT_FEED_TIMES
                    if (lane == 0 && a.startedFlags) __hip_atomic_store(a.startedFlags + 3100 + (feed_count(w) & 63), (unsigned int)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
                    for (int s4 = lane; s4 < kFeedBcastSlots; s4 += 64)
                        __hip_atomic_store(ca->feedBcast + s4 * kFeedBcastStride, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    seen = w;
                }
                while (j < (unsigned int)ca->batchFrames) {
                    const unsigned long long need = ca->feedBase[j & 7] + ca->feedPixels * (unsigned long long)(j / 8 + 1);
                    const unsigned long long have = __hip_atomic_load(ca->feedDone + (j & 7) * kFeedDoneStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!__builtin_amdgcn_readfirstlane((int)(have >= need))) break;
                    j++;
                    if (lane == 0) __hip_atomic_store(ca->feedHostDone, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef PT_FEED_TIMES
                    if (lane == 0 && a.startedFlags) __hip_atomic_store(a.startedFlags + 3000 + (j & 63), (unsigned int)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
                }
                if (((w & kFeedClosed) && j >= feed_count(w)) || j >= (unsigned int)ca->batchFrames) break; // every frame the launch will ever have is complete
                if (launch_abandoned()) break;
                __builtin_amdgcn_s_sleep(8);
            }
            return;
        }
    }
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
    // frame" flag, and since when it waits
    int fj = 0;
    bool pending = false;
    // GRID kernels: the parameter from which the grid walk of the current bounce continues (>= 0 while it is unfinished, else -1:
    // pt_device.hpp, WALK SLICES); dead in the other kernels
    float walkFrom = -1.0f, walkFresh = -1.0f;
    // every site that loads a path into the lane goes through begin_path() (the wait of a finished path for its pixel's previous frame is
    // timed per wavefront: HandoverBound)
    auto begin_path = [&]() -> void { pending = false; walkFrom = -1.0f; };
    // (CARRY: bit 14 of fj = the lane's slot of laneLast holds the pixel's accumulation value as the tile pass read it)

    // compute.glsl:125-129 for one finished path of frame `rfj` of the batch.  False = the pixel still holds an older
    // frame (only possible inside a batch): try again in the next iteration.
    // compute.glsl:125-129 for frame `rfj` of the batch: irradiance / SPP folded into the running mean, alpha = 1 (or the
    // frame tag inside a batch).  The uniform inputs are re-read from the kernarg segment here (see cold_args).
    auto fold = [&](float4 last, v3 rirr, int rfj) -> float4 {
        float w, alpha;
        if constexpr (SPP1) { // irradiance / 1 is exact (x * 1.0f == x bit for bit): skipped; weight and alpha from the per-batch table
            const float2 wa = frame_weights()[rfj];
            w = wa.x;
            alpha = wa.y;
        } else {
            ColdArgs ca = cold_args();
            rirr = v_scale(rirr, f_div_ieee(1.0f, (float)ca->spp));
            w = f_div_ieee(1.0f, (float)(ca->frame + rfj + 1));
            alpha = (rfj == ca->batchFrames - 1 && !ca->keepTags) ? 1.0f : frame_tag(ca->frame + rfj);
        }
        return make_float4(f_mix(last.x, rirr.x, w), f_mix(last.y, rirr.y, w), f_mix(last.z, rirr.z, w), alpha);
    };
    // CARRY: compute.glsl:126-129 with the pixel's value already in hand (read by the tile pass, which also checked the tag): no load
    [[maybe_unused]] auto commit_resolve = [&](int rpix, int rfj, v3 rirr, v3 rlast) -> void {
        const float4 last = make_float4(rlast.x, rlast.y, rlast.z, 0.0f);
        const float4 next = fold(last, rirr, rfj);
        AUDIT_RESOLVE(a, (size_t)rpix, a.frame + rfj, last, next, 7);
        if (!a.tagged) a.accum[rpix] = next;
        else store_pixel_sc1(a.accum + rpix, next);
        store_snapshot(rpix, rfj, next);
        count_resolved(rfj);
    };
    // False = the pixel still holds an older frame (only possible inside a batch): try again in the next iteration.
    auto try_resolve = [&](int rpix, int rfj, v3 rirr) -> bool {
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
        if (expected != 0.0f && last.w != expected) return false;
        if constexpr (FEED) { // fused display: this pixel's tile pass could not show frame rfj - 1 (its value was not there yet): here it is
            if (cold_args()->displayOn != 0) {
                const unsigned int dslot = (rfj == 0) ? (unsigned int)cold_args()->displayPrev : ((lds_load(&feedq.word) >> (2 * ((rfj - 1) & 7))) & 3u);
                if (dslot != 0u) display_pixel(dslot, rpix, last.x, last.y, last.z);
            }
        }
        if (AUDIT_SABOTAGED(a, rpix, rfj)) last.x += 1.0f; // (audit build + PT_AUDIT_SABOTAGE only: a simulated stale / torn read)
        CHAOS(11);
        const float4 next = fold(last, rirr, rfj);
        AUDIT_RESOLVE(a, (size_t)rpix, a.frame + rfj, last, next, 4);
        store_pixel_sc1(ptr, next);
        store_snapshot(rpix, rfj, next); // the launch's last frame (FEED: every frame): the present snapshot
        count_resolved(rfj);
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
    auto service_parked = [&]() -> void {
        parkProgress = false;
        if (nparked == 0) return;
        CHAOS(4);
        const bool mine = lane < nparked;
        ParkedResolve e = {0, 0, {0.0f, 0.0f, 0.0f}};
        bool keep = false;
        if (mine) {
            e = parkedList[lane];
            keep = !try_resolve(e.pix, e.fj, V(e.irr[0], e.irr[1], e.irr[2]));
        }
        const unsigned long long km = __ballot(keep);
        __builtin_amdgcn_wave_barrier(); // every entry has been read before the survivors are written back
        if (keep) parkedList[__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u))] = e;
        const int left = __builtin_popcountll(km);
        parkProgress = left < nparked;
        nparked = left;
        __builtin_amdgcn_wave_barrier();
    };

    [[maybe_unused]] bool starved = false;          // FEED: the queue said "not yet" in this iteration (wave-uniform)
    [[maybe_unused]] unsigned int idlePolls = 0u;
    [[maybe_unused]] unsigned int lastLimit = 0u;   // FEED: feedq.limit when this wavefront last looked (a change = the host published)
    [[maybe_unused]] unsigned int idleSince = 0u;   // FEED: (clock | 1) since when this wavefront has had nothing to trace and nothing published
    for (;;) {
        // ---- feed idle lanes: pop from the ring; if the ring runs dry while lanes are still idle, refill it with the next
        // tile and pop again in the SAME iteration (a lane never idles through a bounce iteration because the ring happened
        // to hold fewer rays than there were idle lanes)
        bool idle = pix < 0;
        unsigned long long m = __ballot(idle);
        if constexpr (FEED) starved = false;
        for (int pass = 0; (SPP1 ? pass < 16 : pass < 2) && m != 0ull; pass++) {
            if (avail == 0) {
                if (exhausted) break;
                    // ---- refill the ring: one tile, every lane generates one primary ray
                    int tile;
                    if constexpr (FEED) {
                        tile = queue_pop_tile_feed(&queue, &feedq);
                        if (tile < 0) release_resolved(); // (no tile: nothing else will make this wavefront wait for its stores)
                        if ((tile & 7) == 0 || tile < 0) flush_resolved(); // (the first tile of a ticket, or no tile: move the workgroup's counts on)
                        if (tile == QUEUE_NOT_YET) { // the next frame is not published yet: go on with what the lanes hold, ask again next iteration
                            starved = true;
                            break;
                        }
                        idleSince = 0u;
                    } else {
                        tile = queue_pop_tile(&queue);
                    }
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
                        const float invW = ca->invW, invH = ca->invH;
                        int tfj, tx, ty; // frame of the batch this (frame, tile) ticket belongs to; tile column, row
                        fast_divmod(tile, numTilesFrame, ca->tilesFrameMagic, tfj, tile);
                        fast_divmod(tile, tilesX, ca->tilesXMagic, ty, tx);
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
                        unsigned long long masks[5];
                        if (const unsigned long long *cached = ca->tileMasks) { // (wave-uniform tile: four scalar loads, no arithmetic)
                            const __attribute__((address_space(4))) unsigned long long *tm =
                                (const __attribute__((address_space(4))) unsigned long long *)cached + (size_t)tile * kTileMaskWords;
                            masks[0] = tm[0]; masks[1] = tm[1]; masks[2] = tm[2]; masks[3] = tm[3]; masks[4] = tm[4];
                        } else {
                            cull_spheres(sc, a.numSpheres, valid, to, td, masks);
                        }
                        // (a launch that starts on alpha = 1 and ends on alpha = 1 notes where its first frame ran: all the repair pass has to
                        // tell an untouched pixel from a finished one, see FrameArgs::tileFlags)
                        if (unsigned int *flags = ca->tileFlags)
                            if (tfj == 0 && lane == 0) flags[tile] = ca->launchSeq;
                        bool tcont = false, tkeep = false; // tkeep: the path goes to the ring (it continues, or its resolve must wait)
                        [[maybe_unused]] bool plastOk = false; // (CARRY)
                        [[maybe_unused]] float4 plast = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if (valid) {
                            if (0 < a.rayDepth)
                                tcont = bounce_step_t<true, MATLDS>(sc, a.numSpheres, a.numCuboids, env, to, td, tthr, trad, tseed, masks, walkFresh PROF_DUMMY);
                            if (1 >= a.rayDepth) tcont = false;
                            tkeep = tcont;
                            if constexpr (CARRY) {
                                // imageLoad (compute.glsl:126) for the whole tile, all lanes together: 8 rows x 128 B = full lines, one memory
                                // round trip per TILE instead of one per bounce iteration.  (After the first bounce: held across it, the four
                                // registers spill.)  A plain load: a stale cached copy can only show an OLDER tag, and then the pixel takes the
                                // coherent load of try_resolve when its path ends.
                                plast = a.accum[tpix];
                                const float expected = tfj > 0 ? frame_tag(a.frame + tfj - 1) : a.chainTag;
                                plastOk = !a.tagged || expected == 0.0f || plast.w == expected;
                                if constexpr (FEED) {
                                    // (the wavefront has just waited for the tile's pixels: every store it issued before — the pixels it
                                    // has counted so far — is complete; the resolves below are counted now and released by the NEXT tile pass)
                                    release_resolved();
                                    // FUSED DISPLAY: the tile's 64 pixels as frame tfj - 1 left them are in registers, all lanes
                                    // together — if that frame is shown, this is its tone map (a lane whose previous frame is not in yet
                                    // shows it when its own resolve loads the pixel: try_resolve)
                                    if (cold_args()->displayOn != 0) { // (wave-uniform)
                                        const unsigned int dslot = display_slot_before(tfj);
                                        if (dslot != 0u && plastOk) display_pixel(dslot, tpix, plast.x, plast.y, plast.z);
                                    }
                                }
                            }
                            if (CARRY && !tcont && plastOk) { // ended at its first bounce, previous frame already there: fold and store, no second load
                                commit_resolve(tpix, tfj, v_add(V(0.0f, 0.0f, 0.0f), trad), V(plast.x, plast.y, plast.z));
                            } else if (!tcont) { // the path ended at its first bounce: compute.glsl:125-129 right away
                                v3 tirr = v_add(V(0.0f, 0.0f, 0.0f), trad);
                                tkeep = !try_resolve(tpix, tfj, tirr);
                            }
                        }
                        if (parking) { // resolves that have to wait for the previous frame: parked (else through the ring)
                            const bool twait = valid && !tcont && tkeep;
                            if (park_resolves(twait, tpix, tfj, v_add(V(0.0f, 0.0f, 0.0f), trad))) tkeep = false;
                        }
                        const unsigned long long cm = __ballot(tkeep);
                        if (tkeep) {
                            int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
                            PathRec e;
                            // a path whose resolve has to wait re-enters the bounce loop "at full depth": it is resolved there
                            e.pix = tpix; e.bounce = (tcont ? 1 : a.rayDepth) | (tfj << 16); e.seed = tseed;
                            e.ro[0] = to.x; e.ro[1] = to.y; e.ro[2] = to.z;
                            e.rd[0] = td.x; e.rd[1] = td.y; e.rd[2] = td.z;
                            e.thr[0] = tthr.x; e.thr[1] = tthr.y; e.thr[2] = tthr.z;
                            if constexpr (CARRY) {
                                // (one triple: the pixel's value when the path's rad is still +0 — bit for bit — and the value is valid, else rad)
                                const bool radZero = (__float_as_uint(trad.x) | __float_as_uint(trad.y) | __float_as_uint(trad.z)) == 0u;
                                const bool carryLast = plastOk && radZero;
                                if (carryLast) e.bounce |= PATH_HAS_LAST;
                                e.x[0] = carryLast ? plast.x : trad.x; e.x[1] = carryLast ? plast.y : trad.y; e.x[2] = carryLast ? plast.z : trad.z;
                            } else {
                                e.rad[0] = trad.x; e.rad[1] = trad.y; e.rad[2] = trad.z;
                            }
                            pring[slot] = e;
                        }
                        __builtin_amdgcn_wave_barrier(); // ring entries are read by other lanes of this wave below
                        avail = __builtin_popcountll(cm);
                        if constexpr (!CARRY) release_resolved(); // (FEED kernels that do not read the tile's pixels here: an explicit wait for the wavefront's stores)
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
                        const float invW = ca->invW, invH = ca->invH;
                        int tfj, tx, ty; // frame of the batch this (frame, tile) ticket belongs to; tile column, row
                        fast_divmod(tile, numTilesFrame, ca->tilesFrameMagic, tfj, tile);
                        if (unsigned int *flags = ca->tileFlags) // (FrameArgs::tileFlags: where the launch's first frame ran)
                            if (tfj == 0 && lane == 0) flags[tile] = ca->launchSeq;
                        fast_divmod(tile, tilesX, ca->tilesXMagic, ty, tx);
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
                    PathRec e = pring[avail - 1 - rank];
                    pix = e.pix;
                    bounce = e.bounce & 0xffff;
                    fj = (e.bounce >> 16) & 0x3fff;
                    if constexpr (CARRY) {
                        if (e.bounce & PATH_HAS_LAST) fj |= 0x4000;
                        laneLast[0] = e.x[0]; laneLast[64] = e.x[1]; laneLast[128] = e.x[2]; // (only read when PATH_HAS_LAST)
                    }
                    pending = false;   // (= begin_path(), written out: through the lambda the compiler copies the 16 registers of path state
                    walkFrom = -1.0f;  //  twice per pop-loop round — 2.4 M of 110 M vector instructions per 1080p frame)
                    seed = e.seed;
                    ro = V(e.ro[0], e.ro[1], e.ro[2]);
                    rd = V(e.rd[0], e.rd[1], e.rd[2]);
                    throughput = V(e.thr[0], e.thr[1], e.thr[2]);
                    if constexpr (CARRY) {
                        const bool hasLast = (e.bounce & PATH_HAS_LAST) != 0;
                        rad = V(hasLast ? 0.0f : e.x[0], hasLast ? 0.0f : e.x[1], hasLast ? 0.0f : e.x[2]);
                    } else {
                        rad = V(e.rad[0], e.rad[1], e.rad[2]);
                    }
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
                        begin_path();
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
                            begin_path(); // (an adopted path that was waiting starts its wait anew; a donor never holds an unfinished walk: see the donation)
                            pending = (st.counters >> 25) & 1;
                            fj = (st.counters >> 26) & 0x3f;
                            if constexpr (CARRY) {
                                if (st.counters & 1) fj |= 0x4000;
                                laneLast[0] = st.irr[0]; laneLast[64] = st.irr[1]; laneLast[128] = st.irr[2];
                            }
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
        if constexpr (FEED) {
            if (am != 0ull) idleSince = 0u; // (the idle clock runs only while the wavefront has nothing to trace)
        }
        if (am == 0ull) {
            if (parking && nparked > 0) { // nothing to trace: look after the parked resolves (they must be gone before leaving)
                service_parked();
                if (nparked > 0) {
                    if (bound.tick(0ull, parkSince, !parkProgress)) { // (abandoned launch: the host's repair pass renders them)
                        nparked = 0;
                        stop_queue(&queue);
                    }
                }
                if (nparked > 0 && exhausted && avail == 0) __builtin_amdgcn_s_sleep(8);
            }
            if constexpr (FEED) {
                if (starved && !exhausted && nparked == 0) {
                    // nothing to trace, nothing published: wait for the host — bounded; a host that has stopped rendering must not keep
                    // the GPU (hand-over bound, reason "idle": the host's next call joins, repairs what a racing publish left undone,
                    // and launches anew)
                    // Wait HERE, cheaply, until the workgroup's view of the feed word changes (a frame was published, or the launch
                    // closed) — a waiting wavefront must cost the wavefronts still working next to it nothing: going round the main
                    // loop (ballots, the queue's LDS atomics, its lock) every 2 us, five waiting wavefronts per SIMD issued more
                    // instructions than the one still tracing, and a lone frame took 0.5 ms.  One round = a long sleep + two LDS reads;
                    // the workgroup's wavefronts take turns at looking at its broadcast slot (no lock: they would store the same words).
                    // idle = no NEW frame for feedIdleTicks (on a small image most wavefronts never get a tile while the host publishes
                    // away); the clock and the abandon word are looked at every 8th round (~30 us).
                    release_resolved(); // (whatever this wavefront has counted since its last tile — parked results that came in — must not wait here with it)
                    flush_resolved();
                    for (unsigned int round = 0;; round++) {
                        const unsigned int lim = (unsigned int)__builtin_amdgcn_readfirstlane((int)lds_load(&feedq.limit));
                        const unsigned int wd = (unsigned int)__builtin_amdgcn_readfirstlane((int)lds_load(&feedq.word));
                        if (lim != lastLimit || (wd >> 31) != 0u) { // more frames, or closed: back to the queue
                            lastLimit = lim;
                            idleSince = 0u;
                            break;
                        }
                        if ((round & 7u) == 0u) {
                            const unsigned int now = (unsigned int)wall_clock64();
                            if (idleSince == 0u) idleSince = now | 1u;
                            if ((int)(now - idleSince) > (int)cold_args()->feedIdleTicks) {
                                abandon_launch(kAbandonIdle);
                                stop_queue(&queue);
                                break;
                            }
                            if (launch_abandoned()) {
                                stop_queue(&queue);
                                break;
                            }
                        }
                        __builtin_amdgcn_s_sleep(127);
                        if ((round & (NWAVES - 1)) == (unsigned int)wave) { // this wavefront's turn to look at the workgroup's broadcast slot
                            ColdArgs ca = cold_args();
                            unsigned int w = __hip_atomic_load(ca->feedBcast + (blockIdx.x % kFeedBcastSlots) * kFeedBcastStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            w = (unsigned int)__builtin_amdgcn_readfirstlane((int)w);
                            if (lane == 0) {
                                lds_store(&feedq.limit, feed_count(w) * (unsigned int)(ca->tilesX * ca->tilesY));
                                lds_store(&feedq.word, w);
                            }
                        }
                    }
                    continue;
                }
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
                        if (CARRY && (fj & 0x4000)) st.counters |= 1; // (the sample field is unused with one sample per pixel)
                        st.seed = seed;
                        st.ro[0] = ro.x; st.ro[1] = ro.y; st.ro[2] = ro.z;
                        st.rd[0] = rd.x; st.rd[1] = rd.y; st.rd[2] = rd.z;
                        st.thr[0] = throughput.x; st.thr[1] = throughput.y; st.thr[2] = throughput.z;
                        st.rad[0] = rad.x; st.rad[1] = rad.y; st.rad[2] = rad.z;
                        st.irr[0] = SPP1 ? 0.0f : irr.x; st.irr[1] = SPP1 ? 0.0f : irr.y; st.irr[2] = SPP1 ? 0.0f : irr.z;
                        if constexpr (CARRY) { st.irr[0] = laneLast[0]; st.irr[1] = laneLast[64]; st.irr[2] = laneLast[128]; } // (irr is unused with one sample per pixel)
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
            // ONE divergent region around the bounce (not three nested ones: the compiler copies the whole path state — 16 registers — in
            // front of every level of a nest whose inside modifies it)
            const bool trace = active && !pending;
            bool cont = false;
            if (trace && bounce < a.rayDepth) cont = bounce_step_t<false, MATLDS, GRID, (MATLDS && !GRID)>(sc, a.numSpheres, a.numCuboids, env, ro, rd, throughput, rad, seed, nullptr, walkFrom PROF_PASS);
            if (trace && !(GRID && walkFrom >= 0.0f)) { // (else: the grid walk of this bounce continues in the next iteration, pt_device.hpp WALK SLICES)
                bounce++;
                pending = !cont || bounce >= a.rayDepth;
            }
#ifdef PT_PROFILE
            prof_t = __builtin_readcyclecounter();
#endif
            if (CARRY && active && pending && (fj & 0x4000)) { // the tile pass read the pixel (and saw the previous frame's tag): no load, cannot fail
                commit_resolve(pix, fj & 0x3fff, v_add(V(0.0f, 0.0f, 0.0f), rad), V(laneLast[0], laneLast[64], laneLast[128]));
                pix = -1;
                pending = false;
            }
            if (active && pending) {
                v3 firr = v_add(V(0.0f, 0.0f, 0.0f), rad);
                if (try_resolve(pix, fj, firr)) {
                    pix = -1;
                    pending = false;
                }
            }
            if (parking) {
                // a resolve that has to wait gives its lane back: the result is parked, the wavefront's first lanes retry it
                const bool wait = pix >= 0 && pending;
                if (park_resolves(wait, pix, fj, v_add(V(0.0f, 0.0f, 0.0f), rad))) {
                    pix = -1;
                    pending = false;
                }
                service_parked();
            }
            // results that wait for their pixel's previous frame (in their lanes, or parked): the wall-clock bound; an abandoned launch drops them
            const bool waits = pix >= 0 && pending;
            const unsigned long long waitMask = __ballot(waits);
            if (waitMask != 0ull || nparked > 0) {
                if (bound.tick(waitMask, parkSince, nparked > 0 && !parkProgress)) {
                    if (waits) {
                        pix = -1;
                        pending = false;
                    }
                    nparked = 0;
                    stop_queue(&queue);
                } else if (__ballot(pix >= 0 && !pending) == 0ull && avail == 0) {
                    __builtin_amdgcn_s_sleep(8); // nothing but waiting paths left in this wavefront: do not hammer the pixel
                }
            }
        } else {
        if (active && needRay) { // only for spp > 1: the next sample continues the pixel's RNG stream (compute.glsl:110)
            primary_ray(a, px, py, seed, ro, rd);
            throughput = V(1.0f, 1.0f, 1.0f);
            rad = V(0.0f, 0.0f, 0.0f);
            bounce = 0;
            needRay = false;
        }
        {
            const bool trace = active && !pending; // (one divergent region around the bounce, as above)
            bool cont = false;
            if (trace && bounce < a.rayDepth) cont = bounce_step_t<false, MATLDS, GRID>(sc, a.numSpheres, a.numCuboids, env, ro, rd, throughput, rad, seed, nullptr, walkFrom PROF_PASS);
            if (trace) {
                const bool sliced = GRID && walkFrom >= 0.0f; // (the grid walk of this bounce continues in the next iteration)
                if (!sliced) bounce++;
                if (!sliced && (!cont || bounce >= a.rayDepth)) {
                    irr = v_add(irr, rad);
                    sample++;
                    if (sample < a.spp) needRay = true;
                    else pending = true; // the pixel's last sample: fold into the accumulation image
                }
            }
#ifdef PT_PROFILE
            prof_t = __builtin_readcyclecounter();
#endif
            if (active && pending) {
                if (try_resolve(pix, fj, irr)) {
                    pix = -1;
                    pending = false;
                }
            }
        }
        {
            const bool waits = pix >= 0 && pending;
            const unsigned long long waitMask = __ballot(waits);
            if (waitMask != 0ull) {
                unsigned int noList = 0u;
                if (bound.tick(waitMask, noList, false)) { // (abandoned launch: waiting results are dropped, the host's repair pass renders them)
                    if (waits) {
                        pix = -1;
                        pending = false;
                    }
                    stop_queue(&queue);
                } else if (__ballot(pix >= 0 && !pending) == 0ull) {
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        }
        } // !SPP1
        PROF_MARK(7) // resolve
    }
    release_resolved();
    flush_resolved();
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

// ---- cached tile masks (FrameArgs::tileMasks): one wavefront per 8x8 tile, lane j tests sphere j (+64, +128, +192) against the cone that
// bounds every primary ray the tile can cast (tile_cone).  Runs once per camera / scene change, never per frame.
__global__ __launch_bounds__(256) void pt_tile_masks_kernel(const FrameArgs a, unsigned long long *out)
{
    SceneLds sc = stage_scene(a); // (geometry only: the launch clears materialsInLds / envFormat / gridLdsBytes)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.tilesX * a.tilesY) return;
    const int tx = tile % a.tilesX, ty = tile / a.tilesX;
    const int gy0 = global_row(a, ty * 8); // (band heights are multiples of 8: a tile's eight rows are consecutive image rows)
    v3 O, A, dirs[4];
    float rho, ct, lensSin;
    tile_cone<const float *>(a.invProj, a.invW, a.invH, tx * 8, gy0, O, A, rho, ct, dirs, lensSin);
    unsigned long long masks[5];
    cone_sphere_masks(sc, a.numSpheres, O, A, rho, ct, masks);
    masks[4] = cone_cuboid_mask(sc, a.numCuboids, O, rho, ct, dirs, lensSin);
    if (lane == 0) {
        unsigned long long *o = out + (size_t)tile * kTileMaskWords;
        o[0] = masks[0]; o[1] = masks[1]; o[2] = masks[2]; o[3] = masks[3]; o[4] = masks[4]; o[5] = o[6] = o[7] = 0ull;
    }
}

hipError_t launch_tile_masks(const FrameArgs &args, unsigned long long *masks, hipStream_t stream)
{
    FrameArgs a = args;
    a.materialsInLds = 0;
    a.envFormat = 0;
    a.gridLdsBytes = 0;
    const int tiles = a.tilesX * a.tilesY;
    if (tiles <= 0) return hipSuccess;
    const size_t lds = scene_lds_bytes(a.numSpheres, a.numCuboids, 0, false);
    hipLaunchKernelGGL(pt_tile_masks_kernel, dim3((tiles + 3) / 4), dim3(256), lds, stream, a, masks);
    return hipGetLastError();
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

hipError_t launch_integrate(const FrameArgs &args, hipStream_t stream, unsigned int *ticketsConsumed, int *workgroups, bool *fed)
{
    FrameArgs a = args;
    const bool wantFeed = fed != nullptr && *fed;
    if (fed) *fed = false;
    if (!wantFeed) { a.feedHost = nullptr; a.feedBcast = nullptr; a.feedDone = nullptr; a.displayImages[0] = a.displayImages[1] = a.displayImages[2] = nullptr; a.displayPrev = 0; a.displayOn = 0; }
    a.materialsInLds = 1;
    a.gridLdsBytes = 0;
    *ticketsConsumed = 0;
    int tiles = a.tilesX * a.tilesY;
    a.tilesFrameMagic = div_magic((unsigned int)tiles);
    a.tilesXMagic = div_magic((unsigned int)a.tilesX);
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
        a.parkedMax = PARKED_MAX;
        const Tuning &tune = tuning(); // (pt_tuning.hpp: A/B knobs, set through pt_debug_set only)
        if (tune.parkedMax >= 0) a.parkedMax = tune.parkedMax > PARKED_MAX ? PARKED_MAX : tune.parkedMax;
        const int parkedMaxPlain = a.parkedMax;
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
        // hold, so the cycle cannot form any more (tools/handover_stress --multisample --tune batch_pass_min_tiles=0 sends every
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
        // Large scenes: the generic bounce walks the sphere grid (ray_trace_t<GRID>); the grid rides in LDS next to the scene
        const bool noGrid = tune.noSphereGrid != 0; // A/B runs
#ifdef PT_PROFILE
        const bool perWaveTimeline = false; // (the buffer only receives the section counters: every kernel writes them, tools/profile_sections.py)
#else
        const bool perWaveTimeline = a.timeline != nullptr; // tools/timeline.py: the TIMELINE instantiation (materials in LDS, no grid)
#endif
        const bool useGrid = a.grid != nullptr && a.gridBytes > 0 && !noGrid && !perWaveTimeline;
        a.gridLdsBytes = useGrid ? (a.gridBytes + 15) & ~15 : 0;
        if (useGrid && spp1) a.drainCompaction = 0; // (the tile-pass grid kernel exists without drain compaction only)
        lds += (size_t)a.gridLdsBytes;
        // CARRY (the pixel travels with its path, see PathEntryCarry): full-size images of scenes whose materials stay in LDS — the 12 more
        // bytes per ring entry and the 3 KB of lane slots are what the parked resolves take elsewhere, and only a GPU that owns few
        // tiles per frame needs those (+0.3 % at full 1080p); with the sphere grid, or on a small share, the plain kernel stays
        // Sphere-grid scenes (round 5, knob grid_carry): the grid kernel carries the pixel too, at FIVE workgroups per CU (its rings and lane
        // slots need 4.9 KB more than six leave room for; 96 VGPRs instead of 80)
        const bool gridCarry = useGrid && spp1 && tune.gridCarry != 0 && tiles >= 12000 && tune.carryLast != 0 && a.drainCompaction == 0 && !perWaveTimeline;
        if (gridCarry && blocksPerCU > 5 && tune.gridCarry == 1) { // (grid_carry = 2: at six workgroups per CU — the 60-byte carried record of round 6 leaves the room)
            blocksPerCU = 5;
            nwg = a.numCUs * blocksPerCU;
            if (nwg > numChunks) nwg = numChunks;
            if (nwg < 1) nwg = 1;
        }
        bool carry = spp1 && (!useGrid || gridCarry) && !perWaveTimeline && tiles >= 12000 && tune.carryLast != 0 && a.drainCompaction == 0;
        auto queue_bytes = [&](bool c) -> size_t {
            if (useBatchPass) return (size_t)waves * (64 * sizeof(PathEntryM) + (size_t)park * sizeof(ContEntry));
            return (spp1 ? frame_weight_bytes(a.batchFrames) : 0) +
                   (size_t)waves * 64 * (spp1 ? (c ? sizeof(PathEntryCarry) : sizeof(PathEntry)) + lane_last_bytes(c) : sizeof(RingEntry)) +
                   (a.drainCompaction != 0 ? (size_t)pool_slots(waves) * sizeof(PathState) : 0) // no pool without drain compaction
                   + (spp1 && a.tagged && a.drainCompaction == 0 ? (size_t)waves * a.parkedMax * sizeof(ParkedResolve) : 0); // parked resolves of tagged launches
        };
        // materials leave LDS when they would cost a resident workgroup (160 KB per CU; 64 B of static LDS per workgroup)
        const size_t ldsPerCU = 160 * 1024, fixedLds = wantFeed ? 256 : 64; // (static LDS of the persistent kernels: queue, drain control; frame-fed: + feed queue, per-wavefront counters)
        const bool forceLean = tune.forceLeanLds != 0; // A/B runs: materials always from the UBO copy
        size_t queues = 0, ldsTotal = 0;
        a.parkedMax = carry ? 0 : parkedMaxPlain;
        for (;;) {
            queues = queue_bytes(carry);
            ldsTotal = lds + queues;
            const size_t ldsLean = scene_lds_bytes(a.numSpheres, a.numCuboids, a.envFormat, false, a.gridLdsBytes) + queues;
            // The hardware hands out LDS in granules of 1,280 bytes (160 KB / 128) — measured on the 256-sphere scene: 26,880 bytes per
            // workgroup (21 granules) run six workgroups per CU, 27,136 run five, although 6 x 27,136 < 160 KB and
            // hipOccupancyMaxActiveBlocksPerMultiprocessor answers 6 for both.
            auto fit = [&](size_t bytes) { return ldsPerCU / ((bytes + fixedLds + 1279) / 1280 * 1280); };
            size_t wgFull = fit(ldsTotal), wgLean = fit(ldsLean);
            if (wgFull > (size_t)blocksPerCU) wgFull = (size_t)blocksPerCU;
            if (wgLean > (size_t)blocksPerCU) wgLean = (size_t)blocksPerCU;
            const bool lean = wgLean > wgFull || forceLean || useGrid; // (the grid kernel is only instantiated for materials in device memory)
            if (carry && gridCarry && wgLean >= (size_t)blocksPerCU) { // (the carrying grid kernel: materials in device memory like every grid kernel)
                a.materialsInLds = 0;
                ldsTotal = ldsLean;
                break;
            }
            if (carry && (lean || wgFull < (size_t)blocksPerCU)) { // (the carrying kernel exists with materials in LDS only, and must not cost a workgroup)
                carry = false;
                a.parkedMax = parkedMaxPlain;
                continue;
            }
            // the parked-resolve lists give way before a resident workgroup does (a list of 16 is plenty on a full-size image; the 256-sphere
            // scene with its grid sits 100 bytes over six workgroups per CU with lists of 64)
            if (!carry && spp1 && a.tagged && a.drainCompaction == 0 && tune.parkedMax < 0 && a.parkedMax > 16 &&
                (wgFull > wgLean ? wgFull : wgLean) < (size_t)blocksPerCU) {
                a.parkedMax -= 8;
                continue;
            }
            a.materialsInLds = lean ? 0 : 1;
            if (lean) ldsTotal = ldsLean;
            break;
        }
        a.sceneLdsBytes = (int)scene_lds_bytes(a.numSpheres, a.numCuboids, a.envFormat, a.materialsInLds != 0, a.gridLdsBytes);
        if (tuning().logLaunch > 0) {
            tuning().logLaunch--;
            std::fprintf(stderr, "mi355pt launch: spp1 %d grid %d carry %d matLds %d batchPass %d frames %d tiles %d wg/CU %d | LDS scene %d + queues %zu = %zu (+%zu static) -> %zu per CU fit, parkedMax %d\n",
                         (int)spp1, (int)useGrid, (int)carry, a.materialsInLds, (int)useBatchPass, a.batchFrames, tiles, blocksPerCU, a.sceneLdsBytes, queues, ldsTotal, fixedLds,
                         ldsPerCU / ((ldsTotal + fixedLds + 1279) / 1280 * 1280), a.parkedMax);
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
        // frame-fed launch: instantiated for the tile-pass kernels with materials in LDS (the default scene's kernels), five workgroups per CU
        const bool feed = wantFeed && spp1 && matLds && !useGrid && !perWaveTimeline && a.tagged && a.drainCompaction == 0 && blocksPerCU <= 6 &&
                          a.batchFrames == kFeedCapacity && a.feedHost && a.feedBcast && a.feedDone;
        if (wantFeed && !feed) return hipErrorNotSupported; // (the caller launches the classic way instead; nothing was enqueued)
        if (feed && carry) hipLaunchKernelGGL((pt_integrate_persistent_kernel<4, 6, false, true, true, false, true, false, true>), dim3(nwg), dim3(256), ldsTotal, stream, a);
        else if (feed) hipLaunchKernelGGL((pt_integrate_persistent_kernel<4, 6, false, true, true, false, false, false, true>), dim3(nwg), dim3(256), ldsTotal, stream, a);
        else
        if (perWaveTimeline && spp1 && matLds) PT_LAUNCH_PERSISTENT(true, true, true); // per-wavefront timestamps (tools/timeline.py)
        else if (spp1 && matLds && carry) hipLaunchKernelGGL((pt_integrate_persistent_kernel<4, PT_SPP1_WAVES, false, true, true, false, true, false>), dim3(nwg), dim3(256), ldsTotal, stream, a);
        else if (spp1 && matLds && a.drainCompaction == 0) hipLaunchKernelGGL((pt_integrate_persistent_kernel<4, PT_SPP1_WAVES, false, true, true, false, false, false>), dim3(nwg), dim3(256), ldsTotal, stream, a);
        else if (spp1 && matLds) PT_LAUNCH_PERSISTENT(false, true, true);
        else if (spp1 && useGrid && carry && blocksPerCU > 5) hipLaunchKernelGGL((pt_integrate_persistent_kernel<4, PT_GRID_MIN_WAVES, false, true, false, true, true, false>), dim3(nwg), dim3(256), ldsTotal, stream, a);
        else if (spp1 && useGrid && carry) hipLaunchKernelGGL((pt_integrate_persistent_kernel<4, 5, false, true, false, true, true, false>), dim3(nwg), dim3(256), ldsTotal, stream, a);
        else if (spp1 && useGrid) hipLaunchKernelGGL((pt_integrate_persistent_kernel<4, PT_GRID_MIN_WAVES, false, true, false, true, false, false>), dim3(nwg), dim3(256), ldsTotal, stream, a);
        else if (spp1) PT_LAUNCH_PERSISTENT(false, true, false);
        else if (useBatchPass) (void)launch_multisample(a, nwg, ldsTotal, stream, matLds, useGrid);
        else if (matLds) PT_LAUNCH_PERSISTENT(false, false, true);
        else if (useGrid) hipLaunchKernelGGL((pt_integrate_persistent_kernel<4, 5, false, false, false, true>), dim3(nwg), dim3(256), ldsTotal, stream, a);
        else PT_LAUNCH_PERSISTENT(false, false, false);
#undef PT_LAUNCH_PERSISTENT
        // every workgroup draws tickets until its first failing one: (numChunks - nwg) successful + nwg failing
        // (a pipelined batch draws every chunk dynamically: numChunks successful + nwg failing)
        *ticketsConsumed = a.tagged ? (unsigned int)(numChunks + nwg) : (unsigned int)(numChunks > nwg ? numChunks : nwg);
        if (feed) { // (the successful tickets depend on how many frames the host publishes: added when it closes the launch)
            *ticketsConsumed = (unsigned int)nwg;
            *fed = true;
        }
        if (workgroups) *workgroups = nwg;
    } else {
        int poolTiles = pool_tiles_for_variant(a.variant);
        int pools = (tiles + poolTiles - 1) / poolTiles;
        int nwg = (pools + 3) / 4;
        hipLaunchKernelGGL(pt_integrate_pool_kernel, dim3(nwg), dim3(256), lds, stream, a, poolTiles);
    }
    return hipGetLastError();
}

} // namespace pt
