// pt_kernel_common.hpp — what the integrator kernels share (included by pt_integrate_persistent.hip and pt_integrate_multisample.hip
// only): scene staging into LDS, the frame-pipelining hand-over primitives (alpha tags, device-coherent 16-byte pixel accesses), the
// path / continuation record layouts and the two-level tile queue.
#pragma once
#include <type_traits>

#include "pt_debug_hooks.hpp"
#include "pt_device.hpp"
#include "pt_kernels.hpp"
#include "pt_math.hpp"

namespace pt {

extern __shared__ float4 g_lds[];

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

// ---- frame pipelining (persistent spp = 1 kernels).  One launch can render a BATCH of consecutive frames: its tile queue
// runs over (frame, tile) pairs, frame-major, so wavefronts only drain once per batch instead of once per frame (the
// drain tail is ~75 us of a ~215 us frame at 1080p).  The only dependency between frames is per pixel: the running
// mean of frame f+1 needs the pixel's value after frame f (compute.glsl:126-129).  It is carried IN the pixel: inside a
// batch, frame j of the batch stores alpha = FRAME_TAG + j instead of 1 (the last frame of the batch stores the 1 the
// reference stores), and the resolve of frame j only proceeds when it reads the tag of frame j-1.  Pixels are written
// with ONE 16-byte device-scope (sc1) store and read with ONE 16-byte sc1 load — single-copy atomic and coherent
// across the 8 XCD L2s — so colour and tag always belong together.  A resolve that finds its predecessor missing is
// simply retried in the wavefront's next iteration (never a spin loop: the predecessor may live in another lane of
// the same wavefront).  The wait is bounded by WALL CLOCK (FrameArgs::waitBudget), and a result whose wait runs out is
// never folded onto a stale pixel: its launch is abandoned and the host re-renders what is missing (see "hand-over bound" below).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float FRAME_TAG = 2.0f;
// the tag of (absolute) frame f: distinct for any two frames that can be in flight together, exact in binary32.  Launches
// may CHAIN: the first frame of a tagged launch waits for the tag of the previous launch's last frame (FrameArgs::chainTag), so
// two launches on different streams overlap like the frames inside one launch do (the second fills the wavefront slots the
// first one's drain frees) — the host restores alpha = 1 before anything can observe the image (pt_set_alpha_kernel).
PT_DEV float frame_tag(int absFrame) { return FRAME_TAG + (float)(absFrame & kFrameTagMask); }

// ---- hand-over bound (round 5): wall clock, and giving up means ABANDONING the launch — never a fold onto a stale pixel.
// Time is the constant-rate counter (s_memrealtime, 100 MHz) >> 10: one unit = 10.24 us, 32 bits wrap after 12 hours (differences are
// taken modulo 2^32).  A lane (or parked list) notes when it started to wait; once per FrameArgs::waitCheckInterval a waiting wavefront
// looks at the handle's abandon word and at its own waits: one of them older than FrameArgs::waitBudget -> the wavefront abandons the
// launch (atomicMin of the launch's sequence number into the abandon word, host-visible flag raised); a launch whose sequence number is
// >= the abandon word stops drawing tickets (queue_pop_tile) and its wavefronts DROP the results that still wait: their pixels keep
// the tag of the last frame that was folded, which is all the host's repair pass (pt_repair_kernel) needs to re-render exactly the
// missing (pixel, frame) pairs behind the launch, in order.  Results that do not have to wait are folded as usual, abandoned or not.
PT_DEV unsigned int wait_clock() { return (unsigned int)((unsigned long long)wall_clock64() >> 10); }
constexpr unsigned int ABANDON_NONE = 0xffffffffu;
// has this launch (or an earlier one of its handle whose frames it builds on) been abandoned?  Device-scope load, wave-uniform.
PT_DEV bool launch_abandoned()
{
    ColdArgs ca = cold_args();
    if (ca->abandonWord == nullptr) return false;
    const unsigned int w = __hip_atomic_load((const unsigned int *)ca->abandonWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane((int)(w <= ca->launchSeq)) != 0;
}
// give the launch up (any lane may call; one atomic per wavefront is enough, more are harmless)
// (reason: kAbandonContended = a hand-over ran out of its budget; kAbandonIdle = a frame-fed launch waited too long for its next frame)
PT_DEV void abandon_launch(unsigned int reason = kAbandonContended)
{
    ColdArgs ca = cold_args();
    if (ca->abandonWord == nullptr) return;
    if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(true))) {
        atomicMin((unsigned int *)ca->abandonWord, ca->launchSeq);
        __hip_atomic_fetch_or((unsigned int *)ca->errorWord, reason, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// Wave-uniform state of the bound.  tick() is called once per iteration in which something of the wavefront waits: waitMask = the lanes
// whose result waits in the lane (timed per WAVEFRONT: "the same lanes have been waiting since" — a set of waiting lanes that changes is
// progress; a lane that can never be served belongs to an abandoned chain, which the periodic look at the abandon word finds);
// listSince / listStuck = a parked list (0 / false when it is empty or made progress).  Returns true when waiting results must be DROPPED.
// Only every 64th call reads the clock (s_memrealtime is a scalar memory instruction: the wavefront would wait for it in every
// iteration in which one lane waits — measured: +35 % in the resolve section of the default kernel); the calls between cost a few scalar
// instructions.  An iteration takes 1 us (a wavefront that only waits) to 20 us: the clock is looked at every 0.1 - 1 ms.
struct HandoverBound {
    unsigned int nextCheck; // wait_clock() value from which the abandon word is looked at again
    unsigned int laneSince; // wait_clock() | 1 of the first clock reading since which the set of waiting lanes looked the same (0: nobody waited)
    unsigned int lastLanes; // ... that set at the last clock reading, folded to 32 bits
    unsigned int calls;     // tick() calls (only every 64th reads the clock)
    bool listMoved;         // the parked list made progress (or was empty) in some call since the last clock reading
    bool abandoned;         // latched
    PT_DEV void init() { nextCheck = wait_clock(); laneSince = 0u; lastLanes = 0u; calls = 0u; listMoved = false; abandoned = false; }
    PT_DEV bool tick(unsigned long long waitMask, unsigned int &listSince, bool listStuck)
    {
        if (abandoned) return true;
        listMoved = listMoved || !listStuck;
        if ((++calls & 63u) != 0u) return false;
        // ---- every 64th call: read the clock, compare what waits now with what waited at the last reading
        const unsigned int now = wait_clock();
        const unsigned int lanes = (unsigned int)waitMask ^ (unsigned int)(waitMask >> 32) ^ (waitMask != 0ull ? 0x80000000u : 0u);
        if (lanes != lastLanes || waitMask == 0ull) laneSince = waitMask != 0ull ? (now | 1u) : 0u; // (changed: progress; the clock starts again)
        lastLanes = lanes;
        if (listMoved) listSince = 0u;
        else if (listSince == 0u) listSince = now | 1u;
        listMoved = false;
        if ((int)(now - nextCheck) >= 0) { // (taken once per waitCheckInterval)
            ColdArgs ca = cold_args();
            nextCheck = now + ca->waitCheckInterval;
            // (signed differences: a start time is stored with its lowest bit set — 0 means "not waiting" — and may lie one unit ahead)
            const int budget = (int)ca->waitBudget;
            if ((laneSince != 0u && (int)(now - laneSince) > budget) || (listSince != 0u && (int)(now - listSince) > budget)) {
                abandon_launch();
                abandoned = true;
            } else {
                abandoned = launch_abandoned();
            }
        }
        return abandoned;
    }
};
constexpr int MAX_BATCH_FRAMES = kMaxBatchFrames; // (one workgroup fills the weight table: <= its 256 threads; tags: pt_kernels.hpp kFrameTagMask)

PT_DEV float4 load_pixel_sc1(const float4 *p)
{
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
PT_DEV void store_pixel_sc1(float4 *p, float4 c)
{
    f32x4 v = {c.x, c.y, c.z, c.w};
    // (s_nop 1: a VMEM store of more than 64 bits must not be followed within 2 wait states by a VALU write of its data registers — gfx940+
    // "VMEM store data hazard".  The compiler's hazard recognizer covers its own stores but cannot see into an asm block: without the
    // nop the instruction it schedules next may overwrite v[0] while the store still reads it — seen in round 6 as snapshot pixels whose
    // red channel was 0.)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

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
};
// CARRY kernels (the default kernel on full-size images): the pixel's accumulation value, read by the TILE PASS with all 64 lanes
// (8 rows x 128 B: full lines) right after the first bounce, travels with the path: its resolve then needs no load (and no memory
// round trip) — nobody else writes the pixel between frame f-1's resolve and frame f's.  Only valid when the tile pass already saw
// the previous frame's tag (PATH_HAS_LAST).  Memory-side reads 84 -> 46 MB per 1080p frame (profiles/r04/xcd_affine_traffic.json).
// Round 6: the record stays at 60 bytes.  `rad` is +0 for every path that has not yet met the one emissive object (nearly all of them, when
// they enter the ring after their first bounce), so the last three floats hold EITHER the pixel's value (PATH_HAS_LAST: rad is +0, bit for
// bit) OR a non-zero rad (then the pixel does not travel: its resolve loads it, like any pixel whose previous frame was not in yet).  The
// 12 bytes per entry this saves (3 KB per workgroup) are what lets the sphere-grid kernel carry the pixel at six workgroups per CU.
struct PathEntryCarry {
    int pix;
    int bounce;
    uint32_t seed;
    float ro[3], rd[3], thr[3];
    float x[3]; // last (PATH_HAS_LAST) or rad
};
static_assert(sizeof(PathEntryCarry) == sizeof(PathEntry), "the carried record costs no LDS");
constexpr int PATH_HAS_LAST = 1 << 30;
// per lane: the pixel value read by the tile pass (LDS slot while the path is in a lane, see the kernel)
__host__ __device__ constexpr size_t lane_last_bytes(bool carry) { return carry ? 12 : 0; }

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
// LDS bytes of the per-launch frame table (spp = 1 persistent kernels), 16-byte aligned: per frame of the batch its running-mean weight
// 1 / (frame + 1) and the alpha it stores (its tag, or the reference's 1 for the launch's last frame) — everything a resolve needs to
// know about its frame in one 8-byte LDS read, no scalar arithmetic
__host__ __device__ constexpr size_t frame_weight_bytes(int batchFrames) { return (size_t)((batchFrames + 31) & ~31) * 8; }

struct BlockQueue {            // one per workgroup, in static LDS
    unsigned long long pair;   // (end << 32) | cursor : absolute tile indices of the current chunk
    unsigned int lock;         // refill lock
    unsigned int done;         // global queue exhausted
};


// relaxed workgroup-scope loads/stores of LDS control words (compile to ds_read / ds_write, never cached in registers)
PT_DEV unsigned int lds_load(const unsigned int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PT_DEV unsigned long long lds_load64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PT_DEV void lds_store(unsigned int *p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// an abandoned launch hands out no more work: a wavefront that notices closes its workgroup's queue (the tiles of the chunk it holds are
// still rendered; what the launch leaves undone is re-rendered by the host's repair pass, which also resets the ticket counters)
PT_DEV void stop_queue(BlockQueue *q)
{
    if ((threadIdx.x & 63) == 0) lds_store(&q->done, 1u);
}

// n / d and n % d for wave-uniform n < 2^31 with the host's magic = floor(2^32 / d) (d = 1: 2^32 - 1): the estimate mul_hi(n, magic) is the
// quotient or one less, so ONE correction step makes it exact — 6 scalar instructions where the compiler's division expands to 25.
__host__ __device__ inline unsigned int div_magic(unsigned int d) { return d <= 1u ? 0xffffffffu : (unsigned int)(0x100000000ull / d); }
PT_DEV void fast_divmod(int n, int d, unsigned int magic, int &q, int &r)
{
    unsigned int qq = __umulhi((unsigned int)n, magic);
    unsigned int rr = (unsigned int)n - qq * (unsigned int)d;
    if (rr >= (unsigned int)d) { qq++; rr -= (unsigned int)d; }
    q = (int)qq;
    r = (int)rr;
}

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
                // (an abandoned launch: the wavefronts that notice — every one that waits — close their workgroup's queue themselves,
                // stop_queue() below; the refill path stays as it was)
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

// ---- frame-fed launches (FrameArgs::feedHost): the workgroup's view of what the host has published, and the ticket it holds meanwhile
struct FeedQueue {             // one per workgroup, static LDS (only the FEED kernels touch it)
    unsigned long long stash;  // (end << 32) | first + 1 of a drawn ticket whose tiles are not all published yet; 0 = none
    unsigned int limit;        // (frame, tile) pairs below this index may be started: published frames x tiles per frame
    unsigned int word;         // the feed word as last seen: kFeedClosed | count << 16 | display slots (FrameArgs::feedHost)
    unsigned int pollAt;       // feed_refresh calls of this workgroup so far (every fourth one looks at the broadcast slot)
    unsigned int pad[3];
};
PT_DEV unsigned int feed_count(unsigned int word) { return (word >> 16) & 0x7fffu; }
constexpr int QUEUE_NOT_YET = -2; // queue_pop_tile_feed: work exists, but the host has not published its frame yet
constexpr unsigned int kStashRemainder = 0x80000000u; // FeedQueue::stash, low word: first tile + 1 | this flag (see queue_pop_tile_feed)

// What has the host published?  The launch's MONITOR wavefront is the only one that reads the host word (over PCIe) — it broadcasts every
// new word into kFeedBcastSlots device words, and a workgroup reads the slot blockIdx % kFeedBcastSlots.  (Round 6, first version: every
// workgroup polled ONE device word, at most one of them per microsecond the host word.  Whenever the launch ran out of published frames —
// every frame, for a host that shows each frame before it renders the next but one — 1,536 workgroups hammered that one address and the
// abandon word: loads of one address serialise at the memory side, the monitor's own loads queued behind them, and a frame took 1 ms.)
// Call from ONE wavefront of the workgroup (the refill lock's holder); wave-uniform.
PT_DEV void feed_refresh(FeedQueue *fq)
{
    ColdArgs ca = cold_args();
    // (rate limit by COUNTING the workgroup's calls, not by the clock: s_memrealtime from thousands of waiting wavefronts is itself a
    // hot spot — every fourth call looks, i.e. about every 2 - 3 us while the workgroup's four wavefronts retry)
    const unsigned int calls = (unsigned int)__builtin_amdgcn_readfirstlane((int)lds_load(&fq->pollAt));
    if ((threadIdx.x & 63) == 0) lds_store(&fq->pollAt, calls + 1u);
    if ((calls & 3u) != 0u) return;
    unsigned int w = __hip_atomic_load(ca->feedBcast + (blockIdx.x % kFeedBcastSlots) * kFeedBcastStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w = (unsigned int)__builtin_amdgcn_readfirstlane((int)w);
    if ((threadIdx.x & 63) == 0) {
        lds_store(&fq->limit, feed_count(w) * (unsigned int)(ca->tilesX * ca->tilesY));
        lds_store(&fq->word, w);
    }
}

// queue_pop_tile for a frame-fed launch: a ticket's tiles are handed out only as far as their frames are published; the rest of the ticket
// waits in the workgroup's stash.  -> tile index, -1 (all handed out: closed, or capacity used up), QUEUE_NOT_YET.  Wave-uniform.
// Every workgroup still draws exactly one failing ticket, and every ticket below ceil(final tiles / chunk) succeeds: the host's ticket
// accounting (queueBase) only needs the final frame count.
PT_DEV int queue_pop_tile_feed(BlockQueue *q, FeedQueue *fq)
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
            bool notYet = false;
            unsigned long long cur = lds_load64(&q->pair);
            unsigned int isDone = lds_load(&q->done);
            if (!isDone && (unsigned int)cur >= (unsigned int)(cur >> 32)) {
                ColdArgs ca = cold_args();
                const unsigned int capTiles = (unsigned int)(ca->tilesX * ca->tilesY * ca->batchFrames), chunk = (unsigned int)ca->queueChunk;
                unsigned long long st = lds_load64(&fq->stash);
                unsigned int first, last;
                bool remainder = false; // the stash is the REST of a ticket part of which has been handed out (kStashRemainder)
                if (__builtin_amdgcn_readfirstlane((int)(st != 0ull))) {
                    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)st);
                    remainder = (lo & kStashRemainder) != 0u;
                    first = (lo & ~kStashRemainder) - 1u;
                    last = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(st >> 32));
                } else {
                    unsigned int ticket = 0;
                    CHAOS(2);
                    if (leader) ticket = atomicAdd(ca->queue, 1u) - ca->queueBase;
                    ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
                    const unsigned long long f64 = (unsigned long long)ticket * chunk;
                    first = f64 > 0xfffffff0ull ? 0xfffffff0u : (unsigned int)f64;
                    last = first + chunk < capTiles ? first + chunk : capTiles;
                }
                unsigned long long newStash = 0ull;
                if (first >= capTiles) { // beyond the launch's capacity: this workgroup's failing ticket
                    if (leader) lds_store(&q->done, 1u);
                } else {
                    unsigned int limit = (unsigned int)__builtin_amdgcn_readfirstlane((int)lds_load(&fq->limit));
                    unsigned int closed = (unsigned int)__builtin_amdgcn_readfirstlane((int)lds_load(&fq->word)) >> 31;
                    if (last > limit && !closed) {
                        feed_refresh(fq);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        limit = (unsigned int)__builtin_amdgcn_readfirstlane((int)lds_load(&fq->limit));
                        closed = (unsigned int)__builtin_amdgcn_readfirstlane((int)lds_load(&fq->word)) >> 31;
                    }
                    if (first < limit) { // (part of) the ticket is published
                        const unsigned int upto = last < limit ? last : limit;
                        if (leader) atomicExch(&q->pair, ((unsigned long long)upto << 32) | (unsigned long long)first);
                        if (upto < last && !closed) newStash = ((unsigned long long)last << 32) | (unsigned long long)((upto + 1u) | kStashRemainder);
                    } else if (closed && !remainder) { // nothing of it will ever be published: the failing ticket
                        if (leader) lds_store(&q->done, 1u);
                    } else if (closed) {
                        // the unpublished rest of a ticket whose first tiles WERE handed out: that ticket counts among the successful ones in the
                        // host's accounting (every ticket below ceil(final tiles / chunk)), so this workgroup still has its failing ticket to
                        // draw — the stash is dropped and the next round draws it.  (Without this the device counter ended one short of the
                        // host's base per such workgroup whenever a launch closed on a frame count whose tiles do not fill whole chunks —
                        // 1440 x 900 has 20,340 tiles, chunk 8 — and a launch of a tiny image, with fewer workgroups than the deficit, then drew
                        // only failing tickets and rendered nothing: found by tools/handover_stress --tune feed_min_tiles=0.)
                    } else {
                        newStash = ((unsigned long long)last << 32) | (unsigned long long)((first + 1u) | (remainder ? kStashRemainder : 0u));
                        notYet = true;
                    }
                }
                if (leader) __hip_atomic_store(&fq->stash, newStash, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (leader) atomicExch(&q->lock, 0u);
            if (notYet) return QUEUE_NOT_YET;
        } else {
            __builtin_amdgcn_s_sleep(2);
        }
    }
}

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


// the spp > 1 batch-pass kernel (pt_integrate_multisample.hip), launched by launch_integrate (pt_integrate_persistent.hip)
hipError_t launch_multisample(const FrameArgs &a, int workgroups, size_t ldsBytes, hipStream_t stream, bool materialsInLds, bool sphereGrid);

} // namespace pt
