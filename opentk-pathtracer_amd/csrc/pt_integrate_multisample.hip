// pt_integrate_multisample.hip — the spp > 1 integrator of gfx950: the BATCH PASS kernel (every sample's first bounce coherent and
// culled).  Same arithmetic contract, same per-pixel sample order as every other kernel: bit-identical images.
// Build flags (see __graft_entry__.build): -O3 -ffp-contract=off -fno-fast-math --offload-arch=gfx950
#include "pt_kernel_common.hpp"

namespace pt {

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
    // IRR_LDS (experiment of round 6, off): the lane's running irradiance sum and its pixel in LDS planes instead of registers, for the
    // instantiations that read materials from device memory (they spilled 6 / 24 VGPRs in round 5).  It removes the last spills, but its
    // 4 KB per workgroup cost the 256-sphere scene its fifth workgroup per CU: 11.85 -> 10.8 Gsamples/s at 4 spp.  What stayed: the lane's
    // three counters packed into one register and the batch pass's dummy walk variable made a local — 6 / 24 spills -> 2 / 1, no LDS.
#ifndef PT_MS_IRR_LDS
#define PT_MS_IRR_LDS 0
#endif
#ifndef PT_MS_EARLY
#define PT_MS_EARLY 40
#endif
    constexpr bool IRR_LDS = !MATLDS && PT_MS_IRR_LDS != 0;
    [[maybe_unused]] float *const laneIrr = (float *)(ringBase + NWAVES * 64 * (int)sizeof(PathEntryM) + NWAVES * parkCapacity * (int)sizeof(ContEntry)) + wave * 4 * 64 + lane;
    v3 irrReg = V(0, 0, 0);
    auto get_irr = [&]() -> v3 {
        if constexpr (IRR_LDS) return V(laneIrr[0], laneIrr[64], laneIrr[128]);
        else return irrReg;
    };
    auto set_irr = [&](v3 v) -> void {
        if constexpr (IRR_LDS) { laneIrr[0] = v.x; laneIrr[64] = v.y; laneIrr[128] = v.z; }
        else irrReg = v;
    };
    // image coordinates of accumulation pixel `p` of this launch: x | global row << 16
    auto pixel_xy = [&](int p) -> int { // p = x | local row << 16 (no division anywhere)
        ColdArgs ca = cold_args();
        const int ly = p >> 16, x = p & 0xffff;
        return x | (global_row_v(ca->bandRows, ca->bandWorld, ca->bandRank, ca->localRow0, ca->y0, ly) << 16);
    };

    int avail = 0, parked = 0, qhead = 0; // wave-uniform: ring entries [0, avail); `parked` continuations from slot qhead on (FIFO, circular)
    // the hand-over's wall-clock bound (pt_kernel_common.hpp), kept per WAVEFRONT: waiting records move between lanes, ring and queue, so
    // what is timed is "this wavefront has done nothing but wait since" (0: some lane traced in the last iteration)
    HandoverBound bound;
    bound.init();
    unsigned int stallSince = 0u;
    auto qslot = [&](int i) -> int { // slot of the i-th parked continuation
        int sl = qhead + i;
        return sl >= parkCapacity ? sl - parkCapacity : sl;
    };
    bool exhausted = false;
    // the lane's counters in ONE register, in PathEntryM's layout: bounces done | sample << 12 | frame of the batch << 24 (three registers
    // live across the bounce were two too many for the instantiations that read materials from device memory)
    // IRR_LDS kernels: the lane's PIXEL lives in a fourth LDS plane too (it is only read where a pixel is parked, resolved or given a new
    // primary ray), and "this lane holds no path" is bit 31 of the counters
    int pixReg = -1, counters = IRR_LDS ? -1 : 0;
    auto lane_idle = [&]() -> bool {
        if constexpr (IRR_LDS) return counters < 0;
        else return pixReg < 0;
    };
    auto set_idle = [&]() -> void {
        if constexpr (IRR_LDS) counters = -1;
        else pixReg = -1;
    };
    auto get_pix = [&]() -> int {
        if constexpr (IRR_LDS) return __builtin_bit_cast(int, laneIrr[192]);
        else return pixReg;
    };
    auto set_pix = [&](int p) -> void { // (followed by an assignment of the counters, which clears the idle bit)
        if constexpr (IRR_LDS) laneIrr[192] = __builtin_bit_cast(float, p);
        else pixReg = p;
    };
    auto c_bounce = [&]() -> int { return counters & 0xfff; };
    auto c_sample = [&]() -> int { return (counters >> 12) & 0xfff; };
    auto c_fj = [&]() -> int { return (counters >> 24) & 0x7f; };
    bool needRay = false, pending = false;
    float walkFrom = -1.0f; // (WALK SLICES, as in the persistent kernel)
#ifdef PT_PROFILE
    unsigned long long prof_dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // (this kernel has no section counters of its own)
#endif
    uint32_t seed = 0;
    v3 ro = V(0, 0, 0), rd = V(0, 0, 1), throughput = V(1, 1, 1), rad = V(0, 0, 0);

    auto fold = [&](float4 last, v3 rirr, int rfj) -> float4 { // compute.glsl:125-129
        ColdArgs ca = cold_args();
        rirr = v_scale(rirr, f_div_ieee(1.0f, (float)ca->spp));
        const float w = f_div_ieee(1.0f, (float)(ca->frame + rfj + 1));
        const float alpha = (rfj == ca->batchFrames - 1 && !ca->keepTags) ? 1.0f : frame_tag(ca->frame + rfj);
        return make_float4(f_mix(last.x, rirr.x, w), f_mix(last.y, rirr.y, w), f_mix(last.z, rirr.z, w), alpha);
    };
    auto try_resolve = [&](int rpix, int rfj, v3 rirr) -> bool {
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
        if (expected != 0.0f && last.w != expected) return false;
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
    // wall-clock bound below abandons the launch instead of hanging (the host's repair pass then renders what is missing).  (Swapping waiting results with queued paths
    // was tried first: it needs the path state to be assignable at a second place, which costs 20 spilled VGPRs.)
    auto rescue = [&]() -> void {
        const int room = parkCapacity - parked;
        if (room > 0) {
            if (lane < room) { // (every lane waits, so lane l parks into the l-th free slot)
                ContEntry e;
                e.pix = get_pix(); e.seed = seed; e.sfj = c_sample() | (c_fj() << 16); // sample == spp marks "last sample done, waiting"
                const v3 irr = get_irr();
                e.irr[0] = irr.x; e.irr[1] = irr.y; e.irr[2] = irr.z;
                cq[qslot(parked + lane)] = e;
                set_idle();
                pending = false;
            }
            parked += room < 64 ? room : 64;
        }
        __builtin_amdgcn_wave_barrier();
    };

    for (;;) {
        bool idle = lane_idle();
        unsigned long long m = __ballot(idle);
        // ---- forced batch pass (the progress guarantee of the pipelining).  When EVERY lane holds a finished pixel that waits for its
        // previous frame, nothing pops the ring or runs a batch pass any more — and the work those lanes wait for may be parked in this
        // wavefront's own queue.  A batch pass needs no idle lane (it computes in its own registers and only needs ring slots), so
        // such a wavefront runs one over the oldest parked records anyway, appending the survivors to the ring.  Every record keeps
        // exactly one place (lane, ring slot or queue slot), a pass that finds real work advances it by a bounce or a sample and
        // leaves at least as much room in the queue as it put paths into the ring — so rescue() below can then free lanes for them.
        // Waiting records it meets are retried and rotate to the back of the FIFO.  With the tickets handed out frame-major, all
        // work of the oldest unfinished frame is therefore always executed by whichever wavefront holds it: no cycle of waits.
        bool forcePass = parked > 0 && avail < 64 && __ballot(!(!lane_idle() && pending && !needRay)) == 0ull;
        // EARLY PASS (sphere-grid scenes only, round 5 measurement / round 6 merge): a batch pass as soon as the queue is within 40 records
        // of full and the ring has room for it, so that the queue never overflows — an overflowing continuation takes its next sample's
        // first bounce in the lane, unculled, against all 256 spheres: +3.5 % at C3 4 spp; the 48-sphere scene's unculled bounce is cheaper
        // than a thin pass (-2 % there), hence tied to GRID (profiles/r05/multisample_early_pass.log)
        if constexpr (GRID && PT_MS_EARLY > 0) forcePass = forcePass || (parked + PT_MS_EARLY >= parkCapacity && avail <= 64 - PT_MS_EARLY && parked > 0);
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
                    if (unsigned int *flags = cold_args()->tileFlags) // (FrameArgs::tileFlags: where the launch's first frame ran)
                        if (tile < numTilesFrame && lane == 0) flags[tile] = cold_args()->launchSeq;
                }
                ColdArgs ca = cold_args();
                ColdFloats cam = (ColdFloats)ca;
                const int width = ca->width;
                const float invW = ca->invW, invH = ca->invH;
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
                    int tx, ty;
                    fast_divmod(tile, numTilesFrame, ca->tilesFrameMagic, tfj, tile);
                    fast_divmod(tile, tilesX, ca->tilesXMagic, ty, tx);
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
                unsigned long long masks[5];
                if (!fromQueue && ca->tileMasks != nullptr) { // a fresh tile (sample 0): the tile's cached masks, as in the spp = 1 tile pass
                    const __attribute__((address_space(4))) unsigned long long *tm =
                        (const __attribute__((address_space(4))) unsigned long long *)ca->tileMasks + (size_t)tile * kTileMaskWords;
                    masks[0] = tm[0]; masks[1] = tm[1]; masks[2] = tm[2]; masks[3] = tm[3]; masks[4] = tm[4];
                } else if (fromQueue && ca->tileMasks != nullptr) {
                    // Continuations of a few tiles (round 5): a tile's cached masks hold every primary ray the tile can EVER cast — any sample's
                    // jitter and lens point — so the union of the masks of the tiles present in this bundle bounds the bundle: spheres AND
                    // cuboids culled (the cone fitted to the rays themselves culls no cuboids and is wide for a mixed bundle), for a
                    // handful of scalar loads per distinct tile.  More than 8 distinct tiles: the cone, as before.
                    masks[0] = masks[1] = masks[2] = masks[3] = masks[4] = 0ull;
                    const int myTile = ((tpix >> 16) >> 3) * ca->tilesX + ((tpix & 0xffff) >> 3);
                    unsigned long long rem = __ballot(valid);
                    int distinct = 0;
                    while (rem != 0ull && distinct < 8) {
                        const int t = __builtin_amdgcn_readlane(myTile, (int)__builtin_ctzll(rem));
                        const __attribute__((address_space(4))) unsigned long long *tm =
                            (const __attribute__((address_space(4))) unsigned long long *)ca->tileMasks + (size_t)t * kTileMaskWords;
                        masks[0] |= tm[0]; masks[1] |= tm[1]; masks[2] |= tm[2]; masks[3] |= tm[3]; masks[4] |= tm[4];
                        rem &= ~__ballot(valid && myTile == t);
                        distinct++;
                    }
                    if (rem != 0ull) cull_spheres(sc, a.numSpheres, valid, to, td, masks);
                } else {
                    cull_spheres(sc, a.numSpheres, valid, to, td, masks); // (continuations of several tiles: bounded from the rays themselves)
                }
                bool tcont = false;
                float walkFresh = -1.0f; // (the batch pass never walks the grid: a local, not a register carried round the main loop)
                if (valid) {
                    if (0 < a.rayDepth) tcont = bounce_step_t<true, MATLDS>(sc, a.numSpheres, a.numCuboids, env, to, td, tthr, trad, tseed, masks, walkFresh PROF_DUMMY);
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
                if (twaiting) { // still waiting: back into the queue, as it was (an abandoned launch drops it: the host's repair pass renders it)
                    if (!try_resolve(tpix, tfj, tirr) && !bound.abandoned) tmore = true;
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
                if (tfin && tsample >= a.spp && !try_resolve(tpix, tfj, tirr) && !bound.abandoned) {
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
                set_pix(e.pix);
                counters = e.counters & 0x7fffffff;
                needRay = e.counters < 0;
                pending = false;
                walkFrom = -1.0f; // (a fresh path: no unfinished grid walk)
                seed = e.seed;
                ro = V(e.ro[0], e.ro[1], e.ro[2]);
                rd = V(e.rd[0], e.rd[1], e.rd[2]);
                throughput = V(e.thr[0], e.thr[1], e.thr[2]);
                rad = V(e.rad[0], e.rad[1], e.rad[2]);
                set_irr(V(e.irr[0], e.irr[1], e.irr[2]));
                if (!needRay && c_bounce() >= a.rayDepth && c_sample() >= a.spp) pending = true; // a resolve that had to wait
            }
            const int n = __builtin_popcountll(m);
            avail = n < avail ? avail - n : 0;
            idle = lane_idle();
            m = __ballot(idle);
        }
        bool active = !lane_idle();
        if (__ballot(active) == 0ull) {
            if (exhausted && avail == 0 && parked == 0) break;
            if (bound.tick(0ull, stallSince, true)) stop_queue(&queue); // (only waiting records left in the queue: they are retried — or, abandoned, dropped — by the batch passes above)
            if (parked > 0 && avail == 0) __builtin_amdgcn_s_sleep(8);
            continue;
        }
        if (__ballot(!(active && pending && !needRay)) == 0ull && (avail > 0 || parked > 0)) {
            rescue(); // every lane waits: see above
            // (round 4: the lanes rescue() has just emptied are NOT active any more.  They used to run one bounce of their dead path below —
            // harmless while a bounce left nothing behind in the lane, but a grid walk cut short (WALK SLICES) leaves walkFrom, and the
            // path the lane pops next would have resumed someone else's walk: found by tools/handover_stress --multisample)
            active = !lane_idle();
        }
        // (a wavefront that has done nothing but wait for FrameArgs::waitBudget abandons the launch: waiting records move between lanes,
        // ring and queue, so the bound is kept per wavefront, not per lane)
        // ... and a result that sits in a lane is timed like the spp = 1 kernel's: the same lanes waiting for the whole budget.  (Any waiting
        // lane also makes the wavefront look at the abandon word now and then: an abandoned launch stops drawing tickets.)
        const bool nothingTraced = __ballot(!lane_idle() && !pending) == 0ull;
        const unsigned long long laneWaits = __ballot(!lane_idle() && pending && !needRay);
        if (nothingTraced || stallSince != 0u || laneWaits != 0ull)
            if (bound.tick(laneWaits, stallSince, nothingTraced)) stop_queue(&queue);
        if (active && needRay) { // fallback (queue was full): the next sample's primary ray, generated in the lane
            const int pxy = pixel_xy(get_pix());
            primary_ray(a, pxy & 0xffff, pxy >> 16, seed, ro, rd);
            throughput = V(1.0f, 1.0f, 1.0f);
            rad = V(0.0f, 0.0f, 0.0f);
            counters &= ~0xfff; // bounce = 0
            needRay = false;
        }
        bool wantPark = false;
        const bool trace = active && !pending; // (one divergent region around the bounce, as in the persistent kernel)
        bool cont = false;
        if (trace && c_bounce() < a.rayDepth) cont = bounce_step_t<false, MATLDS, GRID, (MATLDS && !GRID)>(sc, a.numSpheres, a.numCuboids, env, ro, rd, throughput, rad, seed, nullptr, walkFrom PROF_DUMMY);
        if (trace) {
            const bool sliced = GRID && walkFrom >= 0.0f; // (the grid walk of this bounce continues in the next iteration: pt_device.hpp, WALK SLICES)
            if (!sliced) counters++; // bounce++ (< 4096 by the ABI's ray_depth limit: no carry into the sample field)
            if (!sliced && (!cont || c_bounce() >= a.rayDepth)) {
                set_irr(v_add(get_irr(), rad)); // compute.glsl:122
                counters += 1 << 12; // sample++
                if (c_sample() < a.spp) wantPark = true;
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
                e.pix = get_pix(); e.seed = seed; e.sfj = c_sample() | (c_fj() << 16);
                const v3 irr = get_irr();
                e.irr[0] = irr.x; e.irr[1] = irr.y; e.irr[2] = irr.z;
                cq[qslot(parked + rank)] = e;
                set_idle();
            } else if (wantPark) {
                needRay = true;
            }
            const int n = __builtin_popcountll(pm);
            parked += n < room ? n : room;
            __builtin_amdgcn_wave_barrier();
        }
        if (!lane_idle() && pending) {
            if (try_resolve(get_pix(), c_fj(), get_irr()) || bound.abandoned) { // (abandoned launch: a result that still has to wait is dropped, the host's repair pass renders it)
                set_idle();
                pending = false;
            }
        }
        { // nothing but waiting paths left in this wavefront: do not hammer the pixel
            const bool act = !lane_idle();
            if (__ballot(act && pending) != 0ull && __ballot(act && !pending) == 0ull && parked < CONT_BATCH_MIN && avail == 0)
                __builtin_amdgcn_s_sleep(8);
        }
    }
}

hipError_t launch_multisample(const FrameArgs &a, int workgroups, size_t ldsBytes, hipStream_t stream, bool materialsInLds, bool sphereGrid)
{
    if (materialsInLds) hipLaunchKernelGGL(pt_integrate_multisample_kernel<true>, dim3(workgroups), dim3(256), ldsBytes, stream, a);
    else if (sphereGrid) hipLaunchKernelGGL((pt_integrate_multisample_kernel<false, true>), dim3(workgroups), dim3(256), ldsBytes, stream, a);
    else hipLaunchKernelGGL(pt_integrate_multisample_kernel<false>, dim3(workgroups), dim3(256), ldsBytes, stream, a);
    return hipGetLastError();
}

} // namespace pt
