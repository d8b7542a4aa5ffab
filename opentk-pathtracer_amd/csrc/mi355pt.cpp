// mi355pt.cpp — implementation of the C ABI declared in include/mi355pt.h.
//
// Host-side state of one renderer (what the reference keeps in the C# class PathTracer,
// /root/reference/OpenTK-PathTracer/src/Render/PathTracer.cs:9-141, plus the two UBOs MainWindow owns,
// src/MainWindow.cs:195-201) and the HIP plumbing around the kernels of the pt_*.hip files.
// There is deliberately NO CPU fallback: without a HIP device every entry point fails with PT_E_NO_DEVICE.
#include "pt_renderer.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <cstring>
#include <new>

namespace {
thread_local std::string g_create_error;
} // namespace

namespace ptimpl {

#define FEED_LOG(...)                                                                                                                \
    do {                                                                                                                             \
        if (pt::tuning().feedLog) {                                                                                                  \
            std::fprintf(stderr, "[feed %9.1f us] ", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count() - 1e6 * (long long)(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count())); \
            std::fprintf(stderr, "%p ", (void *)h);                                                                                  \
            std::fprintf(stderr, __VA_ARGS__);                                                                                       \
            std::fprintf(stderr, "\n");                                                                                              \
        }                                                                                                                            \
    } while (0)

int fail(pt_handle h, int code, const std::string &msg)
{
    if (h) h->error = msg;
    else g_create_error = msg;
    return code;
}

int hip_fail(pt_handle h, hipError_t e, const char *what)
{
    return fail(h, e == hipErrorOutOfMemory ? PT_E_OUT_OF_MEMORY : PT_E_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}

int bind_device(pt_handle h)
{
    PT_HIP(h, hipSetDevice(h->device));
    return PT_OK;
}

} // namespace ptimpl

using ptimpl::bind_device;
using ptimpl::fail;
using ptimpl::flush_frames;
using ptimpl::hip_fail;
using ptimpl::join_stripes;

namespace {

// GL 4.5 section 8.24 sRGB decode, evaluated in double and rounded once (same table as the oracle's).
void make_srgb_lut(float *lut)
{
    for (int i = 0; i < 256; i++) {
        double cs = i / 255.0;
        double cl = cs <= 0.04045 ? cs / 12.92 : std::pow((cs + 0.055) / 1.055, 2.4);
        lut[i] = (float)cl;
    }
}

} // namespace

namespace ptimpl {
// Launch what pt_render deferred, then make the main stream wait for every stripe kernel still in flight (before
// anything that reads their output or overwrites their inputs).
hipEvent_t next_launch_event(pt_handle h)
{
    hipEvent_t &e = h->launchEvents[h->launchEventNext++ % pt_renderer::kLaunchEvents];
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
    return e;
}

// Hand-over repair: the launches since the last join, in launch order, behind the joined streams (call with the helper streams joined
// into h->stream).  On the device each pass is a single load unless a launch was abandoned (pt_repair_kernel).
static int enqueue_repairs(pt_handle h, bool flagWasDown)
{
    // flagWasDown (the caller has NOT just found and cleared the abandon flag): launches seen COMPLETE with the flag DOWN — read in that
    // order: an abandoning launch raises the flag before it ends — ran to their end and need no pass: a host that joins often (a blocking
    // present per frame) enqueues nothing at all on the usual path.  (After note_abandonment the flag is down because the HOST cleared it:
    // then every remembered launch gets its pass.)
    if (flagWasDown) {
        auto flag_down = [&]() -> bool { return !(h->hostErrWord && *(volatile unsigned int *)h->hostErrWord); };
        while (!h->unverified.empty() && hipEventQuery(h->unverified.front().done) == hipSuccess && flag_down()) {
            FEED_LOG("repairs: launch %u frames [%d,+%d) seen complete with the flag down: no pass", h->unverified.front().a.launchSeq, h->unverified.front().a.frame, h->unverified.front().a.batchFrames);
            h->unverified.pop_front();
        }
        (void)hipGetLastError(); // (hipErrorNotReady of the query is not an error)
    }
    // (nothing remembered and no abandonment noted: nothing to do.  An abandonment noted with nothing remembered — the launches it belongs to
    // had their passes enqueued by an earlier join, before they gave up — still gets the closing kernel: the device's abandon word and
    // ticket counters must never stay behind a flag the host has cleared)
    if (h->unverified.empty() && flagWasDown) return PT_OK;
    for (const pt_renderer::LaunchRecord &r : h->unverified) {
        FEED_LOG("repairs: pass for launch %u frames [%d,+%d) chainTag %g keepTags %d variant %d", r.a.launchSeq, r.a.frame, r.a.batchFrames, (double)r.a.chainTag, r.a.keepTags, r.a.variant);
        PT_HIP(h, pt::launch_repair(r.a, h->dRepairCtl, h->stream));
    }
    PT_HIP(h, pt::launch_repair_done(h->dAbandon, h->dQueue, h->stripeQueueBase[0], h->dQueue + kChainQueueWord, h->chainQueueBase,
                                     h->dRepairCtl, h->stream));
    h->unverified.clear();
    if (h->feedCountersStale && h->dFeedDone) {
        // an abandoned frame-fed launch counted fewer pixels than the host's running totals say: the counters start again from zero —
        // behind everything joined into h->stream, and behind the gates and tone maps still queued on the copy stream
        if (!h->feedResetEvent) PT_HIP(h, hipEventCreateWithFlags(&h->feedResetEvent, hipEventDisableTiming));
        PT_HIP(h, hipEventRecord(h->feedResetEvent, h->copyStream));
        PT_HIP(h, hipStreamWaitEvent(h->stream, h->feedResetEvent, 0));
        PT_HIP(h, hipMemsetAsync(h->dFeedDone, 0, (size_t)16 * pt::kFeedDoneStride * sizeof(unsigned long long), h->stream));
        std::memset(h->feedDoneBase, 0, sizeof h->feedDoneBase);
        h->feedCountersStale = false;
    }
    return PT_OK;
}

// The host found the abandon flag raised: from here on the device is treated as contended (launches of this handle no longer overlap for a
// while), and present slots tone-mapped from a snapshot since the last epoch are tone-mapped again when they are waited for.
static void note_abandonment(pt_handle h)
{
    const unsigned int why = *(volatile unsigned int *)h->hostErrWord;
    *(volatile unsigned int *)h->hostErrWord = 0;
    FEED_LOG("abandon flag %u seen and cleared (frame %d, %zu launches remembered, feed open %d published %d)", why, h->frame, h->unverified.size(), (int)h->feed.open, h->feed.published);
    h->abandonEpoch++;
    // a hand-over that ran out of its budget = a contended device: launches stop overlapping for a while.  A frame-fed launch that ended
    // itself because no frame came (reason "idle") says nothing about the device — only about the host's pace: a host whose fed launches
    // end idle after a frame or two is too slow to feed a resident launch, and gets the classic launches for a (growing) while
    if (why & pt::kAbandonContended) h->overlapHoldoff = 256;
    if (why & pt::kAbandonIdle) {
        h->statFeedIdle++;
        if (h->feed.published <= 2) {
            h->feedIdleStrikes++;
            if (h->feedIdleStrikes >= 2) h->feedHoldoff = h->feedIdleStrikes >= 10 ? 4096 : (16 << (h->feedIdleStrikes - 2));
        } else {
            h->feedIdleStrikes = 0;
        }
    }
    h->feedCountersStale = true; // (an abandoned fed launch counted fewer pixels than the host's running totals say)
    h->repairPendingAll = true; // (the flag is down again because the host cleared it: the launches remembered now all get their repair pass)
    h->repairCheckDue = true; // the next blocking call looks at what the repair passes found (settle_handover)
}

// The open frame-fed launch takes no more frames: the host word gets its final count (| kFeedClosed), and everything that depended on
// that count is settled — the launch's repair record, the ticket counter of its stream, the running totals of the per-frame counters.
unsigned int feed_word(const pt_renderer::FeedState &f, bool closed)
{
    return (closed ? pt::kFeedClosed : 0u) | ((unsigned int)f.published << 16) | (f.slots16 & 0xffffu);
}

void feed_close(pt_handle h)
{
    pt_renderer::FeedState &f = h->feed;
    if (!f.open) return;
    f.open = false;
    __atomic_store_n(&h->hostFeed[f.hostWord], feed_word(f, true), __ATOMIC_RELEASE);
    FEED_LOG("close word %d = %08x (first %d published %d pendingSlot %d)", f.hostWord, feed_word(f, true), f.firstFrame, f.published, f.pendingSlot);
    for (pt_renderer::LaunchRecord &r : h->unverified)
        if (r.a.launchSeq == f.seq) r.a.batchFrames = f.published;
    const unsigned int tickets = (unsigned int)(((long long)f.published * f.tilesFrame + f.queueChunk - 1) / f.queueChunk); // (+ one failing ticket per workgroup: added at the launch)
    (f.streamIdx == 1 ? h->chainQueueBase : h->stripeQueueBase[0]) += tickets;
    for (int j = 0; j < f.published; j++) h->feedDoneBase[f.streamIdx][j & 7] += f.pixelsPerFrame;
    if (f.published >= 8) h->feedIdleStrikes = 0; // (the host kept a launch fed: not a slow host)
    // fused display: the newest frame was presented, and no later frame of this launch will come to tone-map it — the next join does
    // (from the accumulation image, before anything else is rendered: whoever closes a launch joins, or launches through launch_frames)
    if (f.pendingSlot >= 0) {
        h->fusedOrphanSlot = f.pendingSlot;
        f.pendingSlot = -1;
    }
}

// A present into slot fusedOrphanSlot was left without a successor frame (see feed_close): with every launch joined into h->stream the
// accumulation image IS the frame that was shown — tone-map it the classic way.  Call with the streams joined.
static int resolve_fused_orphan(pt_handle h)
{
    if (h->fusedOrphanSlot < 0) return PT_OK;
    FEED_LOG("orphan present of slot %d: tone map from the image", h->fusedOrphanSlot);
    PresentSlot &s = h->slots[h->fusedOrphanSlot];
    h->fusedOrphanSlot = -1;
    if (!s.fusedWaiting) return PT_OK;
    void *const image = s.boundDev ? s.boundDev : s.dRgba8;
    const size_t pixels = (size_t)s.rows * s.width;
    PT_HIP(h, pt::launch_postprocess(h->accum(), image, pixels, h->stream));
    PT_HIP(h, hipEventRecord(s.toneMapped, h->stream));
    PT_HIP(h, hipStreamWaitEvent(h->copyStream, s.toneMapped, 0));
    if (!s.boundDev) PT_HIP(h, hipMemcpyAsync(s.host, s.dRgba8, pixels * 4, hipMemcpyDeviceToHost, h->copyStream));
    PT_HIP(h, hipEventRecord(s.copied, h->copyStream));
    s.fusedWaiting = false;
    s.fedPresent = false; // (completed by the events above, like any present tone-mapped behind a join — the join put the repair passes in front)
    s.snapSource = nullptr;
    return PT_OK;
}

int join_stripes(pt_handle h, bool repairNow)
{
    FEED_LOG("join (repairNow %d, frame %d, pending %d, tagsLive %d, %zu remembered, flag %u)", (int)repairNow, h->frame, h->pendingFrames, (int)h->tagsLive, h->unverified.size(),
             h->hostErrWord ? *(volatile unsigned int *)h->hostErrWord : 0u);
    feed_close(h);
    if (h->pendingFrames > 0) {
        // whatever follows a join is ordered by the streams again, so this launch need not leave its tags in the image: its
        // last frame stores the reference's alpha = 1 and the usual render-N-then-read sequence needs no alpha pass at all
        h->flushFinal = true;
        int rc = flush_frames(h);
        h->flushFinal = false;
        if (rc) return rc;
    }
    h->snapFrame = -1;   // whoever joins may change the image or the frame counter: a present snapshot written earlier is stale
    h->mainDirty = true; // whoever joins is about to put other work on the main stream: the next striped frame orders behind it
    h->chainBroken = true; // ... and the next tagged launch re-joins the two launch streams before it starts
    if (h->chainPending) {
        PT_HIP(h, hipStreamWaitEvent(h->stream, h->chainDone, 0));
        h->chainPending = false;
    }
    for (int j = 0; j < kMaxStripes; j++) {
        if (h->stripePending[j]) {
            if (j > 0) PT_HIP(h, hipStreamWaitEvent(h->stream, h->stripeDone[j], 0)); // (stripe 0 runs on the main stream itself)
            h->stripePending[j] = false;
        }
    }
    // hand-over repair of the launches since the last join (everything they wrote is behind h->stream now).  A caller that synchronises
    // h->stream next and then calls settle_handover() leaves it to that — unless the image still carries tags: the alpha pass that
    // follows such a join would wipe out what the repair reads.
    const bool raised = h->hostErrWord && *(volatile unsigned int *)h->hostErrWord;
    if (raised || (!h->unverified.empty() && (repairNow || h->tagsLive))) {
        if (raised) note_abandonment(h);
        if (int rc = enqueue_repairs(h, !raised && !h->repairPendingAll)) return rc;
        h->repairPendingAll = false;
    }
    return resolve_fused_orphan(h);
}

// h->stream has been synchronised behind join_stripes(h, false): every launch of the handle is complete.  Usually that is all there is
// to know (flag down: forget the launches); an abandoned launch is repaired here, behind one more synchronisation.
int settle_handover(pt_handle h)
{
    if (h->hostErrWord && *(volatile unsigned int *)h->hostErrWord) {
        note_abandonment(h);
        // (the launches the flag belongs to may already have been repaired by an earlier join; then only the closing kernel runs)
        if (int rc = enqueue_repairs(h, false)) return rc;
        h->repairPendingAll = false;
        PT_HIP(h, hipStreamSynchronize(h->stream));
    }
    if (!h->unverified.empty()) FEED_LOG("settle: %zu launches forgotten (synchronised, flag down)", h->unverified.size());
    h->unverified.clear();
    if (h->repairCheckDue) {
        // Repair passes ran since the last look (rare: a contended device).  A pixel whose tag fits nothing the launch sequence can have
        // left means the image cannot be trusted: that is an error of the call that observes it, not a debug counter.
        h->repairCheckDue = false;
        unsigned int ctl[4] = {0, 0, 0, 0};
        PT_HIP(h, hipMemcpy(ctl, h->dRepairCtl, sizeof ctl, hipMemcpyDeviceToHost));
        if (ctl[1] != h->inconsistentSeen) {
            h->inconsistentSeen = ctl[1];
            return fail(h, PT_E_HIP, "hand-over repair: a pixel's frame tag fits no launch of this handle (image inconsistent; pt_reset / pt_set_size start over)");
        }
    }
    return PT_OK;
}

// The host is about to observe the accumulation image (read it, hand out its pointer, synchronise on a bound buffer):
// chained launches leave frame tags in its alpha channel, the reference's constant is 1 (compute.glsl:129).
int fix_alpha(pt_handle h, bool repairNow)
{
    if (int rc = join_stripes(h, repairNow)) return rc;
    if (h->tagsLive) {
        PT_HIP(h, pt::launch_set_alpha(h->accum(), h->tilePixels(), h->stream));
        h->tagsLive = false;
    }
    return PT_OK;
}

// Stripe 0 of a striped frame runs on the handle's main stream, stripe j > 0 on helper stream j, created on first use.  HIP
// multiplexes streams onto a few hardware queues (4 by default) and two streams that share a queue execute in order: the
// fewer streams the library keeps busy, the smaller the chance that two of its concurrent launches (the two stripes, or a
// stripe and the present copy) end up behind each other — the default frame needs main + 1 helper (+ the copy stream).
int ensure_stripe(pt_handle h, int j)
{
    if (j > 0 && !h->stripeStream[j]) PT_HIP(h, hipStreamCreateWithFlags(&h->stripeStream[j], hipStreamNonBlocking));
    if (!h->stripeDone[j]) PT_HIP(h, hipEventCreateWithFlags(&h->stripeDone[j], hipEventDisableTiming));
    return PT_OK;
}

hipStream_t stripe_stream(pt_handle h, int j) { return j == 0 ? h->stream : h->stripeStream[j]; }
} // namespace ptimpl

namespace {

int ensure_accum(pt_handle h)
{
    size_t need = h->tilePixels();
    if (need > h->accumCapacity) {
        if (h->dAccum) PT_HIP(h, hipFree(h->dAccum));
        h->dAccum = nullptr;
        h->accumCapacity = 0;
        PT_HIP(h, hipMalloc((void **)&h->dAccum, need * sizeof(float4)));
        h->accumCapacity = need;
    }
    // cached tile masks (FrameArgs::tileMasks) and FrameArgs::tileFlags: sized with the image so that pt_render never allocates or synchronises
    if ((size_t)((h->width + 7) / 8) * (size_t)((h->rows + 7) / 8) > h->tileMaskTiles) {
        const size_t tiles = (size_t)((h->width + 7) / 8) * (size_t)((h->rows + 7) / 8);
        PT_HIP(h, hipStreamSynchronize(h->stream)); // (callers have joined the handle's streams: nothing reads the old buffer any more)
        if (h->dTileMasks) PT_HIP(h, hipFree(h->dTileMasks));
        h->dTileMasks = nullptr;
        h->tileMaskTiles = 0;
        h->tileMasksValid = false;
        PT_HIP(h, hipMalloc((void **)&h->dTileMasks, tiles * pt::kTileMaskWords * sizeof(unsigned long long)));
        h->tileMaskTiles = tiles;
    }
    const size_t flagTiles = (size_t)((h->width + 7) / 8) * (size_t)((h->rows + 7) / 8);
    if (flagTiles > h->tileFlagTiles) {
        PT_HIP(h, hipStreamSynchronize(h->stream));
        if (h->dTileFlags) PT_HIP(h, hipFree(h->dTileFlags));
        h->dTileFlags = nullptr;
        h->tileFlagTiles = 0;
        PT_HIP(h, hipMalloc((void **)&h->dTileFlags, flagTiles * sizeof(unsigned int)));
        PT_HIP(h, hipMemsetAsync(h->dTileFlags, 0, flagTiles * sizeof(unsigned int), h->stream));
        h->tileFlagTiles = flagTiles;
    }
#ifdef PT_AUDIT
    if (need > h->auditCapacity) {
        PT_HIP(h, hipStreamSynchronize(h->stream));
        if (h->dAudit) PT_HIP(h, hipFree(h->dAudit));
        h->dAudit = nullptr;
        h->auditCapacity = 0;
        PT_HIP(h, hipMalloc((void **)&h->dAudit, need * sizeof(unsigned long long)));
        h->auditCapacity = need;
    }
#endif
    return PT_OK;
}

// hand-over audit: the pixels' history is unknown from here on (cleared / restored / re-bound image, frame counter reset).
// Call with the streams joined; stream-ordered on the main stream.
int audit_forget(pt_handle h)
{
    if (h->dAudit) PT_HIP(h, hipMemsetAsync(h->dAudit, 0xFF, h->tilePixels() * sizeof(unsigned long long), h->stream));
    return PT_OK;
}

int clear_accum(pt_handle h)
{
    if (h->boundAccum && h->boundBytes < h->tilePixels() * sizeof(float4))
        return fail(h, PT_E_BAD_ARGUMENT, "bound result buffer is smaller than the tile");
    if (int rc = join_stripes(h)) return rc;
    PT_HIP(h, pt::launch_clear(h->accum(), h->tilePixels(), h->stream));
    h->tagsLive = false;
    return audit_forget(h);
}

} // namespace

extern "C" {

PT_API const char *pt_version(void) { return "mi355pt 0.1 (gfx950, pt-f32)"; }

PT_API int pt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

PT_API const char *pt_last_error(pt_handle h)
{
    if (h && h->magic == ptimpl::kAlive) return h->error.c_str();
    return g_create_error.c_str();
}

PT_API int pt_create(int device_id, int width, int height, pt_handle *out)
{
    if (!out) return fail(nullptr, PT_E_BAD_ARGUMENT, "out == NULL");
    *out = nullptr;
    if (width <= 0 || height <= 0) return fail(nullptr, PT_E_BAD_ARGUMENT, "width/height must be positive");
    if (width > PT_MAX_IMAGE_DIM || height > PT_MAX_IMAGE_DIM)
        return fail(nullptr, PT_E_OUT_OF_RANGE, "width/height exceed PT_MAX_IMAGE_DIM (32767)");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, PT_E_NO_DEVICE, "no HIP device available (libmi355pt has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail(nullptr, PT_E_BAD_ARGUMENT, "device_id out of range");
    pt_renderer *h = new (std::nothrow) pt_renderer();
    if (!h) return fail(nullptr, PT_E_OUT_OF_MEMORY, "host allocation failed");
    h->device = device_id;
    { // tuning knobs (pt_tuning.hpp: set through pt_debug_set only; the library reads no environment variables)
        const pt::Tuning &t = pt::tuning();
        if (t.drainCompaction >= -1) h->drainCompaction = t.drainCompaction;
        if (t.batchWorkgroupsPerCU >= 1 && t.batchWorkgroupsPerCU <= 8) h->batchWorkgroupsPerCU = t.batchWorkgroupsPerCU;
        if (t.frameBatch >= 1 && t.frameBatch <= 64) { h->maxBatch = t.frameBatch; h->maxBatchExplicit = true; }
        if (t.queueChunk >= 1 && t.queueChunk <= 1024) h->queueChunk = t.queueChunk;
    }
    h->width = width;
    h->height = height;
    h->y0 = 0;
    h->rows = height;
#define PT_CREATE_HIP(call)                                                                                            \
    do {                                                                                                               \
        hipError_t e2_ = (call);                                                                                       \
        if (e2_ != hipSuccess) {                                                                                       \
            int rc_ = hip_fail(nullptr, e2_, #call);                                                                   \
            pt_destroy(h);                                                                                             \
            return rc_;                                                                                                \
        }                                                                                                              \
    } while (0)
    PT_CREATE_HIP(hipSetDevice(device_id));
    PT_CREATE_HIP(hipStreamCreateWithFlags(&h->ownStream, hipStreamNonBlocking));
    h->stream = h->ownStream;
    PT_CREATE_HIP(hipEventCreateWithFlags(&h->inputsReady, hipEventDisableTiming));
    PT_CREATE_HIP(hipEventCreateWithFlags(&h->gatherReady, hipEventDisableTiming));
    PT_CREATE_HIP(hipStreamCreateWithFlags(&h->copyStream, hipStreamNonBlocking));
    PT_CREATE_HIP(hipEventCreate(&h->evBegin));
    PT_CREATE_HIP(hipEventCreate(&h->evEnd));
    PT_CREATE_HIP(hipMalloc((void **)&h->dObjects, PT_GAME_OBJECTS_UBO_SIZE));
    PT_CREATE_HIP(hipMemsetAsync(h->dObjects, 0, PT_GAME_OBJECTS_UBO_SIZE, h->stream));
    PT_CREATE_HIP(hipMalloc((void **)&h->dGrid, (ptgrid::kMaxCells + 1) * 2 + ptgrid::kMaxRefs + 16));
    PT_CREATE_HIP(hipMalloc((void **)&h->dLut, 256 * sizeof(float)));
    PT_CREATE_HIP(hipMalloc((void **)&h->dQueue, ptimpl::kQueueWords * sizeof(unsigned int)));
    PT_CREATE_HIP(hipMemsetAsync(h->dQueue, 0, ptimpl::kQueueWords * sizeof(unsigned int), h->stream));
    PT_CREATE_HIP(hipHostMalloc((void **)&h->hostErrWord, sizeof(unsigned int), hipHostMallocMapped));
    *h->hostErrWord = 0;
    PT_CREATE_HIP(hipHostGetDevicePointer((void **)&h->devErrWord, h->hostErrWord, 0));
    PT_CREATE_HIP(hipMalloc((void **)&h->dAbandon, sizeof(unsigned int)));
    PT_CREATE_HIP(hipMemsetAsync(h->dAbandon, 0xFF, sizeof(unsigned int), h->stream)); // ABANDON_NONE
    PT_CREATE_HIP(hipMalloc((void **)&h->dRepairCtl, 4 * sizeof(unsigned int)));
    PT_CREATE_HIP(hipMemsetAsync(h->dRepairCtl, 0, 4 * sizeof(unsigned int), h->stream));
    {
        // the hand-over's wall-clock budget in units of 1,024 ticks of the device's constant-rate counter (100 MHz on gfx950)
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_id) != hipSuccess || khz <= 0) khz = 100000;
        (void)hipGetLastError();
        h->wallClockKhz = khz;
        const pt::Tuning &t = pt::tuning();
        const double unitsPerMs = (double)khz / 1024.0;
        double budget = (double)t.handoverBudgetMs * unitsPerMs, check = (double)t.handoverCheckUs * unitsPerMs / 1000.0;
        if (budget > 1.0e9) budget = 1.0e9; // (the kernels compare signed 32-bit differences: ~2.8 hours)
        if (check > budget / 4.0) check = budget / 4.0; // (short test budgets: look often enough)
        h->waitBudgetUnits = (unsigned int)budget;
        h->waitCheckUnits = (unsigned int)check;
    }
    PT_CREATE_HIP(hipHostMalloc((void **)&h->hostFeed, pt_renderer::kFeedHostWords * sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent));
    for (int i = 0; i < pt_renderer::kFeedHostWords; i++) h->hostFeed[i] = pt::kFeedClosed;
    PT_CREATE_HIP(hipHostMalloc((void **)&h->hostFeedDone, pt_renderer::kFeedHostWords * sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h->hostFeedDone, 0, pt_renderer::kFeedHostWords * sizeof(unsigned int));
    PT_CREATE_HIP(hipHostGetDevicePointer((void **)&h->devFeedHostDone, h->hostFeedDone, 0));
    PT_CREATE_HIP(hipHostGetDevicePointer((void **)&h->devFeedHost, h->hostFeed, 0));
    PT_CREATE_HIP(hipMalloc((void **)&h->dFeedDev, (size_t)2 * pt::kFeedBcastSlots * pt::kFeedBcastStride * sizeof(unsigned int)));
    PT_CREATE_HIP(hipMemsetAsync(h->dFeedDev, 0, (size_t)2 * pt::kFeedBcastSlots * pt::kFeedBcastStride * sizeof(unsigned int), h->stream));
    PT_CREATE_HIP(hipMalloc((void **)&h->dFeedDone, (size_t)16 * pt::kFeedDoneStride * sizeof(unsigned long long)));
    PT_CREATE_HIP(hipMemsetAsync(h->dFeedDone, 0, (size_t)16 * pt::kFeedDoneStride * sizeof(unsigned long long), h->stream));
    PT_CREATE_HIP(hipHostMalloc((void **)&h->hostStarted, ptimpl::kStartedWords * sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h->hostStarted, 0, ptimpl::kStartedWords * sizeof(unsigned int));
    PT_CREATE_HIP(hipHostGetDevicePointer((void **)&h->devStarted, h->hostStarted, 0));
#ifdef PT_AUDIT
    {
        const size_t words = 4 + (size_t)pt::kAuditLogRecords * pt::kAuditRecordWords;
        PT_CREATE_HIP(hipHostMalloc((void **)&h->hostAuditLog, words * sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h->hostAuditLog, 0, words * sizeof(unsigned int));
        PT_CREATE_HIP(hipHostGetDevicePointer((void **)&h->devAuditLog, h->hostAuditLog, 0));
    }
#endif
    {
        hipDeviceProp_t prop;
        PT_CREATE_HIP(hipGetDeviceProperties(&prop, device_id));
        h->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    float lut[256];
    make_srgb_lut(lut);
    PT_CREATE_HIP(hipMemcpyAsync(h->dLut, lut, sizeof lut, hipMemcpyHostToDevice, h->stream));
    PT_CREATE_HIP(hipStreamSynchronize(h->stream)); // `lut` is a stack array
#undef PT_CREATE_HIP
    int rc = ensure_accum(h);
    if (rc == PT_OK) rc = clear_accum(h);
    if (rc != PT_OK) {
        g_create_error = h->error;
        pt_destroy(h);
        return rc;
    }
    *out = h;
    return PT_OK;
}

PT_API int pt_destroy(pt_handle h)
{
    PT_CHECK_HANDLE(h);
    if (h->isGroup()) return ptimpl::group_destroy(h);
    h->pendingFrames = 0; // frames nobody can observe any more are not worth launching
    (void)hipSetDevice(h->device);
    ptimpl::feed_close(h); // (a frame-fed launch still waiting for frames ends now)
    // (launches may still write present snapshots, tone maps read them: every stream of the handle is drained before they are freed)
    for (int j = 0; j < ptimpl::kMaxStripes; j++)
        if (h->stripeStream[j]) (void)hipStreamSynchronize(h->stripeStream[j]);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->copyStream) (void)hipStreamSynchronize(h->copyStream);
    for (int k = 0; k < pt_renderer::kSnapshots; k++) {
        if (h->snapRead[k]) (void)hipEventDestroy(h->snapRead[k]);
        if (h->dSnap[k]) (void)hipFree(h->dSnap[k]);
    }
    ptimpl::free_slots(h);
    if (h->copyStream) (void)hipStreamDestroy(h->copyStream);
    if (h->gatherReady) (void)hipEventDestroy(h->gatherReady);
    for (int j = 0; j < ptimpl::kMaxStripes; j++)
        if (h->stripeStream[j]) (void)hipStreamSynchronize(h->stripeStream[j]);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int j = 0; j < ptimpl::kMaxStripes; j++) {
        if (h->stripeDone[j]) (void)hipEventDestroy(h->stripeDone[j]);
        if (h->stripeStream[j]) (void)hipStreamDestroy(h->stripeStream[j]);
    }
    if (h->inputsReady) (void)hipEventDestroy(h->inputsReady);
    for (hipEvent_t e : h->launchEvents)
        if (e) (void)hipEventDestroy(e); // (mainDone / chainDone alias entries of this ring)
    if (h->dAbandon) (void)hipFree(h->dAbandon);
    if (h->dRepairCtl) (void)hipFree(h->dRepairCtl);
    if (h->dTileFlags) (void)hipFree(h->dTileFlags);
    if (h->dObjects) (void)hipFree(h->dObjects);
    if (h->dGrid) (void)hipFree(h->dGrid);
    if (h->dLut) (void)hipFree(h->dLut);
    if (h->dQueue) (void)hipFree(h->dQueue);
    if (h->dTileMasks) (void)hipFree(h->dTileMasks);
    if (h->hostErrWord) (void)hipHostFree(h->hostErrWord);
    if (h->hostStarted) (void)hipHostFree(h->hostStarted);
    if (h->hostFeed) (void)hipHostFree(h->hostFeed);
    if (h->hostFeedDone) (void)hipHostFree(h->hostFeedDone);
    if (h->dFeedDev) (void)hipFree(h->dFeedDev);
    if (h->dFeedDone) (void)hipFree(h->dFeedDone);
    for (hipEvent_t e : h->feedWordBusy)
        if (e) (void)hipEventDestroy(e);
    if (h->feedResetEvent) (void)hipEventDestroy(h->feedResetEvent);
    if (h->hostAuditLog) (void)hipHostFree(h->hostAuditLog);
    if (h->dAudit) (void)hipFree(h->dAudit);
    if (h->dEnv) (void)hipFree(h->dEnv);
    if (h->dAccum) (void)hipFree(h->dAccum);
    if (h->dRgba8) (void)hipFree(h->dRgba8);
    if (h->dTimeline) (void)hipFree(h->dTimeline);
    if (h->evBegin) (void)hipEventDestroy(h->evBegin);
    if (h->evEnd) (void)hipEventDestroy(h->evEnd);
    if (h->ownStream) (void)hipStreamDestroy(h->ownStream);
    h->magic = 0;
    delete h;
    return PT_OK;
}

PT_API int pt_set_size(pt_handle h, int width, int height)
{
    PT_CHECK_HANDLE(h);
    if (width <= 0 || height <= 0) return fail(h, PT_E_BAD_ARGUMENT, "width/height must be positive");
    if (width > PT_MAX_IMAGE_DIM || height > PT_MAX_IMAGE_DIM)
        return fail(h, PT_E_OUT_OF_RANGE, "width/height exceed PT_MAX_IMAGE_DIM (32767)");
    if (h->isGroup()) return ptimpl::group_set_size(h, width, height);
    if (int rc = flush_frames(h)) return rc; // pending frames were rendered with the inputs as they were
    if (int rc = bind_device(h)) return rc;
    if (int rc = join_stripes(h)) return rc; // (the launches so far — and their hand-over repair passes — belong to the buffers as they are)
    h->width = width;
    h->height = height;
    h->y0 = 0;
    h->rows = height;
    h->bandRows = 0;
    h->frame = 0; // PathTracer.cs:133
    h->tileMasksValid = false;
    h->launchesSinceInputChange = 0;
    if (int rc = ensure_accum(h)) return rc;
    return clear_accum(h);
}

PT_API int pt_set_tile(pt_handle h, int y0, int rows)
{
    PT_CHECK_HANDLE(h);
    if (h->isGroup()) return fail(h, PT_E_BAD_ARGUMENT, "a group handle owns its tiling (pt_multi_set_partition)");
    if (int rc = flush_frames(h)) return rc; // pending frames were rendered with the inputs as they were
    if (y0 < 0 || rows <= 0 || y0 + rows > h->height) return fail(h, PT_E_BAD_ARGUMENT, "tile outside the image");
    if (int rc = bind_device(h)) return rc;
    if (int rc = join_stripes(h)) return rc; // (see pt_set_size)
    h->y0 = y0;
    h->rows = rows;
    h->bandRows = 0;
    h->frame = 0;
    h->tileMasksValid = false;
    h->launchesSinceInputChange = 0;
    if (int rc = ensure_accum(h)) return rc;
    return clear_accum(h);
}

PT_API int pt_set_interleaved_tile(pt_handle h, int rank, int world, int band_rows)
{
    PT_CHECK_HANDLE(h);
    if (h->isGroup()) return fail(h, PT_E_BAD_ARGUMENT, "a group handle owns its tiling (pt_multi_set_partition)");
    if (int rc = flush_frames(h)) return rc; // pending frames were rendered with the inputs as they were
    if (world < 1 || rank < 0 || rank >= world || band_rows < 8 || (band_rows & 7))
        return fail(h, PT_E_BAD_ARGUMENT, "need 0 <= rank < world and band_rows a positive multiple of 8");
    if (int rc = bind_device(h)) return rc;
    // rows owned: bands rank, rank + world, ... of band_rows rows (the last band of the image may be partial)
    long long rows = 0;
    for (long long b = rank; b * band_rows < h->height; b += world) {
        long long top = (b + 1) * band_rows;
        rows += (top < h->height ? top : h->height) - b * band_rows;
    }
    if (rows <= 0) return fail(h, PT_E_BAD_ARGUMENT, "this rank owns no rows (image too small for world * band_rows)");
    if (int rc = join_stripes(h)) return rc; // (see pt_set_size)
    h->y0 = 0;
    h->rows = (int)rows;
    h->bandRows = band_rows;
    h->bandWorld = world;
    h->bandRank = rank;
    h->frame = 0;
    h->tileMasksValid = false;
    h->launchesSinceInputChange = 0;
    if (int rc = ensure_accum(h)) return rc;
    return clear_accum(h);
}

// group handles: replicate a call to every part; the first failure is reported on the group handle
#define PT_FAN_OUT(h, call)                                                                                            \
    do {                                                                                                               \
        if ((h)->isGroup()) {                                                                                          \
            for (pt_handle part : (h)->parts) {                                                                        \
                int rc_ = (call);                                                                                      \
                if (rc_ != PT_OK) return fail((h), rc_, part->error);                                                  \
            }                                                                                                          \
            return PT_OK;                                                                                              \
        }                                                                                                              \
    } while (0)

PT_API int pt_reset(pt_handle h)
{
    PT_CHECK_HANDLE(h);
    PT_FAN_OUT(h, pt_reset(part));
    if (int rc = flush_frames(h)) return rc; // pending frames were rendered with the inputs as they were
    // (the frame counter goes backwards: the tags in the image must not be mistaken for frames of the new sequence)
    if (int rc = bind_device(h)) return rc;
    if (int rc = ptimpl::fix_alpha(h)) return rc;
    h->frame = 0; // PathTracer.cs:139 — frame 0 weights the old contents by 0, so no clear is needed
    return audit_forget(h);
}

PT_API int pt_set_params(pt_handle h, int num_spheres, int num_cuboids, int ray_depth, int spp, float focal_length,
                         float aperture_diameter)
{
    PT_CHECK_HANDLE(h);
    PT_FAN_OUT(h, pt_set_params(part, num_spheres, num_cuboids, ray_depth, spp, focal_length, aperture_diameter));
    // The reference's setters run whenever the GUI touches a slider, changed or not (PathTracer.cs:11-83): values the renderer already
    // has are not an input change — nothing is flushed, nothing invalidated (bitwise compare: -0.0f / NaN count as changes).
    if (num_spheres == h->numSpheres && num_cuboids == h->numCuboids && ray_depth == h->rayDepth && spp == h->spp &&
        std::memcmp(&focal_length, &h->focalLength, sizeof(float)) == 0 && std::memcmp(&aperture_diameter, &h->apertureDiameter, sizeof(float)) == 0)
        return PT_OK;
    if (num_spheres < 0 || num_spheres > PT_MAX_SPHERES || num_cuboids < 0 || num_cuboids > PT_MAX_CUBOIDS)
        return fail(h, PT_E_OUT_OF_RANGE, "object counts exceed the GameObjectsUBO arrays (256 spheres / 64 cuboids)");
    if (ray_depth < 0 || spp < 1) return fail(h, PT_E_BAD_ARGUMENT, "ray_depth must be >= 0 and spp >= 1");
    // the kernels carry bounce / sample counters in 12-bit fields of their path records
    if (ray_depth > PT_MAX_RAY_DEPTH || spp > PT_MAX_SPP)
        return fail(h, PT_E_OUT_OF_RANGE, "ray_depth / spp exceed PT_MAX_RAY_DEPTH / PT_MAX_SPP (4095)");
    if (int rc = flush_frames(h)) return rc; // pending frames were rendered with the inputs as they were
    if (num_spheres != h->numSpheres) h->gridDirty = true;
    if (num_spheres != h->numSpheres || num_cuboids != h->numCuboids || focal_length != h->focalLength || aperture_diameter != h->apertureDiameter) {
        h->tileMasksValid = false; // (the cached masks cull spheres AND cuboids against the lens' cone)
        h->launchesSinceInputChange = 0;
    }
    h->numSpheres = num_spheres;
    h->numCuboids = num_cuboids;
    h->rayDepth = ray_depth;
    h->spp = spp;
    h->focalLength = focal_length;
    h->apertureDiameter = aperture_diameter;
    return PT_OK;
}

PT_API int pt_upload_basic_data(pt_handle h, int byte_offset, int size, const void *src)
{
    PT_CHECK_HANDLE(h);
    PT_FAN_OUT(h, pt_upload_basic_data(part, byte_offset, size, src));
    if (!src) return fail(h, PT_E_BAD_ARGUMENT, "src == NULL");
    if (byte_offset < 0 || size < 0 || (long long)byte_offset + size > PT_BASIC_DATA_UBO_SIZE)
        return fail(h, PT_E_OUT_OF_RANGE, "BasicDataUBO range outside [0,144)");
    // The reference re-uploads InvView and ViewPos on EVERY focused update, moved or not (MainWindow.cs:131-132): two SubData calls between
    // every pair of Render() calls.  Bytes the renderer already holds are not an input change: no flush (the frames keep pipelining), no
    // invalidation (the cached tile masks stay valid).  A changed byte flushes first — pending frames were rendered with the camera as it was.
    if (std::memcmp(h->basic + byte_offset, src, (size_t)size) == 0) return PT_OK;
    h->statFlushes++;
    if (int rc = flush_frames(h)) return rc;
    std::memcpy(h->basic + byte_offset, src, (size_t)size);
    h->tileMasksValid = false; h->launchesSinceInputChange = 0;
    return PT_OK;
}

PT_API int pt_upload_game_objects(pt_handle h, int byte_offset, int size, const void *src)
{
    PT_CHECK_HANDLE(h);
    PT_FAN_OUT(h, pt_upload_game_objects(part, byte_offset, size, src));
    if (!src) return fail(h, PT_E_BAD_ARGUMENT, "src == NULL");
    if (byte_offset < 0 || size < 0 || (long long)byte_offset + size > PT_GAME_OBJECTS_UBO_SIZE)
        return fail(h, PT_E_OUT_OF_RANGE, "GameObjectsUBO range outside [0,26624)");
    if (size == 0) return PT_OK;
    // (an object the GUI re-uploads unchanged, Gui.cs:212-216, or a scene reload with the same bytes, MainWindow.cs:119-123: see
    // pt_upload_basic_data — nothing is joined, copied or invalidated)
    if (std::memcmp(h->objectsShadow + byte_offset, src, (size_t)size) == 0) return PT_OK;
    h->statFlushes++;
    if (int rc = bind_device(h)) return rc;
    if (int rc = join_stripes(h)) return rc;
    // pageable source: HIP stages the bytes before returning, so the caller may reuse `src` immediately
    PT_HIP(h, hipMemcpyAsync((char *)h->dObjects + byte_offset, src, (size_t)size, hipMemcpyHostToDevice, h->stream));
    std::memcpy(h->objectsShadow + byte_offset, src, (size_t)size);
    if (byte_offset < PT_MAX_SPHERES * 80) h->gridDirty = true; // (the Spheres[] array ends at byte 20,480)
    h->tileMasksValid = false; // (any object: the cached per-tile masks hold a sphere mask and a cuboid mask)
    h->launchesSinceInputChange = 0;
    return PT_OK;
}

PT_API int pt_set_environment(pt_handle h, int face_size, int format, const void *const faces[6])
{
    PT_CHECK_HANDLE(h);
    PT_FAN_OUT(h, pt_set_environment(part, face_size, format, faces));
    if (face_size <= 0 || face_size > 16384) return fail(h, PT_E_BAD_ARGUMENT, "face_size out of range");
    if (format != PT_ENV_RGBA32F && format != PT_ENV_SRGB8_A8) return fail(h, PT_E_BAD_ARGUMENT, "unknown format");
    if (!faces) return fail(h, PT_E_BAD_ARGUMENT, "faces == NULL");
    for (int f = 0; f < 6; f++)
        if (!faces[f]) return fail(h, PT_E_BAD_ARGUMENT, "faces[i] == NULL");
    if (int rc = bind_device(h)) return rc;
    if (int rc = join_stripes(h)) return rc;
    size_t faceBytes = (size_t)face_size * face_size * (format == PT_ENV_RGBA32F ? 16 : 4);
    size_t total = faceBytes * 6;
    if (total > h->envBytes) {
        PT_HIP(h, hipStreamSynchronize(h->stream)); // a queued frame may still sample the old cube
        if (h->dEnv) PT_HIP(h, hipFree(h->dEnv));
        h->dEnv = nullptr;
        h->envBytes = 0;
        PT_HIP(h, hipMalloc(&h->dEnv, total));
        h->envBytes = total;
    }
    for (int f = 0; f < 6; f++)
        PT_HIP(h, hipMemcpyAsync((char *)h->dEnv + f * faceBytes, faces[f], faceBytes, hipMemcpyHostToDevice, h->stream));
    h->envSize = face_size;
    h->envFormat = format;
    return PT_OK;
}

} // extern "C"

namespace {

bool gpu_busy(pt_handle h);

// ---------------------------------------------------------------------------------------------------------------------------------------
// Launching frames [firstFrame, firstFrame + n) with the handle's current inputs, in steps (round 6: what used to be one 326-line function):
//   fill_frame_args      the inputs as the kernels take them (+ the sphere grid of a changed scene)
//   choose_launch_mode   tagged or plain, stripes, kernel variant, workgroups per CU
//   use_tile_masks       cached per-tile cull masks: use / rebuild
//   wait_for_residency   launch chaining: may this launch run BESIDE its predecessor?
//   arm_handover         sequence number + abandon word of a tagged launch; remember_launch: the record its repair pass needs
//   launch_chained / launch_single_stream / launch_striped     the three ways a launch reaches the GPU
// launch_frames() strings them together.  waitUs: how long the call may wait for the predecessor launch to become resident;
// lastOfFlush: this launch holds the newest pending frame (only it may store alpha = 1 last / write a present snapshot).
struct LaunchMode {
    bool tagged = false, chainable = false;
    int stripes = 1, kernelVariant = 0;
};

// ---- step 1: inputs -> FrameArgs (everything that does not depend on how the launch reaches the GPU)
int fill_frame_args(pt_handle h, pt::FrameArgs &a, int firstFrame, int n)
{
    std::memcpy(a.invProj, h->basic, 64);
    std::memcpy(a.invView, h->basic + 64, 64);
    std::memcpy(a.viewPos, h->basic + 128, 12);
    a.focalLength = h->focalLength;
    a.apertureDiameter = h->apertureDiameter;
    a.width = h->width;
    a.height = h->height;
    a.invW = 1.0f / (float)h->width; // (correctly rounded, like the device's f_div_ieee)
    a.invH = 1.0f / (float)h->height;
    a.numSpheres = h->numSpheres;
    a.numCuboids = h->numCuboids;
    a.rayDepth = h->rayDepth;
    a.spp = h->spp;
    a.frame = firstFrame;
    a.batchFrames = n;
    a.envSize = h->envSize;
    a.envFormat = h->envFormat;
    a.objects = h->dObjects;
    a.env = h->dEnv;
    a.srgbLut = h->dLut;
    a.tilesX = (h->width + 7) / 8;
    a.bandRows = h->bandRows;
    a.bandWorld = h->bandWorld;
    a.bandRank = h->bandRank;
    a.localRow0 = 0;
    a.numCUs = h->numCUs;
    // tiles per global ticket: 8; a pipelined launch over a small share of an image (a 1/8 share of 1080p has 4,050 tiles
    // per frame for 6,144 wavefronts) takes 4, which keeps fewer frames in flight at once (+5 % there, neutral above)
    a.queueChunk = h->queueChunk > 0 ? h->queueChunk : (n > 1 && (long long)((h->width + 7) / 8) * ((h->rows + 7) / 8) < 12000 ? 4 : 8);
    a.errorWord = h->devErrWord;
    a.startedFlags = nullptr;
    a.launchSeq = 0;
    a.snapshot = nullptr; // (set next to every a.accum when the launch feeds a present)
    a.audit = nullptr;    // (set next to every a.accum)
    a.auditLog = h->devAuditLog;
    a.auditSabotage = 0;
#ifdef PT_AUDIT
    a.auditSabotage = pt::tuning().auditSabotage;
#endif
    a.timeline = h->dTimeline;
    // sphere grid of large scenes: rebuilt here, before the first launch that sees the changed scene.  Launches still in flight
    // read the old grid: join first, then the copy is ordered behind them on the main stream like a scene upload.
    if (h->gridDirty) {
        h->gridDirty = false;
        h->grid = ptgrid::build((const float *)h->objectsShadow, h->numSpheres);
        ptgrid::sphere_runs((const float *)h->objectsShadow, h->numSpheres, h->sphereRunStart);
        if (h->grid.valid) {
            if (int rc = join_stripes(h)) return rc;
            PT_HIP(h, hipMemcpyAsync(h->dGrid, h->grid.packed.data(), h->grid.packed.size(), hipMemcpyHostToDevice, h->stream));
        }
    }
    a.grid = h->grid.valid ? h->dGrid : nullptr;
    a.gridBytes = h->grid.valid ? (int)h->grid.packed.size() : 0;
    a.gridLdsBytes = 0;
    for (int k = 0; k < 3; k++) {
        a.gridDims[k] = h->grid.dims[k];
        a.gridLo[k] = h->grid.lo[k];
        a.gridHi[k] = h->grid.hi[k];
        a.gridCell[k] = h->grid.cell[k];
        a.gridInvCell[k] = h->grid.invCell[k];
        a.gridCenter[k] = h->grid.center[k];
    }
    a.gridReach2 = h->grid.reach2;
    std::memcpy(a.sphereRunStart, h->sphereRunStart, sizeof(a.sphereRunStart));
    a.tileMasks = nullptr; // (use_tile_masks: only launches that run the tile pass over the handle's whole tile)
    a.tagged = 0;
    a.keepTags = 0;
    a.chainTag = 0.0f;
    a.abandonWord = nullptr;
    a.tileFlags = nullptr;
    a.waitBudget = h->waitBudgetUnits;
    a.waitCheckInterval = h->waitCheckUnits;
    return PT_OK;
}

// ---- step 2: how the frames reach the GPU
// variant -> (kernel variant, stripes): 0 = default (2 stripes x 5 workgroups/CU); 20+k / 30+k / 40+k = 2 / 3 / 4
// stripes of the persistent kernel with k+1 workgroups per CU; everything else = one kernel on the main stream.
// Tagged launches (pixels handed over through alpha tags): every launch of more than one frame, and — once the host has
// pipelined frames on this handle — single frames too, so that they can overlap the launches around them (a host that
// presents every frame, or only ever renders one frame at a time, keeps the faster single-frame stripes) ... and single frames
// whenever the GPU still runs earlier frames of this handle: the frame then chains on the other stream and moves into the wavefront
// slots the previous launch's drain frees (0.154 ms per 1080p frame against 0.182 for the two row stripes); a host that lets the GPU
// run dry between its frames (blocking reads / presents) keeps the stripes, which are the faster way to render ONE frame on an idle machine.
LaunchMode choose_launch_mode(pt_handle h, pt::FrameArgs &a, int n)
{
    LaunchMode m;
    m.kernelVariant = h->variant;
    const bool noSingleTagged = pt::tuning().noSingleTagged != 0; // A/B runs
    m.chainable = h->variant == 0 && !h->externalStream() && !ptimpl::timeline_blocks_pipelining(h);
    // (also the first frame of a burst on an idle GPU once the host has pipelined frames before: the batch that follows then chains
    // on it instead of waiting behind two joined stripes — the driver's `--steps 20` command: 15.4 instead of 14.8 Gsamples/s)
    m.tagged = h->variant == 0 && (n > 1 || (m.chainable && !noSingleTagged && (gpu_busy(h) || (h->sawBatch && h->presentCadence != 1))));
    // (short launches — the interactive modes — take 5 workgroups per CU: 112 instead of 32 free VGPRs per SIMD leave the present's
    // tone map and the runtime's copy kernel room BESIDE the resident persistent wavefronts; 6 per CU starve them until the drain,
    // measured: 0.28 instead of 0.16 ms per displayed frame; a single frame also renders 2 % faster with 5)
    const int shortWg = pt::tuning().shortWorkgroupsPerCU;
    if (m.tagged) { m.stripes = 1; m.kernelVariant = 10 + (n < 8 ? shortWg : h->batchWorkgroupsPerCU) - 1; h->batchLaunched = true; if (n > 1) h->sawBatch = true; }
    else if (h->variant == 0 && h->externalStream()) { m.stripes = 1; m.kernelVariant = 14; } // everything ON the caller's stream
    else if (h->variant == 0) { m.stripes = 2; m.kernelVariant = 14; }
    else if (h->variant >= 20 && h->variant < 50) { m.stripes = h->variant / 10; m.kernelVariant = 10 + h->variant % 10; }
    if (h->rows < 16 * m.stripes) m.stripes = 1; // tiny tiles: not worth splitting
    a.variant = m.kernelVariant;
    // Drain compaction (a thin draining wavefront donates its paths to its workgroup's pool) shortens the tail of ONE
    // launch.  With stripes the tail of one launch is covered by the other stripe's (or the next frame's) main phase, and
    // the pool's LDS and the donor traffic only cost: auto = on for single-launch variants, off for striped frames.
    a.drainCompaction = h->drainCompaction >= 0 ? h->drainCompaction : (m.stripes > 1 || m.tagged ? 0 : 32); // a batch drains once per n frames
    a.tagged = m.tagged ? 1 : 0;
    return m;
}

// ---- step 3: cached tile masks (the tile pass of the spp = 1 kernels, the fresh-tile batch passes of the spp > 1 kernel): valid masks are
// simply used; stale ones are rebuilt once the camera / lens / spheres / tiling have been left alone for two launches — the launches still
// in flight read the old buffer, so the rebuild joins the two launch streams first (chainBroken: this launch then starts on the main
// stream, behind the mask kernel).  Call with a.tilesX / a.tilesY / a.y0 / a.rows set.
int use_tile_masks(pt_handle h, pt::FrameArgs &a)
{
    if (pt::tuning().tileMasks == 0) return PT_OK;
    // (counted in FRAMES since round 6 — a frame-fed launch takes up to 64 frames: a host that moves the camera every frame still never
    // gets here with more than one)
    const bool settled = h->launchesSinceInputChange > 2;
    h->launchesSinceInputChange += a.batchFrames;
    if (!h->tileMasksValid && settled) {
        if (int rc = join_stripes(h)) return rc;
        // (the buffer is sized by ensure_accum with the image: nothing is allocated on the render path)
        if ((size_t)a.tilesX * a.tilesY <= h->tileMaskTiles) {
            PT_HIP(h, pt::launch_tile_masks(a, h->dTileMasks, h->stream));
            h->tileMasksValid = true;
            h->statMaskBuilds++;
        }
    }
    if (h->tileMasksValid) a.tileMasks = h->dTileMasks;
    return PT_OK;
}

// ---- step 4: residency.  A launch may run BESIDE its predecessor (on the other stream) only if that launch is fully resident — then it
// can only ever get the slots the predecessor's workgroups give up when they are done; otherwise it goes behind it on the same stream (no
// overlap, always safe).  After an abandonment the device is evidently contended: for a while launches run behind their predecessor.
bool predecessor_resident(pt_handle h)
{
    for (int i = 0; i < h->lastWorkgroups; i++)
        if (((volatile unsigned int *)h->hostStarted)[i] != h->launchSeq) return false;
    return true;
}
bool wait_for_residency(pt_handle h, const pt::FrameArgs &a, long waitUs)
{
    if (pt::tuning().serialLaunches != 0) return false; // (A/B and profile runs: strictly one launch after the other)
    if (h->overlapHoldoff > 0) h->overlapHoldoff--;
    const bool mayChain = !h->chainBroken && h->lastWorkgroups > 0 && h->lastWorkgroups <= ptimpl::kStartedWords && h->overlapHoldoff == 0;
    bool resident = mayChain && predecessor_resident(h);
    // (small shares — a 1/8 share of 1080p has 4,050 tiles per frame — lose 4 % when two launches overlap: the second launch's first
    // frames all wait for the first one's last; measured, tools/emulate_strong.py)
    const bool bigShare = (long long)a.tilesX * a.tilesY >= 12000;
    if (mayChain && !resident && (bigShare || h->flushFinal)) {
        // Back-pressure (round 3): the host is more than one launch ahead of the GPU — the predecessor still queues behind ITS
        // predecessor.  Launching behind it on the same stream would expose a full drain + ramp per launch (0.096 ms at 1080p);
        // instead the call waits until the predecessor is resident (i.e. until the launch before it has left the machine) and
        // then chains.  Bounded, so a GPU shared with another process falls back to the always-safe same-stream order.  Round 5:
        // pt_render itself no longer waits here — it keeps frames pending until the predecessor is resident (launch_ready) and
        // only a call that blocks anyway (pt_synchronize, a read, a present) waits, for at most chain_wait_us.
        const auto t0 = std::chrono::steady_clock::now();
        while (!resident) {
            const auto waited = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            if (waited >= waitUs) break;
            if (waited > 200) std::this_thread::sleep_for(std::chrono::microseconds(20)); // (long waits: do not burn the core)
            resident = predecessor_resident(h);
        }
    }
    return resident;
}

// ---- step 5: the hand-over bound's bookkeeping (pt_renderer.hpp).  A tagged launch gets a sequence number and the handle's abandon word
// (call with a.chainTag / a.keepTags set), and is remembered — its kernel argument and an event behind it — until a join has put its
// repair pass behind it or it is seen complete with the flag down.
void arm_handover(pt_handle h, pt::FrameArgs &a)
{
    a.launchSeq = ++h->launchSeq;
    if (a.launchSeq == 0 || a.launchSeq >= 0xfffffff0u) a.launchSeq = h->launchSeq = 1; // (0: a fresh roll call; ~0: "no launch abandoned")
    a.abandonWord = h->dAbandon;
    a.tileFlags = a.chainTag == 0.0f ? h->dTileFlags : nullptr; // (the first launch of a chain: alpha = 1 could mean "untouched" or "finished")
}
void remember_launch(pt_handle h, const pt::FrameArgs &a, hipEvent_t done)
{
    // A launch seen COMPLETE with the flag DOWN — read in that order: an abandoning launch raises the flag before it ends — ran to its
    // end and needs no repair pass.
    auto flag_down = [&]() -> bool { return !(h->hostErrWord && *(volatile unsigned int *)h->hostErrWord); };
    while (h->unverified.size() > 2 && hipEventQuery(h->unverified.front().done) == hipSuccess && flag_down()) {
        FEED_LOG("remember: launch %u frames [%d,+%d) seen complete with the flag down: forgotten", h->unverified.front().a.launchSeq, h->unverified.front().a.frame, h->unverified.front().a.batchFrames);
        h->unverified.pop_front();
    }
    (void)hipGetLastError(); // (hipErrorNotReady of the query is not an error)
    static_assert(pt_renderer::kLaunchEvents / 2 <= pt::kMaxUnverifiedLaunches, "frame-tag window (pt_kernels.hpp)");
    if (h->unverified.size() >= (size_t)pt_renderer::kLaunchEvents / 2) { // (the event ring must not lap a remembered launch; also the frame-tag window's bound)
        (void)hipEventSynchronize(h->unverified.front().done);
        if (flag_down()) h->unverified.pop_front();
    }
    h->unverified.push_back({a, done});
    FEED_LOG("launch %u frames [%d,+%d) chainTag %g keepTags %d variant %d spp %d (%zu remembered)", a.launchSeq, a.frame, a.batchFrames, (double)a.chainTag, a.keepTags, a.variant, a.spp, h->unverified.size());
    if (pt::tuning().feedLog >= 2) { // (debug: a blocking look at the device's abandon word and ticket counters)
        unsigned int w = 0, q[2] = {0, 0};
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&w, h->dAbandon, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&q[0], h->dQueue, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&q[1], h->dQueue + ptimpl::kChainQueueWord, 4, hipMemcpyDeviceToHost);
        FEED_LOG("  after launch %u: abandon word %08x, tickets main %u (host %u) chain %u (host %u), flag %u", a.launchSeq, w, q[0], h->stripeQueueBase[0], q[1], h->chainQueueBase,
                 h->hostErrWord ? *(volatile unsigned int *)h->hostErrWord : 0u);
    }
}

// ---- frame-fed launches (pt_renderer.hpp: FeedState).  Wanted when few frames go out and nothing says the host is about to join: the
// reference's own usage.  Conditions that only cost speed elsewhere: one sample per pixel (the tile-pass kernels), a full-size share.
bool feed_wanted(pt_handle h, const pt::FrameArgs &a, int n, bool lastOfFlush)
{
    if (pt::tuning().feed == 0 || pt::tuning().serialLaunches != 0 || n >= 8 || !lastOfFlush || h->flushFinal || a.spp != 1 || (long long)a.tilesX * a.tilesY < pt::tuning().feedMinTiles) return false;
    // ... and a host that has turned deferral OFF (pt_set_frame_batch(1): every Render() is handed to the GPU at once — the interactive
    // setting).  With deferral allowed, frames that arrive while the GPU is busy are collected and go out as one classic launch, which has the
    // lower fixed cost (no open / close hand-shake: the driver's `--steps 20` command measures 17.45 that way against 16.69 Gsamples/s fed)
    if (!h->snapshotTarget && !(h->maxBatchExplicit && h->maxBatch == 1)) return false;
    if (h->feedHoldoff > 0) { // (a host too slow to keep a launch fed: see note_abandonment)
        h->feedHoldoff--;
        return false;
    }
    if (h->snapshotTarget) {
        // A host that shows every frame: the launch tone-maps into the present slots' images itself (fused display) — for slots whose image
        // stays on the device (pt_present_bind_device_image: the interop-style present).  A slot with a host image needs the runtime's copy
        // kernel per frame, which finds no room beside resident wavefronts (measured: it ran when the launch ended) and is PCIe-bound
        // anyway (0.157 ms per 1080p frame): such hosts keep the per-frame launches.
        if (pt::tuning().feedDisplay == 0 || n != 1 || !h->lastPresentBound) return false;
        for (const ptimpl::PresentSlot &s : h->slots)
            if (s.boundDev && s.boundBytes < h->tilePixels() * 4) return false;
    }
    return true;
}

// Turn `a` (a chained launch of frames [firstFrame, firstFrame + n) on stream st = launch stream si) into a frame-fed launch: capacity
// kFeedCapacity, n frames published, control words initialised.  *word = the host word it reads, or -1 when none is free (no fed launch now).
int feed_prepare(pt_handle h, pt::FrameArgs &a, int si, hipStream_t st, int firstFrame, int n, int *word)
{
    *word = -1;
    const int w = h->feedWordNext % pt_renderer::kFeedHostWords;
    if (!h->feedWordBusy[w]) PT_HIP(h, hipEventCreateWithFlags(&h->feedWordBusy[w], hipEventDisableTiming));
    else if (hipEventQuery(h->feedWordBusy[w]) != hipSuccess) { // (the launch that read this word eight fed launches ago still runs)
        (void)hipGetLastError();
        return PT_OK;
    }
    if (h->feedCountersStale) {
        // an abandoned launch left the per-frame counters short of the running totals: start them again from zero — only with nothing of
        // the handle in flight that counts or compares them (else: no fed launch yet; the next join resets them as well)
        if (gpu_busy(h) || hipStreamQuery(h->copyStream) != hipSuccess) {
            (void)hipGetLastError();
            return PT_OK;
        }
        PT_HIP(h, hipMemsetAsync(h->dFeedDone, 0, (size_t)16 * pt::kFeedDoneStride * sizeof(unsigned long long), st));
        std::memset(h->feedDoneBase, 0, sizeof h->feedDoneBase);
        h->feedCountersStale = false;
    }
    h->feedWordNext++;
    pt_renderer::FeedState &f = h->feed;
    f.published = n;
    f.slots16 = 0;
    f.pendingSlot = -1;
    f.display = h->snapshotTarget != nullptr;
    __atomic_store_n(&h->hostFeedDone[w], 0u, __ATOMIC_RELEASE);
    __atomic_store_n(&h->hostFeed[w], ((unsigned int)n << 16), __ATOMIC_RELEASE);
    a.feedHostDone = h->devFeedHostDone + w;
    std::memcpy(a.feedBase, h->feedDoneBase[si], sizeof a.feedBase);
    a.feedPixels = (unsigned long long)h->tilePixels();
    unsigned int *const bcast = h->dFeedDev + (size_t)si * pt::kFeedBcastSlots * pt::kFeedBcastStride; // (one set of broadcast slots per launch stream)
    PT_HIP(h, hipMemsetAsync(bcast, 0, (size_t)pt::kFeedBcastSlots * pt::kFeedBcastStride * sizeof(unsigned int), st)); // (behind the previous launch of this stream)
    a.batchFrames = pt::kFeedCapacity;
    a.keepTags = 1; // (which frame is the last is not known when the launch starts: the host restores alpha = 1 before anything observes the image)
    const int wg = pt::tuning().feedWorkgroupsPerCU;
    a.variant = 10 + (wg >= 1 && wg <= 6 ? wg : 6) - 1;
    a.feedHost = h->devFeedHost + w;
    a.feedBcast = bcast;
    a.feedDone = h->dFeedDone + (size_t)8 * pt::kFeedDoneStride * si;
    const long idleUs = pt::tuning().feedIdleUs;
    a.feedIdleTicks = (unsigned int)((double)(idleUs < 1 ? 1 : (idleUs > 10000000 ? 10000000 : idleUs)) * h->wallClockKhz / 1000.0); // (ticks of the constant-rate counter: 100 MHz on gfx950)
    a.displayImages[0] = a.displayImages[1] = a.displayImages[2] = nullptr;
    a.displayPrev = 0;
    a.displayOn = f.display && pt::tuning().feedDisplay != 2 ? 1 : 0; // (feed_display = 2: debug — the protocol without the tone map)
    if (f.display)
        for (int i = 0; i < PT_PRESENT_SLOTS; i++) a.displayImages[i] = (uchar4 *)h->slots[i].boundDev; // (null: an unbound slot is never entered into the feed word)
    a.snapshot = nullptr;
    f.hostWord = w;
    *word = w;
    return PT_OK;
}

// pt_render with a fed launch open: publish the newest frame (h->frame - 1) into it.  False = it could not take the frame (it has been
// closed: the caller goes on as if there had been none).
bool feed_try_publish(pt_handle h, bool presentsEveryFrame)
{
    pt_renderer::FeedState &f = h->feed;
    const int frame = h->frame - 1;
    const bool ok = f.open && f.published < pt::kFeedCapacity && frame == f.firstFrame + f.published && h->pendingFrames == 1 &&
                    f.display == presentsEveryFrame && !(h->hostErrWord && *(volatile unsigned int *)h->hostErrWord);
    if (!ok) {
        FEED_LOG("publish refused: open %d published %d frame %d first %d pending %d display %d/%d flag %u", (int)f.open, f.published, frame, f.firstFrame, h->pendingFrames,
                 (int)f.display, (int)presentsEveryFrame, h->hostErrWord ? *(volatile unsigned int *)h->hostErrWord : 0u);
        ptimpl::feed_close(h);
        return false;
    }
    // (a launch opened right after an input change runs without cached tile masks: once the inputs have been left alone, let the next
    // launch build them — one launch boundary for ~7 % on every frame after it)
    if (!f.withMasks && pt::tuning().tileMasks != 0 && !h->tileMasksValid && h->launchesSinceInputChange > 2 && f.published >= 3 && f.pendingSlot < 0) {
        ptimpl::feed_close(h);
        return false;
    }
    h->launchesSinceInputChange++;
    const int j = f.published; // this frame's index in the launch
    f.slots16 &= ~(3u << (2 * (j & 7))); // (the entry of frame j - 8 becomes frame j's: not shown so far)
    f.published++;
    __atomic_store_n(&h->hostFeed[f.hostWord], ptimpl::feed_word(f, false), __ATOMIC_RELEASE);
    h->pendingFrames = 0;
    h->lastTag = 2.0f + (float)(frame & pt::kFrameTagMask);
    h->snapFrame = -1;
    h->statPublishes++;
    FEED_LOG("publish frame %d: word %d = %08x pendingSlot %d", frame, f.hostWord, ptimpl::feed_word(f, false), f.pendingSlot);
    if (f.pendingSlot >= 0) { // the frame before this one was presented: its image is complete when this frame is (the launch's monitor says when)
        h->slots[f.pendingSlot].fusedWaiting = false;
        f.pendingSlot = -1;
    }
    return true;
}

// ---- step 6a: chained launch — alternate between the main stream and the chain stream; the pixels' alpha tags order it behind the
// previous launch (which may still be draining on the other stream), nothing else does
int launch_chained(pt_handle h, pt::FrameArgs &a, int firstFrame, int n, long waitUs, bool lastOfFlush)
{
    a.y0 = h->y0;
    a.rows = h->rows;
    a.accum = h->accum();
    a.audit = h->dAudit;
    a.snapshot = h->snapshotTarget;
    a.tilesY = (h->rows + 7) / 8;
    a.keepTags = (h->flushFinal && lastOfFlush) ? 0 : 1;
    if (!lastOfFlush) a.snapshot = nullptr; // (the snapshot shows the flush's LAST frame)
    if (int rc = use_tile_masks(h, a)) return rc;
    const bool resident = wait_for_residency(h, a, waitUs);
    int si = resident ? (h->lastStreamIdx ^ 1) : h->lastStreamIdx;
    if (h->chainBroken) {
        // something else happened since the last tagged launch (an upload, a read, a striped frame ...): it was joined into
        // the main stream; start there again, and let the chain stream see those inputs before its next launch
        if (int rc = join_stripes(h)) return rc;
        si = 0;
        PT_HIP(h, hipEventRecord(h->inputsReady, h->stream));
        h->chainNeedsInputs = true;
    }
    a.chainTag = h->tagsLive ? h->lastTag : 0.0f;
    hipStream_t st = h->stream;
    if (si == 1) {
        // the chain stream IS the helper stream of stripe 1: the library keeps at most three streams busy (main, this one, copy) —
        // HIP multiplexes streams onto 4 hardware queues, and two of the library's streams that share a queue execute in order
        if (int rc = ptimpl::ensure_stripe(h, 1)) return rc;
        st = h->stripeStream[1];
        if (h->chainNeedsInputs) {
            PT_HIP(h, hipStreamWaitEvent(st, h->inputsReady, 0));
            h->chainNeedsInputs = false;
        }
    }
    a.startedFlags = h->devStarted;
    arm_handover(h, a);
    a.queue = h->dQueue + (si == 1 ? ptimpl::kChainQueueWord : 0); // each launch stream draws tickets from its own counter
    a.queueBase = si == 1 ? h->chainQueueBase : h->stripeQueueBase[0];
    unsigned int tickets = 0;
    int workgroups = 0;
    // a frame-fed launch when the host is not far ahead (few frames, not the flush of a call that joins next); else — or when this
    // kernel configuration has no fed instantiation — the classic launch of exactly these n frames
    bool fed = false;
    if (feed_wanted(h, a, n, lastOfFlush)) {
        const pt::FrameArgs classic = a;
        int word = -1;
        if (int rc = feed_prepare(h, a, si, st, firstFrame, n, &word)) return rc;
        if (word >= 0) {
            fed = true;
            const hipError_t e = pt::launch_integrate(a, st, &tickets, &workgroups, &fed);
            if (e != hipSuccess && e != hipErrorNotSupported) return hip_fail(h, e, "launch_integrate (frame-fed)");
            if (e != hipSuccess) {
                fed = false;
                h->feedHoldoff = 256; // (this scene's kernel has no fed instantiation: do not prepare one per launch)
            }
            (void)hipGetLastError();
        }
        if (!fed) a = classic;
    }
    if (!fed) {
        if (a.snapshot && h->snapReadPending[h->snapshotIndex]) PT_HIP(h, hipStreamWaitEvent(st, h->snapRead[h->snapshotIndex], 0));
        PT_HIP(h, pt::launch_integrate(a, st, &tickets, &workgroups));
    }
    (si == 1 ? h->chainQueueBase : h->stripeQueueBase[0]) += tickets;
    h->lastWorkgroups = workgroups;
    h->lastStreamIdx = si;
    hipEvent_t done = ptimpl::next_launch_event(h);
    if (!done) return fail(h, PT_E_HIP, "hipEventCreate failed");
    PT_HIP(h, hipEventRecord(done, st));
    remember_launch(h, a, done);
    if (si == 1) {
        h->chainDone = done;
        h->chainInFlight = h->chainPending = true;
    } else {
        h->mainDone = done;
        h->mainInFlight = true;
    }
    if (a.snapshot) h->snapLaunches.push_back({st, done, 0, h->tilePixels()});
    if (fed) { // pt_render publishes the frames that follow into this launch (feed_try_publish) until something closes it
        pt_renderer::FeedState &f = h->feed;
        PT_HIP(h, hipEventRecord(h->feedWordBusy[f.hostWord], st));
        f.open = true;
        f.firstFrame = firstFrame;
        f.published = n;
        f.streamIdx = si;
        f.seq = a.launchSeq;
        f.workgroups = workgroups;
        f.queueChunk = a.queueChunk;
        f.tilesFrame = (long long)a.tilesX * a.tilesY;
        f.pixelsPerFrame = (unsigned long long)h->tilePixels();
        f.withMasks = a.tileMasks != nullptr;
        std::memcpy(f.base, h->feedDoneBase[si], sizeof f.base);
        h->statFeedOpens++;
        FEED_LOG("open: first %d n %d stream %d word %d display %d wg %d", firstFrame, n, si, f.hostWord, (int)f.display, workgroups);
    }
    h->chainBroken = false; // (join_stripes above set it; this launch re-opens the chain)
    h->mainDirty = true;    // a striped frame that follows must order its helper stripe behind this launch
    h->tagsLive = a.keepTags != 0;
    h->lastTag = 2.0f + (float)((firstFrame + n - 1) & pt::kFrameTagMask); // pt::frame_tag of the launch's last frame
    return PT_OK;
}

// ---- step 6b: one kernel on the main stream (A/B variants, caller-owned streams)
int launch_single_stream(pt_handle h, pt::FrameArgs &a, const LaunchMode &m)
{
    if (int rc = join_stripes(h)) return rc;
    h->tagsLive = false; // (the plain path stores alpha = 1 for every pixel; a tagged launch of an A/B variant stores 1 last)
    a.y0 = h->y0;
    a.rows = h->rows;
    a.accum = h->accum();
    a.audit = h->dAudit;
    a.snapshot = m.kernelVariant >= 10 ? h->snapshotTarget : nullptr; // (only the persistent kernels write snapshots)
    if (a.snapshot && h->snapReadPending[h->snapshotIndex]) PT_HIP(h, hipStreamWaitEvent(h->stream, h->snapRead[h->snapshotIndex], 0));
    a.tilesY = (h->rows + 7) / 8;
    a.queue = h->dQueue;
    a.queueBase = h->stripeQueueBase[0];
    if (m.tagged) arm_handover(h, a); // (a tagged launch of an A/B variant: alone on the main stream, but its frames still hand pixels over)
    unsigned int tickets = 0;
    PT_HIP(h, pt::launch_integrate(a, h->stream, &tickets));
    h->stripeQueueBase[0] += tickets; // unsigned wrap-around is fine: the kernel subtracts queueBase modulo 2^32
    hipEvent_t done = ptimpl::next_launch_event(h);
    if (!done) return fail(h, PT_E_HIP, "hipEventCreate failed");
    PT_HIP(h, hipEventRecord(done, h->stream));
    if (m.tagged) remember_launch(h, a, done);
    h->mainDone = done;
    h->mainInFlight = true;
    if (a.snapshot) h->snapLaunches.push_back({h->stream, h->mainDone, 0, h->tilePixels()});
    return PT_OK;
}

// ---- step 6c: one frame as row stripes, each on its own stream.  Inputs uploaded on the main stream (scene, environment, clears) must
// be visible to the stripe streams; a stripe's frame f+1 follows its own frame f in stream order, which is the only dependency between
// frames (the helpers only wait when something other than stripe 0's own frames went onto the main stream since the last striped frame:
// stripe 0 runs there, and the helper stripes must not wait for ITS previous frame — overlapping one stripe's drain with the other's
// main phase is the point of the stripes)
int launch_striped(pt_handle h, pt::FrameArgs &a, const LaunchMode &m)
{
    if (h->chainPending || h->tagsLive) { // a chained launch may still run on the chain stream: the stripes read its pixels plainly
        if (int rc = join_stripes(h)) return rc;
        h->tagsLive = false; // every pixel gets alpha = 1 from this frame
    }
    const bool orderHelpers = h->mainDirty;
    if (orderHelpers) PT_HIP(h, hipEventRecord(h->inputsReady, h->stream));
    h->mainDirty = false;
    for (int j = 0; j < m.stripes; j++) {
        int r0 = (int)((long long)h->rows * j / m.stripes), r1 = (int)((long long)h->rows * (j + 1) / m.stripes);
        r0 &= ~7; // stripe boundaries on 8-row tile boundaries (the last stripe takes the ragged remainder)
        if (j + 1 < m.stripes) r1 &= ~7;
        if (r1 <= r0) continue;
        a.y0 = h->y0 + r0;
        a.localRow0 = r0;
        a.rows = r1 - r0;
        a.accum = h->accum() + (size_t)r0 * h->width;
        a.audit = h->dAudit ? h->dAudit + (size_t)r0 * h->width : nullptr;
        a.snapshot = h->snapshotTarget ? h->snapshotTarget + (size_t)r0 * h->width : nullptr;
        a.tilesY = (a.rows + 7) / 8;
        a.queue = h->dQueue + 16 * j;
        a.queueBase = h->stripeQueueBase[j];
        if (int rc = ptimpl::ensure_stripe(h, j)) return rc;
        h->stripeRow0[j] = r0;
        h->stripeRows[j] = r1 - r0;
        hipStream_t st = ptimpl::stripe_stream(h, j);
        if (j > 0 && orderHelpers) PT_HIP(h, hipStreamWaitEvent(st, h->inputsReady, 0));
        if (a.snapshot && h->snapReadPending[h->snapshotIndex]) PT_HIP(h, hipStreamWaitEvent(st, h->snapRead[h->snapshotIndex], 0));
        unsigned int tickets = 0;
        PT_HIP(h, pt::launch_integrate(a, st, &tickets));
        h->stripeQueueBase[j] += tickets;
        PT_HIP(h, hipEventRecord(h->stripeDone[j], st));
        if (a.snapshot) h->snapLaunches.push_back({st, h->stripeDone[j], (size_t)r0 * h->width, (size_t)(r1 - r0) * h->width});
        h->stripePending[j] = true;
        h->stripeInFlight[j] = true;
    }
    return PT_OK;
}

int launch_frames(pt_handle h, int firstFrame, int n, long waitUs, bool lastOfFlush)
{
    if (int rc = bind_device(h)) return rc;
    h->statLaunches++;
    if (!h->snapshotTarget) h->snapFrame = -1; // frames rendered without a present snapshot: an older snapshot no longer shows the image
    pt::FrameArgs a;
    if (int rc = fill_frame_args(h, a, firstFrame, n)) return rc;
    const LaunchMode m = choose_launch_mode(h, a, n);
    if (m.tagged && h->hostErrWord && *(volatile unsigned int *)h->hostErrWord) {
        // a launch of this handle was abandoned: put its repair (and that of everything launched since) behind a join before anything new
        // builds on those frames
        if (int rc = join_stripes(h)) return rc;
        if (*(volatile unsigned int *)h->hostErrWord) ptimpl::note_abandonment(h);
    }
    if (m.tagged && m.chainable) return launch_chained(h, a, firstFrame, n, waitUs, lastOfFlush);
    if (m.stripes == 1) return launch_single_stream(h, a, m);
    return launch_striped(h, a, m);
}

// Is an integrator launch of this handle still running (or queued) on the GPU?  Cheap event queries, no waiting.
bool gpu_busy(pt_handle h)
{
    bool busy = false;
    if (h->mainInFlight) {
        if (hipEventQuery(h->mainDone) == hipErrorNotReady) busy = true;
        else h->mainInFlight = false;
    }
    if (h->chainInFlight) {
        if (hipEventQuery(h->chainDone) == hipErrorNotReady) busy = true;
        else h->chainInFlight = false;
    }
    for (int j = 0; j < ptimpl::kMaxStripes; j++) {
        if (!h->stripeInFlight[j]) continue;
        if (hipEventQuery(h->stripeDone[j]) == hipErrorNotReady) busy = true;
        else h->stripeInFlight[j] = false;
    }
    return busy;
}

int ensure_rgba8(pt_handle h)
{
    size_t need = h->tilePixels();
    if (need > h->rgba8Capacity) {
        if (h->dRgba8) {
            PT_HIP(h, hipStreamSynchronize(h->stream));
            PT_HIP(h, hipFree(h->dRgba8));
        }
        h->dRgba8 = nullptr;
        h->rgba8Capacity = 0;
        PT_HIP(h, hipMalloc(&h->dRgba8, need * 4));
        h->rgba8Capacity = need;
    }
    return PT_OK;
}

} // namespace

namespace ptimpl {

// frames one launch may hold for this handle right now (pt_set_frame_batch; automatic: 64, or 256 on a small share)
int batch_limit(pt_handle h)
{
    // (a GPU that owns a small share of the image — fewer than 12,000 tiles per frame, e.g. 1/8 of 1080p — pipelines up to 256 frames
    // per launch when the batch size was left at its default: every launch boundary costs ~0.1 ms of drain + ramp, 7 % of a 64-frame
    // launch there; spp > 1 keeps 64, its kernels carry the frame index in 7 bits)
    if (!h->maxBatchExplicit && h->spp == 1 && (long long)((h->width + 7) / 8) * ((h->rows + 7) / 8) < 12000) return 256;
    return h->maxBatch < 1 ? 1 : h->maxBatch;
}

// Would a tagged launch issued now start BESIDE its predecessor (the predecessor is resident), or on an idle chain?  (Otherwise it would
// have to queue behind it on the same stream, or wait: pt_render keeps the frames pending instead.)
bool launch_ready(pt_handle h)
{
    if (h->chainBroken || h->lastWorkgroups <= 0 || h->lastWorkgroups > kStartedWords || h->overlapHoldoff > 0 || pt::tuning().serialLaunches != 0) return true; // (no chaining decision to wait for)
    for (int i = 0; i < h->lastWorkgroups; i++)
        if (((volatile unsigned int *)h->hostStarted)[i] != h->launchSeq) return false;
    return true;
}

// Launch the pending frames, oldest first, in launches of at most batch_limit() frames.  waitUs: how long EACH launch may wait for its
// predecessor to become resident (pt_render passes 0 or a small bound and relies on launch_ready(); blocking entry points pass the
// tuning knob chain_wait_us).  maxLaunches > 0: stop after that many launches (the rest stays pending).
int flush_frames_bounded(pt_handle h, long waitUs, int maxLaunches)
{
    feed_close(h); // (whoever flushes is about to launch, or to change an input: the open fed launch takes no more frames)
    const int limit = batch_limit(h);
    int launches = 0;
    while (h->pendingFrames > 0 && (maxLaunches <= 0 || launches < maxLaunches)) {
        const int pending = h->pendingFrames;
        // (only the default kernel pipelines: any other configuration never has more than one frame pending)
        const int n = pending < limit ? pending : limit;
        const int first = h->frame - pending;
        h->pendingFrames = 0; // (launch_frames calls join_stripes, which flushes what is pending: nothing, while this launch is built)
        const int rc = launch_frames(h, first, n, waitUs, pending == n);
        h->pendingFrames = pending - n;
        if (rc) return rc;
        launches++;
    }
    return PT_OK;
}

int flush_frames(pt_handle h) { return flush_frames_bounded(h, pt::tuning().chainWaitUs, 0); }

// PostProcessing/fragment.glsl:17-26 over this handle's rows into `dst`, ordered behind every frame rendered so far
int tone_map_into(pt_handle h, void *dst)
{
    if (int rc = join_stripes(h)) return rc;
    PT_HIP(h, pt::launch_postprocess(h->accum(), dst, h->tilePixels(), h->stream));
    return PT_OK;
}

int ensure_slot_events(pt_handle h, int slot)
{
    PresentSlot &s = h->slots[slot];
    if (!s.toneMapped) PT_HIP(h, hipEventCreateWithFlags(&s.toneMapped, hipEventDisableTiming));
    if (!s.copied) PT_HIP(h, hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    return PT_OK;
}

int ensure_slot_device(pt_handle h, int slot, size_t pixels)
{
    PresentSlot &s = h->slots[slot];
    if (pixels > s.devPixels) {
        if (s.dRgba8) {
            if (s.copied && s.inFlight) PT_HIP(h, hipEventSynchronize(s.copied)); // a copy may still read the old buffer
            PT_HIP(h, hipStreamSynchronize(h->stream));
            PT_HIP(h, hipFree(s.dRgba8));
        }
        s.dRgba8 = nullptr;
        s.devPixels = 0;
        PT_HIP(h, hipMalloc(&s.dRgba8, pixels * 4));
        s.devPixels = pixels;
    }
    return PT_OK;
}

int ensure_slot_host(pt_handle h, int slot, size_t pixels)
{
    PresentSlot &s = h->slots[slot];
    if (pixels > s.hostPixels) {
        if (s.host) {
            if (s.copied && s.inFlight) PT_HIP(h, hipEventSynchronize(s.copied));
            PT_HIP(h, hipHostFree(s.host));
        }
        s.host = nullptr;
        s.hostPixels = 0;
        s.valid = false;
        PT_HIP(h, hipHostMalloc((void **)&s.host, pixels * 4, hipHostMallocDefault));
        s.hostPixels = pixels;
    }
    return PT_OK;
}

// Launch the pending frames with a present snapshot attached: their launch stores its last frame's pixels into snapshot buffer
// h->snapNext while it resolves them (FrameArgs::snapshot).  On return h->snapLaunches lists the launches that write it (empty if
// a kernel variant without snapshot support rendered the frames) and h->snapFrame is the frame count the snapshot shows.
static int ensure_snapshot_buffer(pt_handle h, int k, size_t pixels)
{
    if (pixels > h->snapCapacity[k]) {
        // (a launch queued earlier may still write the old buffer, a tone map may still read it: drain the handle's streams first)
        feed_close(h);
        for (int j = 0; j < ptimpl::kMaxStripes; j++)
            if (h->stripeStream[j]) PT_HIP(h, hipStreamSynchronize(h->stripeStream[j]));
        PT_HIP(h, hipStreamSynchronize(h->stream));
        PT_HIP(h, hipStreamSynchronize(h->copyStream));
        if (h->snapReadPending[k]) PT_HIP(h, hipEventSynchronize(h->snapRead[k]));
        h->snapReadPending[k] = false;
        if (h->dSnap[k]) PT_HIP(h, hipFree(h->dSnap[k]));
        h->dSnap[k] = nullptr;
        h->snapCapacity[k] = 0;
        PT_HIP(h, hipMalloc((void **)&h->dSnap[k], pixels * sizeof(float4)));
        h->snapCapacity[k] = pixels;
    }
    if (!h->snapRead[k]) PT_HIP(h, hipEventCreateWithFlags(&h->snapRead[k], hipEventDisableTiming));
    return PT_OK;
}

int flush_with_snapshot(pt_handle h, long waitUs)
{
    const int k = h->snapNext;
    const size_t pixels = h->tilePixels();
    if (int rc = ensure_snapshot_buffer(h, k, pixels)) return rc;
    h->snapshotTarget = h->dSnap[k];
    h->snapshotIndex = k;
    h->snapGeneration[k]++;
    h->snapLaunches.clear();
    h->snapFrame = -1;
    const int rc = flush_frames_bounded(h, waitUs, 0);
    h->snapshotTarget = nullptr;
    if (rc) return rc;
    if (!h->snapLaunches.empty()) {
        h->snapFrame = h->frame;
        // every snapshot launch takes the next buffer, presented or not: two launches that may run beside each other must never
        // write the same snapshot (their plain stores to one pixel are not ordered by the tags)
        h->snapNext = (k + 1) % pt_renderer::kSnapshots;
    }
    return PT_OK;
}

// Wait (on the host) until the previous present into slot `s` has left it: an event for the classic paths; for a FUSED present
// (frame-fed launch) the launch's monitor wavefront reports complete frames in a host-mapped word — a fused present whose successor frame
// was never published is first turned into a classic one (close + join: resolve_fused_orphan).  Returns with the slot idle; *abandoned
// is set when a fused present's launch gave up (the caller re-does the image behind a join).
int wait_slot(pt_handle h, PresentSlot &s, bool *abandoned)
{
    if (abandoned) *abandoned = false;
    if (!s.inFlight) return PT_OK;
    if (s.fedPresent && s.fusedWaiting) {
        feed_close(h);
        if (int rc = join_stripes(h)) return rc;
        if (s.fusedWaiting) return fail(h, PT_E_HIP, "present: a fused present was left without its image");
    }
    if (s.fedPresent) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned long spins = 0;; spins++) {
            if (__atomic_load_n(&h->hostFeedDone[s.fusedWord], __ATOMIC_ACQUIRE) >= s.fusedNeed) break;
            if ((h->hostErrWord && *(volatile unsigned int *)h->hostErrWord) || h->abandonEpoch != s.abandonEpoch) {
                // its launch was abandoned (a contended device, or it ended idle) — now, or since the present (the host has already noted it)
                if (abandoned) *abandoned = true;
                break;
            }
            if ((spins & 63) == 63) {
                const auto waited = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                if (waited > 2000000) { // (2 s: nothing of this library takes that long — treat the launch as lost, the join below sorts it out)
                    if (abandoned) *abandoned = true;
                    break;
                }
                if (waited > 100) std::this_thread::yield();
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
    } else {
        PT_HIP(h, hipEventSynchronize(s.copied));
    }
    return PT_OK;
}

void free_slots(pt_handle h)
{
    for (PresentSlot &s : h->slots) {
        if (s.dRgba8) (void)hipFree(s.dRgba8);
        if (s.host) (void)hipHostFree(s.host);
        if (s.toneMapped) (void)hipEventDestroy(s.toneMapped);
        if (s.copied) (void)hipEventDestroy(s.copied);
        s = PresentSlot();
    }
}

} // namespace ptimpl

extern "C" {

PT_API int pt_render(pt_handle h, int *out_total_samples)
{
    PT_CHECK_HANDLE(h);
    if (h->isGroup()) {
        int total = 0;
        for (pt_handle part : h->parts) {
            int rc = pt_render(part, &total);
            if (rc != PT_OK) return fail(h, rc, part->error);
        }
        h->frame = h->parts[0]->frame;
        if (out_total_samples) *out_total_samples = total;
        return PT_OK;
    }
    if (!h->dEnv) return fail(h, PT_E_NO_ENVIRONMENT, "pt_render called before pt_set_environment / pt_atmosphere_render");
    if (h->boundAccum && h->boundBytes < h->tilePixels() * sizeof(float4))
        return fail(h, PT_E_BAD_ARGUMENT, "bound result buffer is smaller than the tile");
    // Only the default kernel pipelines frames; the A/B variants launch at once.  On a caller-owned stream nothing is
    // deferred: the contract there is that the frame is enqueued on that stream when pt_render returns.
    // A host that presents EVERY frame (MainWindow.cs:49-56) gains nothing from holding a frame back — the present that
    // follows launches it anyway — and loses if it waits for a present slot in between (the GPU would run dry): launch at once.
    // (only the first pt_render after a present is treated that way: a host that goes on rendering without presenting gets
    // its frames pipelined again)
    const bool presentsEveryFrame = h->presentCadence == 1 && h->rendersSincePresent == 0;
    const bool batchable = h->variant == 0 && h->maxBatch > 1 && !ptimpl::timeline_blocks_pipelining(h) && !h->externalStream() && !presentsEveryFrame;
    h->rendersSincePresent++;
    h->pendingFrames++;
    h->frame++; // PathTracer.cs:117 post-increment
    if (out_total_samples) *out_total_samples = h->frame * h->spp; // PathTracer.cs:112
    // Round 6: a frame-fed launch is open — this frame is PUBLISHED into it (one store to a host-mapped word; the resident wavefronts
    // start it when they run out of earlier work) instead of being launched.  If it cannot take the frame it is closed, and the frame
    // goes the usual way below.
    if (h->feed.open && feed_try_publish(h, presentsEveryFrame)) return PT_OK;
    // Round 3: a host that presents every frame through pt_present_rgba8_async gets this frame launched with a present snapshot
    // attached: the launch stores the frame's pixels a second time while it resolves them (FrameArgs::snapshot), the present that
    // follows tone-maps that copy, and the tone map never stands between two frames.  (Launched NOW, not by the present: a host
    // that first waits for a present slot and then presents must find the GPU busy.  If no present follows, the copy is ignored.)
    if (presentsEveryFrame && (h->variant == 0 || h->variant >= 10) && !h->externalStream() && !ptimpl::timeline_blocks_pipelining(h))
        return ptimpl::flush_with_snapshot(h, pt::tuning().renderWaitUs);
    // Frames are only held back while the GPU still has integrator work of this handle in flight: deferring can then
    // never idle the device, and a host that leaves time between its frames gets every frame launched at once.
    if (!batchable) return flush_frames(h);
    if (!gpu_busy(h)) return ptimpl::flush_frames_bounded(h, 0, 0); // (idle device: nothing to wait for)
    const int limit = ptimpl::batch_limit(h);
    if (h->pendingFrames < limit) return PT_OK;
    // A full batch.  Round 5: pt_render never sits in the back-pressure wait (it was up to chain_wait_us = 60 ms): the batch is launched
    // when it can start beside its predecessor (that one is resident: the launch before it has left the machine); until then the
    // frames simply stay pending — the GPU has two launches' worth of work queued, nothing idles — and whichever call comes next
    // looks again.  Only a host that runs more than 16 launches ahead is held, for at most 2 ms per call, and then queues the batch
    // behind its predecessor.
    // (a small share — fewer than 12,000 tiles — never waited: its launches go behind each other unless the predecessor happens to be
    // resident, which is what it measures best with: tools/emulate_strong.py)
    const bool bigShare = (long long)((h->width + 7) / 8) * ((h->rows + 7) / 8) >= 12000;
    // (a limit the host set itself bounds latency AND deferral: such a batch is launched at the limit, behind its predecessor if need be)
    if (!bigShare || h->maxBatchExplicit || ptimpl::launch_ready(h)) return ptimpl::flush_frames_bounded(h, 0, 1);
    if (h->pendingFrames >= 16 * limit) {
        // a host that runs this far ahead is paced: the call waits (at most render_wait_us = 2 ms) for the moment the batch can start beside
        // its predecessor and launches it then; if that moment does not come in time the frames simply stay pending — never a launch BEHIND
        // the predecessor from here (that order costs a drain + ramp per launch: 1.4 % at 1080p; measured as a 2.4 % lower steady rate
        // when this path still did it)
        const long limitUs = pt::tuning().renderWaitUs;
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            if (ptimpl::launch_ready(h)) return ptimpl::flush_frames_bounded(h, 0, 1);
            const auto waited = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            if (waited >= limitUs) break;
            std::this_thread::sleep_for(std::chrono::microseconds(waited > 200 ? 50 : 5));
        }
    }
    return PT_OK;
}

PT_API int pt_read_result(pt_handle h, float *dst, size_t row_pitch_bytes)
{
    PT_CHECK_HANDLE(h);
    if (!dst) return fail(h, PT_E_BAD_ARGUMENT, "dst == NULL");
    size_t rowBytes = (size_t)h->width * 16;
    if (row_pitch_bytes == 0) row_pitch_bytes = rowBytes;
    if (row_pitch_bytes < rowBytes) return fail(h, PT_E_BAD_ARGUMENT, "row pitch smaller than a row");
    if (h->isGroup()) return ptimpl::group_read_result(h, dst, row_pitch_bytes);
    if (int rc = bind_device(h)) return rc;
    // (the frames first, then the hand-over check — a repair pass in the rare case — then the copy of a settled image)
    if (int rc = ptimpl::fix_alpha(h, false)) return rc;
    PT_HIP(h, hipStreamSynchronize(h->stream));
    if (int rc = ptimpl::settle_handover(h)) return rc;
    PT_HIP(h, hipMemcpy2DAsync(dst, row_pitch_bytes, h->accum(), rowBytes, rowBytes, (size_t)h->rows,
                               hipMemcpyDeviceToHost, h->stream));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return PT_OK;
}

PT_API int pt_write_result(pt_handle h, const float *src, size_t row_pitch_bytes, int frame_index)
{
    PT_CHECK_HANDLE(h);
    if (!src || frame_index < 0) return fail(h, PT_E_BAD_ARGUMENT, "src == NULL or negative frame index");
    size_t rowBytes = (size_t)h->width * 16;
    if (row_pitch_bytes == 0) row_pitch_bytes = rowBytes;
    if (row_pitch_bytes < rowBytes) return fail(h, PT_E_BAD_ARGUMENT, "row pitch smaller than a row");
    if (h->isGroup()) return ptimpl::group_write_result(h, src, row_pitch_bytes, frame_index);
    if (int rc = bind_device(h)) return rc;
    if (int rc = join_stripes(h)) return rc;
    PT_HIP(h, hipMemcpy2DAsync(h->accum(), rowBytes, src, row_pitch_bytes, rowBytes, (size_t)h->rows,
                               hipMemcpyHostToDevice, h->stream));
    // alpha is the reference's constant 1 (compute.glsl:129) whatever the file held: inside a pipelined launch alpha
    // carries the frame tag, so a restored 2.0 must never reach the kernel
    PT_HIP(h, pt::launch_set_alpha(h->accum(), h->tilePixels(), h->stream));
    h->tagsLive = false;
    if (int rc = audit_forget(h)) return rc;
    PT_HIP(h, hipStreamSynchronize(h->stream));
    h->frame = frame_index;
    return PT_OK;
}

PT_API int pt_present_rgba8(pt_handle h, uint8_t *dst, size_t row_pitch_bytes)
{
    PT_CHECK_HANDLE(h);
    if (!dst) return fail(h, PT_E_BAD_ARGUMENT, "dst == NULL");
    size_t rowBytes = (size_t)h->width * 4;
    if (row_pitch_bytes == 0) row_pitch_bytes = rowBytes;
    if (row_pitch_bytes < rowBytes) return fail(h, PT_E_BAD_ARGUMENT, "row pitch smaller than a row");
    if (h->isGroup()) return ptimpl::group_present_rgba8(h, dst, row_pitch_bytes);
    if (int rc = bind_device(h)) return rc;
    if (int rc = ensure_rgba8(h)) return rc;
    if (int rc = ptimpl::tone_map_into(h, h->dRgba8)) return rc; // (behind a join: the hand-over repair passes are in front of the tone map)
    PT_HIP(h, hipMemcpy2DAsync(dst, row_pitch_bytes, h->dRgba8, rowBytes, rowBytes, (size_t)h->rows, hipMemcpyDeviceToHost,
                               h->stream));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return PT_OK;
}

PT_API int pt_postprocess_device(pt_handle h, void **out_device_ptr, size_t *out_bytes)
{
    PT_CHECK_HANDLE(h);
    if (h->isGroup()) return fail(h, PT_E_BAD_ARGUMENT, "pt_postprocess_device is not available on a group handle");
    if (int rc = bind_device(h)) return rc;
    if (int rc = ensure_rgba8(h)) return rc;
    if (int rc = ptimpl::tone_map_into(h, h->dRgba8)) return rc;
    if (out_device_ptr) *out_device_ptr = h->dRgba8;
    if (out_bytes) *out_bytes = h->tilePixels() * 4;
    return PT_OK;
}

PT_API int pt_present_rgba8_async(pt_handle h, int slot)
{
    PT_CHECK_HANDLE(h);
    if (slot < 0 || slot >= PT_PRESENT_SLOTS) return fail(h, PT_E_BAD_ARGUMENT, "slot must be 0..PT_PRESENT_SLOTS-1");
    if (h->isGroup()) return ptimpl::group_present_async(h, slot);
    if (int rc = bind_device(h)) return rc;
    ptimpl::PresentSlot &s = h->slots[slot];
    const size_t pixels = h->tilePixels();
    h->presentCadence = h->rendersSincePresent;
    h->rendersSincePresent = 0;
    h->lastPresentBound = h->slots[slot].boundDev != nullptr;
    // (an open fed launch that does not show its frames itself takes no more of them: this present is about to flush and join — and may
    // allocate first: the launch must not sit waiting for frames meanwhile)
    if (h->feed.open && !h->feed.display) ptimpl::feed_close(h);
    if (int rc = ptimpl::ensure_slot_events(h, slot)) return rc;
    if (s.boundDev) { // the displayed image stays on the device (interop-style present): no slot images of the library's own
        if (s.boundBytes < pixels * 4) return fail(h, PT_E_BAD_ARGUMENT, "bound device image is smaller than rows*width*4 bytes");
    } else {
        if (int rc = ptimpl::ensure_slot_device(h, slot, pixels)) return rc;
        if (int rc = ptimpl::ensure_slot_host(h, slot, pixels)) return rc;
    }
    void *const image = s.boundDev ? s.boundDev : s.dRgba8;
    // ---- snapshot path: the launch of the frames shown writes its last frame into a snapshot buffer while it resolves the pixels;
    // the tone map follows that launch on its stream, the copy on the copy stream, and nothing waits for them: the next pt_render
    // chains its launch beside this one.
    // ---- frame-fed path (round 6), FUSED DISPLAY: the newest frame sits in the open launch.  Nothing is launched, flushed or tone-mapped
    // here: the slot is entered into the feed word, and the tile passes of the NEXT frame — which read every pixel of this one anyway —
    // tone-map it into the slot's image (FrameArgs::displayImages).  The gate that tells when the image is complete is enqueued when that
    // next frame is published (feed_try_publish); if none comes, whoever closes the launch tone-maps the classic way (resolve_fused_orphan).
    if (h->feed.open && h->feed.display && h->pendingFrames == 0 && h->feed.pendingSlot < 0 && h->frame == h->feed.firstFrame + h->feed.published &&
        !(h->hostErrWord && *(volatile unsigned int *)h->hostErrWord)) {
        pt_renderer::FeedState &f = h->feed;
        if (s.boundDev != nullptr) { // (the launch knows the slots' bound images: binding closes it)
            if (s.inFlight) { // (a host that reuses a slot before having waited for it: wait here)
                if (int rc = ptimpl::wait_slot(h, s, nullptr)) return rc;
                s.inFlight = false;
            }
            if (!f.open) return pt_present_rgba8_async(h, slot); // (the wait closed the launch: the classic way — presentCadence is set again, harmless)
            const int j = f.published - 1;
            f.slots16 = (f.slots16 & ~(3u << (2 * (j & 7)))) | ((unsigned int)(slot + 1) << (2 * (j & 7)));
            f.pendingSlot = slot; // (the word that publishes the next frame carries the entry)
            FEED_LOG("present frame %d into slot %d (fused, waits for the next frame)", h->frame, slot);
            s.inFlight = true;
            s.valid = false;
            s.frame = h->frame;
            s.rows = h->rows;
            s.width = h->width;
            s.snapSource = image; // (non-null: pt_present_wait looks at the abandon epoch)
            s.snapGeneration = 0;
            s.abandonEpoch = h->abandonEpoch;
            s.fedPresent = true;
            s.fusedWaiting = true;
            s.fusedWord = f.hostWord;
            s.fusedNeed = (unsigned int)(j + 2); // (frame j + 1 of the launch complete = its tile passes have tone-mapped frame j everywhere)
            return PT_OK;
        }
    }
    if (s.inFlight && s.fedPresent) { // (the slot's previous present was a fused one: no event to order behind — wait for it here)
        if (int rc = ptimpl::wait_slot(h, s, nullptr)) return rc;
        s.inFlight = false;
    }
    s.fedPresent = false;
    const bool snapshotCapable = (h->variant == 0 || h->variant >= 10) && !h->externalStream() && !ptimpl::timeline_blocks_pipelining(h);
    if (snapshotCapable && h->pendingFrames > 0)
        if (int rc = ptimpl::flush_with_snapshot(h, pt::tuning().renderWaitUs)) return rc;
    if (snapshotCapable && h->pendingFrames == 0 && h->snapFrame == h->frame && !h->snapLaunches.empty()) {
        // (the frames' launch — flushed just now, or by the pt_render before this call — wrote snapshot buffer snapshotIndex)
        {
            const int k = h->snapshotIndex;
            // The tone map of a launch's rows runs ON that launch's stream, right behind it (a few microseconds: the kernel raises its
            // wave priority, the next launch is already resident on the other stream).  Only the NEXT-BUT-ONE launch queues behind it.
            // Not on the copy stream: tone map + 150 us copy in one queue would bound the display rate; not on a stream of its own: a
            // fourth busy stream shares a hardware queue with a launch stream and would run behind the next launch.
            // (a slot the host reuses before its previous copy has landed: wait for that copy HERE, on the host — queued on the launch's
            // stream instead, the next-but-one integrator launch would stall behind a PCIe copy.  A host that calls pt_present_wait
            // on a slot before it presents into it again, as the header recommends, never waits here)
            if (s.inFlight && hipEventQuery(s.copied) != hipSuccess) PT_HIP(h, hipEventSynchronize(s.copied));
            (void)hipGetLastError(); // (hipErrorNotReady of the query is not an error)
            for (const pt_renderer::SnapLaunch &l : h->snapLaunches) {
                PT_HIP(h, pt::launch_postprocess(h->dSnap[k] + l.firstPixel, (char *)image + l.firstPixel * 4, l.pixels, l.stream));
                PT_HIP(h, hipEventRecord(l.done, l.stream)); // "launch done" now includes its tone map
                PT_HIP(h, hipStreamWaitEvent(h->copyStream, l.done, 0));
            }
            PT_HIP(h, hipEventRecord(h->snapRead[k], h->copyStream)); // (behind every tone map that read the snapshot)
            h->snapReadPending[k] = true;
            if (!s.boundDev) PT_HIP(h, hipMemcpyAsync(s.host, s.dRgba8, pixels * 4, hipMemcpyDeviceToHost, h->copyStream));
            PT_HIP(h, hipEventRecord(s.copied, h->copyStream));
            s.inFlight = true;
            s.valid = false;
            s.frame = h->frame;
            s.rows = h->rows;
            s.width = h->width;
            s.snapSource = h->dSnap[k];
            s.snapGeneration = h->snapGeneration[k];
            s.abandonEpoch = h->abandonEpoch;
            h->snapFrame = -1; // consumed
            h->snapLaunches.clear();
            return PT_OK;
        }
    }
    // (nothing pending and no fresh snapshot — the frames were launched by another call, or by a kernel variant that writes none:
    // present from the accumulation image)
    if (int rc = flush_frames(h)) return rc;
    bool striped = false;
    for (int j = 0; j < ptimpl::kMaxStripes; j++) striped = striped || h->stripePending[j];
    if (striped) {
        // The last frame was launched as row stripes on their own streams.  Tone-map every stripe's rows ON ITS stream:
        // stripe A's next frame then only waits for stripe A's own frame + tone map, never for the other stripe's drain
        // (joining into the main stream here would put a device-wide barrier between consecutive displayed frames).
        for (int j = 0; j < ptimpl::kMaxStripes; j++) {
            if (!h->stripePending[j]) continue;
            // the slot's previous copy must have left the device image before it is overwritten (device-side wait only)
            hipStream_t st = ptimpl::stripe_stream(h, j);
            if (s.inFlight) PT_HIP(h, hipStreamWaitEvent(st, s.copied, 0));
            const size_t first = (size_t)h->stripeRow0[j] * h->width, count = (size_t)h->stripeRows[j] * h->width;
            PT_HIP(h, pt::launch_postprocess(h->accum() + first, (char *)image + first * 4, count, st));
            PT_HIP(h, hipEventRecord(h->stripeDone[j], st)); // "stripe done" now includes its tone map
            PT_HIP(h, hipStreamWaitEvent(h->copyStream, h->stripeDone[j], 0));
        }
    } else {
        if (int rc = join_stripes(h)) return rc;
        if (s.inFlight) PT_HIP(h, hipStreamWaitEvent(h->stream, s.copied, 0));
        PT_HIP(h, pt::launch_postprocess(h->accum(), image, pixels, h->stream));
        PT_HIP(h, hipEventRecord(s.toneMapped, h->stream));
        // later frames only wait for the tone-map pass (stream order); the PCIe copy runs beside them on the copy stream
        PT_HIP(h, hipStreamWaitEvent(h->copyStream, s.toneMapped, 0));
    }
    if (!s.boundDev) PT_HIP(h, hipMemcpyAsync(s.host, s.dRgba8, pixels * 4, hipMemcpyDeviceToHost, h->copyStream));
    PT_HIP(h, hipEventRecord(s.copied, h->copyStream));
    s.inFlight = true;
    s.valid = false;
    s.frame = h->frame;
    s.rows = h->rows;
    s.width = h->width;
    s.snapSource = nullptr; // (tone-mapped from the accumulation image behind a join: the repair passes ran in front of it)
    return PT_OK;
}

PT_API int pt_present_bind_device_image(pt_handle h, int slot, void *device_rgba8, size_t bytes)
{
    PT_CHECK_HANDLE(h);
    if (slot < 0 || slot >= PT_PRESENT_SLOTS) return fail(h, PT_E_BAD_ARGUMENT, "slot must be 0..PT_PRESENT_SLOTS-1");
    if (h->isGroup()) return fail(h, PT_E_BAD_ARGUMENT, "pt_present_bind_device_image is not available on a group handle");
    if (device_rgba8 && bytes < h->tilePixels() * 4) return fail(h, PT_E_BAD_ARGUMENT, "device image smaller than rows*width*4 bytes");
    if (int rc = bind_device(h)) return rc;
    ptimpl::PresentSlot &s = h->slots[slot];
    if (h->feed.open || h->fusedOrphanSlot >= 0) { // (an open fed launch writes the slots' images as they were when it started)
        ptimpl::feed_close(h);
        if (int rc = join_stripes(h)) return rc;
    }
    if (s.inFlight) { // a present into the slot's previous image may still be running
        if (int rc = ptimpl::wait_slot(h, s, nullptr)) return rc;
        s.inFlight = false;
    }
    s.valid = false;
    s.fedPresent = false;
    s.boundDev = device_rgba8;
    s.boundBytes = device_rgba8 ? bytes : 0;
    return PT_OK;
}

PT_API int pt_present_wait(pt_handle h, int slot, const uint8_t **out_host_rgba8, size_t *out_row_pitch_bytes,
                           int *out_frame_index)
{
    PT_CHECK_HANDLE(h);
    if (slot < 0 || slot >= PT_PRESENT_SLOTS) return fail(h, PT_E_BAD_ARGUMENT, "slot must be 0..PT_PRESENT_SLOTS-1");
    ptimpl::PresentSlot &s = h->slots[slot];
    if (!s.inFlight && !s.valid) return fail(h, PT_E_BAD_ARGUMENT, "nothing was presented into this slot");
    if (int rc = bind_device(h)) return rc;
    if (s.inFlight) {
        // (a fused present whose successor frame has not been published: nobody will tone-map it on the device — wait_slot closes the launch
        // and joins: resolve_fused_orphan tone-maps the frame from the accumulation image, which holds exactly that frame)
        bool fusedAbandoned = false;
        if (int rc = ptimpl::wait_slot(h, s, &fusedAbandoned)) return rc;
        if (fusedAbandoned) s.abandonEpoch = h->abandonEpoch - 1; // (the image is re-done below, behind a join)
        s.inFlight = false;
        s.valid = true;
        // The image's frames are complete (the copy stream is behind them).  An image tone-mapped behind a JOIN had the hand-over repair
        // passes in front of its tone map (group handles, presents from the accumulation image).  A SNAPSHOT present has no join: if a
        // launch of the handle was abandoned since the tone map was enqueued, the snapshot may lack pixels — repair (the pass also
        // completes the snapshot of the launch that wrote it), tone-map again, copy again.
        if (!h->isGroup() && s.snapSource != nullptr) {
            if (h->hostErrWord && *(volatile unsigned int *)h->hostErrWord) {
                if (int rc = join_stripes(h)) return rc;
                if (*(volatile unsigned int *)h->hostErrWord) ptimpl::note_abandonment(h);
            }
            if (s.abandonEpoch != h->abandonEpoch && s.fedPresent) {
                // a frame-fed launch was abandoned (a hand-over out of its budget, or it ended idle while this frame was being published):
                // the snapshot may lack pixels and its buffer may already hold later frames.  Behind the join the accumulation image is
                // complete (repair passes): show THAT — every frame rendered so far, at least the frame this slot was presented at
                if (int rc = ptimpl::fix_alpha(h)) return rc;
                void *const image = s.boundDev ? s.boundDev : s.dRgba8;
                const size_t pixels = (size_t)s.rows * s.width;
                PT_HIP(h, pt::launch_postprocess(h->accum(), image, pixels, h->stream));
                if (!s.boundDev) PT_HIP(h, hipMemcpyAsync(s.host, s.dRgba8, pixels * 4, hipMemcpyDeviceToHost, h->stream));
                PT_HIP(h, hipStreamSynchronize(h->stream));
                if (int rc = ptimpl::settle_handover(h)) return rc;
                s.abandonEpoch = h->abandonEpoch;
                s.frame = h->frame;
            } else if (s.abandonEpoch != h->abandonEpoch) {
                int k = -1;
                for (int i = 0; i < pt_renderer::kSnapshots; i++)
                    if (h->dSnap[i] == s.snapSource && h->snapGeneration[i] == s.snapGeneration) k = i;
                if (k < 0)
                    return fail(h, PT_E_HIP, "present: a launch was abandoned and this slot's snapshot has been reused (call pt_present_wait on a slot before presenting into it again)");
                if (int rc = join_stripes(h)) return rc;
                void *const image = s.boundDev ? s.boundDev : s.dRgba8;
                const size_t pixels = (size_t)s.rows * s.width;
                PT_HIP(h, pt::launch_postprocess(h->dSnap[k], image, pixels, h->stream));
                if (!s.boundDev) PT_HIP(h, hipMemcpyAsync(s.host, s.dRgba8, pixels * 4, hipMemcpyDeviceToHost, h->stream));
                PT_HIP(h, hipStreamSynchronize(h->stream));
                s.abandonEpoch = h->abandonEpoch;
            }
        }
    }
    if (out_host_rgba8) *out_host_rgba8 = s.boundDev ? nullptr : s.host; // (a bound slot's image is in the caller's device memory)
    if (out_row_pitch_bytes) *out_row_pitch_bytes = (size_t)s.width * 4;
    if (out_frame_index) *out_frame_index = s.frame;
    return PT_OK;
}

PT_API int pt_get_frame_index(pt_handle h, int *out)
{
    PT_CHECK_HANDLE(h);
    if (!out) return fail(h, PT_E_BAD_ARGUMENT, "out == NULL");
    *out = h->isGroup() ? h->parts[0]->frame : h->frame;
    return PT_OK;
}

PT_API int pt_synchronize(pt_handle h)
{
    PT_CHECK_HANDLE(h);
    PT_FAN_OUT(h, pt_synchronize(part));
    if (int rc = bind_device(h)) return rc;
    if (int rc = ptimpl::fix_alpha(h, false)) return rc; // (a bound buffer is observed after this call)
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return ptimpl::settle_handover(h);
}

PT_API int pt_atmosphere_upload_data(pt_handle h, int byte_offset, int size, const void *src)
{
    PT_CHECK_HANDLE(h);
    PT_FAN_OUT(h, pt_atmosphere_upload_data(part, byte_offset, size, src));
    if (!src) return fail(h, PT_E_BAD_ARGUMENT, "src == NULL");
    if (byte_offset < 0 || size < 0 || (long long)byte_offset + size > PT_ATMOSPHERE_UBO_SIZE)
        return fail(h, PT_E_OUT_OF_RANGE, "AtmosphericDataUBO range outside [0,464)");
    std::memcpy(h->atmoUbo + byte_offset, src, (size_t)size);
    return PT_OK;
}

PT_API int pt_atmosphere_render(pt_handle h, int size, int i_steps, int j_steps, const float light_pos[3],
                                float light_intensity)
{
    PT_CHECK_HANDLE(h);
    // every device of a group computes its own copy of the cube (deterministic kernel: the copies are identical)
    PT_FAN_OUT(h, pt_atmosphere_render(part, size, i_steps, j_steps, light_pos, light_intensity));
    if (size <= 0 || size > 8192 || i_steps < 0 || j_steps < 0 || !light_pos)
        return fail(h, PT_E_BAD_ARGUMENT, "bad atmosphere parameters");
    if (int rc = bind_device(h)) return rc;
    if (int rc = join_stripes(h)) return rc;
    size_t total = (size_t)6 * size * size * 16;
    if (total > h->envBytes) {
        PT_HIP(h, hipStreamSynchronize(h->stream));
        if (h->dEnv) PT_HIP(h, hipFree(h->dEnv));
        h->dEnv = nullptr;
        h->envBytes = 0;
        PT_HIP(h, hipMalloc(&h->dEnv, total));
        h->envBytes = total;
    }
    pt::AtmoArgs a;
    std::memcpy(a.invProj, h->atmoUbo, 64);
    std::memcpy(a.invView, h->atmoUbo + 64, 6 * 64);
    a.lightPos[0] = light_pos[0];
    a.lightPos[1] = light_pos[1];
    a.lightPos[2] = light_pos[2];
    a.lightIntensity = light_intensity < 0.0f ? 0.0f : light_intensity; // AtmosphericScatterer.cs:52
    a.size = size;
    a.iSteps = i_steps;
    a.jSteps = j_steps;
    a.out = (float4 *)h->dEnv;
    PT_HIP(h, pt::launch_atmosphere(a, h->stream));
    h->envSize = size;
    h->envFormat = PT_ENV_RGBA32F;
    return PT_OK;
}

PT_API int pt_read_environment(pt_handle h, float *dst, int *out_face_size)
{
    PT_CHECK_HANDLE(h);
    if (h->isGroup()) {
        int rc = pt_read_environment(h->parts[0], dst, out_face_size);
        return rc == PT_OK ? rc : fail(h, rc, h->parts[0]->error);
    }
    if (!h->dEnv) return fail(h, PT_E_NO_ENVIRONMENT, "no environment set");
    if (out_face_size) *out_face_size = h->envSize;
    if (!dst) return PT_OK; // size query
    if (int rc = bind_device(h)) return rc;
    if (int rc = join_stripes(h)) return rc;
    size_t n = (size_t)6 * h->envSize * h->envSize;
    if (h->envFormat == PT_ENV_RGBA32F) {
        PT_HIP(h, hipMemcpyAsync(dst, h->dEnv, n * 16, hipMemcpyDeviceToHost, h->stream));
        PT_HIP(h, hipStreamSynchronize(h->stream));
        return PT_OK;
    }
    float4 *tmp = nullptr;
    PT_HIP(h, hipMalloc((void **)&tmp, n * 16));
    hipError_t e = pt::launch_env_to_float(h->dEnv, h->envSize, h->envFormat, h->dLut, tmp, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dst, tmp, n * 16, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return hip_fail(h, e, "pt_read_environment");
    return PT_OK;
}

PT_API int pt_result_device_ptr(pt_handle h, void **out_ptr, size_t *out_bytes)
{
    PT_CHECK_HANDLE(h);
    if (h->isGroup()) return ptimpl::group_result_device_ptr(h, out_ptr, out_bytes);
    // whatever pt_render deferred is launched and joined into the handle's stream first: work the caller orders behind
    // that stream (or behind pt_synchronize) then sees every frame rendered so far
    if (int rc = bind_device(h)) return rc;
    if (int rc = ptimpl::fix_alpha(h)) return rc;
    if (out_ptr) *out_ptr = h->accum();
    if (out_bytes) *out_bytes = h->tilePixels() * sizeof(float4);
    return PT_OK;
}

PT_API int pt_bind_result_buffer(pt_handle h, void *device_ptr, size_t bytes)
{
    PT_CHECK_HANDLE(h);
    if (h->isGroup()) return fail(h, PT_E_BAD_ARGUMENT, "pt_bind_result_buffer is not available on a group handle");
    if (device_ptr && bytes < h->tilePixels() * sizeof(float4))
        return fail(h, PT_E_BAD_ARGUMENT, "buffer smaller than rows*width*16 bytes");
    if (int rc = bind_device(h)) return rc;
    if (int rc = ptimpl::fix_alpha(h)) return rc; // (the image rendered so far stays behind with alpha = 1)
    h->boundAccum = (float4 *)device_ptr;
    h->boundBytes = device_ptr ? bytes : 0;
    // alpha doubles as the frame tag inside pipelined launches: whatever the caller's memory holds, it starts as 1
    if (device_ptr) PT_HIP(h, pt::launch_set_alpha(h->boundAccum, h->tilePixels(), h->stream));
    return audit_forget(h);
}

PT_API int pt_set_stream(pt_handle h, void *hip_stream)
{
    PT_CHECK_HANDLE(h);
    if (h->isGroup()) return fail(h, PT_E_BAD_ARGUMENT, "pt_set_stream is not available on a group handle");
    if (int rc = bind_device(h)) return rc;
    if (int rc = ptimpl::fix_alpha(h)) return rc;
    PT_HIP(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->ownStream;
    return PT_OK;
}

PT_API int pt_timer_begin(pt_handle h)
{
    PT_CHECK_HANDLE(h);
    PT_FAN_OUT(h, pt_timer_begin(part));
    if (int rc = bind_device(h)) return rc;
    if (int rc = join_stripes(h)) return rc;
    PT_HIP(h, hipEventRecord(h->evBegin, h->stream));
    return PT_OK;
}

PT_API int pt_timer_end(pt_handle h, float *out_ms)
{
    PT_CHECK_HANDLE(h);
    if (!out_ms) return fail(h, PT_E_BAD_ARGUMENT, "out == NULL");
    if (h->isGroup()) return ptimpl::group_timer_end(h, out_ms);
    if (int rc = bind_device(h)) return rc;
    if (int rc = join_stripes(h)) return rc;
    PT_HIP(h, hipEventRecord(h->evEnd, h->stream));
    PT_HIP(h, hipEventSynchronize(h->evEnd));
    PT_HIP(h, hipEventElapsedTime(out_ms, h->evBegin, h->evEnd));
    return PT_OK;
}

// Tuning aid (not declared in the public header): per-wavefront timestamps of the next persistent-kernel launches.
extern "C" __attribute__((visibility("default"))) int pt_debug_timeline(pt_handle h, unsigned long long *host_out, int max_waves)
{
    PT_CHECK_HANDLE(h);
    if (h->isGroup()) return fail(h, PT_E_BAD_ARGUMENT, "not available on a group handle");
    if (int rc = bind_device(h)) return rc;
    if (int rc = join_stripes(h)) return rc;
    if (!h->dTimeline) {
        PT_HIP(h, hipMalloc((void **)&h->dTimeline, (size_t)65536 * 4 * sizeof(unsigned long long)));
        PT_HIP(h, hipMemsetAsync(h->dTimeline, 0, (size_t)65536 * 4 * sizeof(unsigned long long), h->stream));
        return PT_OK;
    }
    PT_HIP(h, hipStreamSynchronize(h->stream));
    if (host_out) PT_HIP(h, hipMemcpy(host_out, h->dTimeline, (size_t)max_waves * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PT_OK;
}

// Tuning aid (not declared in the public header): set one of pt_tuning.hpp's knobs for this process.  The library reads no environment
// variables; this is the only way a knob changes.  Affects renderers created / frames launched afterwards.
extern "C" __attribute__((visibility("default"))) int pt_debug_set(const char *key, long long value)
{
    if (!key || !pt::tuning_set(key, value)) return fail(nullptr, PT_E_BAD_ARGUMENT, "pt_debug_set: unknown knob");
    return PT_OK;
}

// Test aid (not declared in the public header): the hand-over bound's counters of this handle (all parts of a group together).  Drains the
// handle first.  out[0] = (pixel, frame) pairs the repair passes re-rendered, [1] = pixels whose tag fitted nothing the launch sequence can
// have left (must stay 0), [2] = joins that had something to repair, [3] = times the host found the abandon flag raised.
extern "C" __attribute__((visibility("default"))) int pt_debug_handover_stats(pt_handle h, unsigned int out[4])
{
    PT_CHECK_HANDLE(h);
    if (!out) return fail(h, PT_E_BAD_ARGUMENT, "out == NULL");
    if (int rc = pt_synchronize(h)) return rc;
    out[0] = out[1] = out[2] = out[3] = 0;
    std::vector<pt_handle> hs = h->isGroup() ? h->parts : std::vector<pt_handle>{h};
    for (pt_handle p : hs) {
        if (int rc = bind_device(p)) return rc;
        unsigned int ctl[4] = {0, 0, 0, 0};
        PT_HIP(h, hipMemcpy(ctl, p->dRepairCtl, sizeof ctl, hipMemcpyDeviceToHost));
        out[0] += ctl[0];
        out[1] += ctl[1];
        out[2] += ctl[2];
        out[3] += p->abandonEpoch;
    }
    return PT_OK;
}

// Test aid (not declared in the public header): how the handle has been launching.  out[0] = pt_render-driven launch_frames calls (a striped
// frame counts once), [1] = 1 while the cached tile masks are valid, [2] = frames accepted by pt_render and not launched yet, [3] = tile-mask
// rebuilds, [4] = frame counter, [5] = flushes forced by an input change (upload / set_params with different values), [6] = frames published into frame-fed launches, [7] = fed
// launches opened, [8] = fed launches that ended idle, [9] = 1 while one is open, [10] = 1 once the host has pipelined frames (single frames then go
// out as tagged launches).  Does not flush or join.
#ifdef PT_FEED_TIMES
extern "C" __attribute__((visibility("default"))) int pt_debug_feed_times(pt_handle h, unsigned int *out128)
{
    for (int i = 0; i < 64; i++) { out128[i] = h->hostStarted[3000 + i]; out128[64 + i] = h->hostStarted[3100 + i]; }
    return PT_OK;
}
#endif
extern "C" __attribute__((visibility("default"))) int pt_debug_launch_stats(pt_handle h, unsigned long long out[12])
{
    PT_CHECK_HANDLE(h);
    if (!out) return fail(h, PT_E_BAD_ARGUMENT, "out == NULL");
    pt_handle r = h->isGroup() ? h->parts[0] : h;
    out[0] = r->statLaunches;
    out[1] = r->tileMasksValid ? 1 : 0;
    out[2] = (unsigned long long)r->pendingFrames;
    out[3] = r->statMaskBuilds;
    out[4] = (unsigned long long)r->frame;
    out[5] = r->statFlushes;
    out[6] = r->statPublishes;
    out[7] = r->statFeedOpens;
    out[8] = r->statFeedIdle;
    out[9] = r->feed.open ? 1 : 0;
    out[10] = r->sawBatch ? 1 : 0;
    out[11] = 0;
    return PT_OK;
}

// Test aid (not declared in the public header): the sphere grid the NEXT launch would use for the current scene.
// out[0..2] = cells per axis, out[3] = sphere references, out[4] = 1 if a grid exists (else the in-order loop runs).
extern "C" __attribute__((visibility("default"))) int pt_debug_sphere_grid(pt_handle h, int out[5])
{
    PT_CHECK_HANDLE(h);
    if (!out) return fail(h, PT_E_BAD_ARGUMENT, "out == NULL");
    pt_handle r = h->isGroup() ? h->parts[0] : h;
    const ptgrid::SphereGrid g = r->gridDirty ? ptgrid::build((const float *)r->objectsShadow, r->numSpheres) : r->grid;
    for (int k = 0; k < 3; k++) out[k] = g.dims[k];
    out[3] = g.numRefs;
    out[4] = g.valid ? 1 : 0;
    return PT_OK;
}

// Test aid (not declared in the public header; needs no GPU): build the sphere grid of a std140 GameObjectsUBO on the host.
// header[0..2] = cells per axis, [3] = references, [4] = valid; box[0..2] = lo, [3..5] = hi, [6..8] = centre, [9] = reach^2;
// packed (capacity bytes) receives uint16 starts[cells + 1] followed by uint8 refs[].  Returns the packed size.
extern "C" __attribute__((visibility("default"))) int pt_debug_build_sphere_grid(const float *objects, int num_spheres, int header[5],
                                                                                 float box[10], unsigned char *packed, int capacity)
{
    if (!objects || !header || !box) return PT_E_BAD_ARGUMENT;
    const ptgrid::SphereGrid g = ptgrid::build(objects, num_spheres);
    for (int k = 0; k < 3; k++) {
        header[k] = g.dims[k];
        box[k] = g.lo[k];
        box[3 + k] = g.hi[k];
        box[6 + k] = g.center[k];
    }
    header[3] = g.numRefs;
    header[4] = g.valid ? 1 : 0;
    box[9] = g.reach2;
    if (packed && capacity >= (int)g.packed.size() && !g.packed.empty()) std::memcpy(packed, g.packed.data(), g.packed.size());
    return (int)g.packed.size();
}

// Test aid (not declared in the public header): the hand-over audit of the -DPT_AUDIT build.  Drains the handle, then copies up to
// max_records violation records (pt::kAuditRecordWords words each) and returns their number (all parts of a group together), or
// -1000 when the library was built without PT_AUDIT.  The log is cleared.
extern "C" __attribute__((visibility("default"))) int pt_debug_audit_read(pt_handle h, unsigned int *out_records, int max_records)
{
    PT_CHECK_HANDLE(h);
#ifndef PT_AUDIT
    (void)out_records;
    (void)max_records;
    return -1000;
#else
    if (int rc = pt_synchronize(h)) return rc;
    int total = 0, copied = 0;
    std::vector<pt_handle> hs = h->isGroup() ? h->parts : std::vector<pt_handle>{h};
    for (pt_handle p : hs) {
        volatile unsigned int *log = p->hostAuditLog;
        if (!log) continue;
        const int n = (int)log[0], kept = n < pt::kAuditLogRecords ? n : pt::kAuditLogRecords;
        for (int i = 0; i < kept && copied < max_records && out_records; i++, copied++)
            for (int k = 0; k < pt::kAuditRecordWords; k++)
                out_records[(size_t)copied * pt::kAuditRecordWords + k] = log[4 + (size_t)i * pt::kAuditRecordWords + k];
        total += n;
        log[0] = 0;
    }
    return total;
#endif
}

PT_API int pt_set_frame_batch(pt_handle h, int max_frames)
{
    PT_CHECK_HANDLE(h);
    if (max_frames < 0 || max_frames > 64) return fail(h, PT_E_BAD_ARGUMENT, "max_frames must be 0..64");
    PT_FAN_OUT(h, pt_set_frame_batch(part, max_frames));
    if (int rc = flush_frames(h)) return rc;
    h->maxBatch = max_frames > 0 ? max_frames : 64;
    h->maxBatchExplicit = max_frames > 0; // (a limit the host asked for bounds latency and deferral: never raised behind its back; 0 = automatic again)
    return PT_OK;
}

PT_API int pt_set_variant(pt_handle h, int variant)
{
    PT_CHECK_HANDLE(h);
    PT_FAN_OUT(h, pt_set_variant(part, variant));
    if (int rc = flush_frames(h)) return rc; // pending frames were rendered with the inputs as they were
    if (int rc = bind_device(h)) return rc;
    if (int rc = join_stripes(h)) return rc; // a different stripe partition must not overlap frames in flight
    h->variant = variant;
    return PT_OK;
}

PT_API int pt_device_count_of(pt_handle h, int *out_n_devices)
{
    PT_CHECK_HANDLE(h);
    if (!out_n_devices) return fail(h, PT_E_BAD_ARGUMENT, "out == NULL");
    *out_n_devices = h->isGroup() ? (int)h->parts.size() : 1;
    return PT_OK;
}

} // extern "C"
