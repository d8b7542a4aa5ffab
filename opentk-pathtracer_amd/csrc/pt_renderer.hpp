// pt_renderer.hpp — host-side state behind a pt_handle, shared by mi355pt.cpp (one renderer on one GPU) and
// mi355pt_multi.cpp (a group of renderers, one per GPU, row-tiled; gather over xGMI at read / present time).
//
// What the reference keeps in the C# class PathTracer (/root/reference/OpenTK-PathTracer/src/Render/PathTracer.cs:9-141)
// plus the two UBOs MainWindow owns (src/MainWindow.cs:195-201), and the HIP plumbing around the kernels of
// the pt_*.hip files.  There is deliberately NO CPU fallback anywhere behind this struct.
#pragma once
#include "../../include/mi355pt.h"

#include <hip/hip_runtime.h>

#include <deque>
#include <string>
#include <vector>

#include "pt_kernels.hpp"
#include "pt_sphere_grid.hpp"

namespace ptimpl {
constexpr uint32_t kAlive = 0x4d335054u; // "M3PT"
constexpr int kMaxStripes = 4;
// global ticket counters of the persistent kernel: stripe j draws from word 16 * j, the chain stream from its own 128-byte line
constexpr int kQueueWords = 128, kChainQueueWord = 64;
// The per-wavefront timestamps of tools/timeline.py need one plain launch per frame (no batching, chaining or snapshots).  In a
// -DPT_PROFILE build the same buffer only receives the section counters (tools/profile_sections.py), which every kernel and every
// launch mode can write: the library keeps pipelining, so that the profile is taken in the mode bench.py measures.
#ifdef PT_PROFILE
#define PT_TIMELINE_BLOCKS(h) false
#else
#define PT_TIMELINE_BLOCKS(h) ((h)->dTimeline != nullptr)
#endif
constexpr int kStartedWords = pt::kStartedWords; // >= workgroups of any persistent launch (8 per CU)

// One image of the non-blocking present path (pt_present_rgba8_async / pt_present_wait).
struct PresentSlot {
    void *dRgba8 = nullptr;       // device RGBA8 image: this handle's rows (compact), or the whole image on a group handle
    void *boundDev = nullptr;     // caller-owned device image (pt_present_bind_device_image): the tone map writes here, nothing is copied
    size_t boundBytes = 0;
    size_t devPixels = 0;         // capacity of dRgba8
    uint8_t *host = nullptr;      // pinned host image (absent on the parts of a group: only the group handle copies to the host)
    size_t hostPixels = 0;
    hipEvent_t toneMapped = nullptr; // recorded behind the tone-map pass on the handle's stream
    hipEvent_t copied = nullptr;     // recorded behind the device-to-host copy on the copy stream
    bool inFlight = false;        // a copy from dRgba8 / into host was enqueued and not yet waited for
    bool valid = false;           // host holds an image (after pt_present_wait)
    int frame = 0, rows = 0, width = 0;
    // snapshot presents (no join between the frames' launch and the tone map): the snapshot the image was tone-mapped from, and the
    // handle's abandon epoch at that time — if a launch was abandoned meanwhile, pt_present_wait repairs and tone-maps again
    const void *snapSource = nullptr;
    unsigned int snapGeneration = 0, abandonEpoch = 0;
    bool fedPresent = false;      // the image is produced by a frame-fed launch's fused display (FrameArgs::displayImages)
    int fusedWord = 0;            // ... which host word of the feed ring its launch reports completed frames in,
    unsigned int fusedNeed = 0;   // ... and how many frames of that launch must be complete for the image to be
    bool fusedWaiting = false;    // ... and the frame that produces it (the one after the frame shown) has not been published yet: no event to wait for so far
};
} // namespace ptimpl

struct pt_renderer {
    uint32_t magic = ptimpl::kAlive;
    int device = 0;
    int width = 0, height = 0;
    int y0 = 0, rows = 0;
    int bandRows = 0, bandWorld = 1, bandRank = 0; // block-cyclic row ownership (pt_set_interleaved_tile)
    int numSpheres = 0, numCuboids = 0, rayDepth = 1, spp = 1;
    float focalLength = 0.0f, apertureDiameter = 0.0f;
    int frame = 0; // thisRenderNumFrame, PathTracer.cs:113
    int variant = 0;

    unsigned char basic[PT_BASIC_DATA_UBO_SIZE] = {0};   // host shadow of UBO 0 (travels as kernel argument)
    unsigned char atmoUbo[PT_ATMOSPHERE_UBO_SIZE] = {0}; // host shadow of UBO 2

    float *dObjects = nullptr;      // 26,624 B device copy of UBO 1
    // Sphere grid of large scenes (pt_sphere_grid.hpp): rebuilt from a host shadow of UBO 1 before the first launch after the
    // scene or the sphere count changed
    unsigned char objectsShadow[PT_GAME_OBJECTS_UBO_SIZE] = {0};
    bool gridDirty = true;
    ptgrid::SphereGrid grid;
    unsigned long long sphereRunStart[4] = {~0ull, ~0ull, ~0ull, ~0ull}; // FrameArgs::sphereRunStart, rebuilt with the grid
    // Cached tile masks (FrameArgs::tileMasks): rebuilt by pt_tile_masks_kernel when the camera, the lens, the spheres or the tiling
    // changed AND the camera has then been left alone for two launches (a host that moves the camera every frame keeps the per-tile
    // culling of the tile pass: rebuilding needs the two launch streams joined, which would break its chaining)
    unsigned long long *dTileMasks = nullptr;
    size_t tileMaskTiles = 0;        // capacity in tiles
    bool tileMasksValid = false;
    int launchesSinceInputChange = 0; // (frames launched or published since the inputs last changed; the name is from round 4, when it counted launches)
    unsigned long long statLaunches = 0, statMaskBuilds = 0, statFlushes = 0; // pt_debug_launch_stats: integrator launches, mask rebuilds, input-change flushes
    unsigned char *dGrid = nullptr; // (kMaxCells + 1) * 2 + kMaxRefs bytes
    float *dLut = nullptr;          // 256-entry sRGB table
    unsigned int *dQueue = nullptr; // global chunk-ticket counter of the persistent kernel (never reset: epoch scheme)
    // Hand-over bound (round 5; pt_kernel_common.hpp).  A tagged launch whose wait for a pixel's previous frame runs out of its
    // wall-clock budget ABANDONS itself instead of folding onto a stale pixel: dAbandon (device word, ~0u = none) receives the lowest
    // abandoned launch sequence number, hostErrWord (ONE page-locked host word the kernels can reach) is raised so that the host
    // notices without a copy.  Every tagged launch since the last join of the handle's streams is kept in `unverified` (its kernel
    // argument + an event behind it); a join enqueues pt_repair_kernel for each of them, in order, behind the joined streams — a no-op
    // on the device unless dAbandon says otherwise — and forgets them; launches seen complete with hostErrWord == 0 are forgotten
    // earlier.  Whatever the host or a later launch reads behind a join is therefore the image an undisturbed run produces.
    unsigned int *hostErrWord = nullptr, *devErrWord = nullptr;
    unsigned int *dAbandon = nullptr;
    unsigned int *dRepairCtl = nullptr;   // 4 words: (pixel, frame) pairs re-rendered, inconsistent pixels, joins with repairs, spare
    unsigned int *dTileFlags = nullptr;   // FrameArgs::tileFlags: one word per 8x8 tile of the handle's rows
    size_t tileFlagTiles = 0;
    struct LaunchRecord { pt::FrameArgs a; hipEvent_t done; };
    std::deque<LaunchRecord> unverified;
    static constexpr int kLaunchEvents = 256; // ring of "launch done" events (mainDone / chainDone alias the latest ones)
    hipEvent_t launchEvents[kLaunchEvents] = {};
    unsigned int launchEventNext = 0;
    bool repairPendingAll = false;        // the host cleared a raised abandon flag and has not yet enqueued the repair passes of the launches remembered
    bool repairCheckDue = false;          // repair passes may have run since dRepairCtl was last read (checked by the next blocking call)
    unsigned int inconsistentSeen = 0;    // dRepairCtl[1] at that reading
    unsigned int abandonEpoch = 0;        // bumped whenever the host finds hostErrWord raised (present slots remember it)
    int overlapHoldoff = 0;               // launches that still run BEHIND their predecessor after an abandonment (a contended device)
    int wallClockKhz = 100000;            // rate of the device's constant-rate counter (s_memrealtime)
    unsigned int waitBudgetUnits = 0, waitCheckUnits = 0; // FrameArgs::waitBudget / waitCheckInterval for this device's wall clock
    int queueChunk = 0;             // tiles per global ticket; 0 = automatic (tuning knob queue_chunk)
    unsigned long long *dTimeline = nullptr; // tuning only (pt_debug_timeline)
    // Frame pipelining: consecutive pt_render calls are collected and launched as ONE batch kernel (see pt_integrate_persistent.hip)
    // when nothing observable happens in between; every other entry point launches what is pending first.
    int pendingFrames = 0;        // frames accepted by pt_render, not launched yet
    int maxBatch = 64;            // pt_set_frame_batch: 1 turns batching off (every pt_render launches at once)
    bool maxBatchExplicit = false; // the host called pt_set_frame_batch: its limit is kept as given (no automatic 256-frame launches)
    int batchWorkgroupsPerCU = 6; // grid of the batch kernel (tuning knob batch_wg)
    bool batchLaunched = false;   // a batch kernel ran since the last error-word check
    bool lastPresentBound = false; // the newest present went into a slot bound to caller-owned device memory (pt_present_bind_device_image)
    int rendersSincePresent = 0;  // pt_render calls since the last pt_present_rgba8_async ...
    int presentCadence = 0;       // ... and how many there were before that present (1 = the host presents every frame)
    hipEvent_t mainDone = nullptr; // recorded behind the last integrator launch on the main stream
    bool mainInFlight = false;     // ... and not yet seen complete
    // Launch chaining (pipelined launches of the default kernel): consecutive tagged launches alternate between the main stream
    // and `chainStream` and are ordered per PIXEL by the alpha tags, not by the streams, so a launch starts in the wavefront
    // slots the previous launch's drain frees.  While `tagsLive` the image's alpha channel holds tags: every entry point that
    // lets the host observe the image restores alpha = 1 first (fix_alpha).
    // (the second launch stream is stripeStream[1]: the helper stream of the striped frames doubles as the chain stream)
    hipEvent_t chainDone = nullptr;    // recorded behind the last launch on chainStream
    // A launch may only run BESIDE its predecessor (on the other stream) when that predecessor is fully resident: a dependent launch
    // that got hold of the machine first would leave the launch it waits for a handful of workgroup slots (seen once: a 4K
    // half-image took 1.7 s per frame).  Every workgroup of a tagged launch reports in by storing the launch's sequence number
    // into its word of this host-mapped array; the host looks before it chooses the stream of the next launch.
    unsigned int *hostStarted = nullptr, *devStarted = nullptr; // kStartedWords words
    unsigned int launchSeq = 0;        // sequence number of the latest tagged launch
    int lastWorkgroups = 0;            // ... its grid size
    int lastStreamIdx = 0;             // ... and its stream (0 = main, 1 = chainStream)
    bool chainInFlight = false, chainPending = false; // (for gpu_busy / the main stream has not yet waited for it)
    bool chainNeedsInputs = false;     // the chain stream has not yet waited for the inputs put on the main stream
    bool chainBroken = true;           // something other than a tagged launch happened since the last one: streams re-join first
    bool tagsLive = false;             // the image's alpha holds frame tags (last one: lastTag)
    float lastTag = 0.0f;
    bool flushFinal = false;           // the flush comes from an entry point that joins the streams: its launch stores alpha = 1 last
    bool sawBatch = false;             // the host has pipelined frames before: single frames launch tagged too, so that they overlap
    bool stripeInFlight[ptimpl::kMaxStripes] = {false, false, false, false}; // same for the stripe streams
    int drainCompaction = -1;      // donate threshold in live paths (<= 32), 0 = off, -1 = auto; tuning knob drain_compaction
    int numCUs = 256;
    void *dEnv = nullptr; // current environment cube
    size_t envBytes = 0;
    int envSize = 0, envFormat = PT_ENV_RGBA32F;

    float4 *dAccum = nullptr;     // internal accumulation image (rows x width)
    size_t accumCapacity = 0;     // in pixels
    float4 *boundAccum = nullptr; // caller-owned target (pt_bind_result_buffer)
    void *dRgba8 = nullptr;       // post-processed RGBA8 image of the tile (pt_present_rgba8)
    size_t rgba8Capacity = 0;     // in pixels
    size_t boundBytes = 0;

    // hand-over audit (only allocated by the -DPT_AUDIT build, see pt_debug_hooks.hpp): side word per accumulation pixel + violation log
    unsigned long long *dAudit = nullptr;
    size_t auditCapacity = 0; // in pixels
    unsigned int *hostAuditLog = nullptr, *devAuditLog = nullptr;

    hipStream_t ownStream = nullptr, stream = nullptr;
    // Stripes: one frame = `stripes` persistent kernels over contiguous row ranges of the tile, each on its own
    // stream, so that one stripe's frame-end drain overlaps the other stripe's main phase.
    hipStream_t stripeStream[ptimpl::kMaxStripes] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t stripeDone[ptimpl::kMaxStripes] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t inputsReady = nullptr;
    bool mainDirty = true; // work other than stripe 0's frames was put on the main stream since the last striped frame
    bool stripePending[ptimpl::kMaxStripes] = {false, false, false, false};
    int stripeRow0[ptimpl::kMaxStripes] = {0, 0, 0, 0}, stripeRows[ptimpl::kMaxStripes] = {0, 0, 0, 0}; // rows of the last launch
    unsigned int stripeQueueBase[ptimpl::kMaxStripes] = {0, 0, 0, 0};
    unsigned int chainQueueBase = 0; // same for the chain stream's counter
    hipEvent_t evBegin = nullptr, evEnd = nullptr;

    // non-blocking present
    hipStream_t copyStream = nullptr;
    ptimpl::PresentSlot slots[PT_PRESENT_SLOTS];
    // Present snapshots (round 3): the launch that pt_present_rgba8_async flushes stores its last frame's pixels into one of these
    // (FrameArgs::snapshot) while it resolves them, so the tone map reads a frame that no later launch touches and the next
    // launch need not wait for it: it chains on the other stream like any pipelined launch.  Three buffers in rotation; a buffer
    // is rewritten only after the tone map that read it (snapRead) has run.
    static constexpr int kSnapshots = 3;
    float4 *dSnap[kSnapshots] = {nullptr, nullptr, nullptr};
    size_t snapCapacity[kSnapshots] = {0, 0, 0}; // in pixels
    hipEvent_t snapRead[kSnapshots] = {nullptr, nullptr, nullptr};
    bool snapReadPending[kSnapshots] = {false, false, false};
    int snapNext = 0;
    unsigned int snapGeneration[kSnapshots] = {0, 0, 0}; // bumped whenever a launch is told to write the buffer
    float4 *snapshotTarget = nullptr;            // set around the flush of pt_present_rgba8_async: the launch's FrameArgs::snapshot
    int snapshotIndex = -1;
    int snapFrame = -1;                          // frame counter value whose image the snapshot written last holds (-1: none)
    struct SnapLaunch { hipStream_t stream; hipEvent_t done; size_t firstPixel, pixels; };
    std::vector<SnapLaunch> snapLaunches;        // the launch(es) that write the snapshot: stream, its "done" event, the rows they cover

    // Frame-fed launch (round 6; FrameArgs::feedHost, pt_integrate_persistent.hip FEED).  When a launch of a few frames goes out while the
    // host is not far ahead — the reference's own usage: Render() then show, every frame (MainWindow.cs:49-64) — it is started with room
    // for kFeedCapacity frames, and the pt_render calls that follow PUBLISH their frame into it (one store to a host-mapped word) instead
    // of launching: the wavefronts stay resident between frames.  Only the host closes a launch (the count is then final and exact);
    // anything that launches, joins, flushes or changes an input closes it first.  A launch whose wavefronts wait too long for the next
    // frame abandons itself with reason "idle" (hand-over bound), so a host that stops rendering never keeps the GPU.
    struct FeedState {
        bool open = false;           // pt_render may publish into the launch
        int firstFrame = 0, published = 0;
        int streamIdx = 0, hostWord = 0;
        unsigned int seq = 0;        // its launchSeq (its LaunchRecord is patched with the final frame count when it is closed)
        int workgroups = 0, queueChunk = 8;
        long long tilesFrame = 0;
        bool withMasks = false;      // it runs with the cached tile masks
        bool display = false;        // FUSED DISPLAY: the launch tone-maps the frames the host shows into the present slots' images itself
        unsigned int slots16 = 0;    // ... the low 16 bits of the feed word: present slot + 1 of the 8 most recent frames
        int pendingSlot = -1;        // ... the slot the newest published frame was presented into: its image is produced by the NEXT frame's tile passes
        unsigned long long base[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // feedDoneBase of its stream when it was opened
        unsigned long long pixelsPerFrame = 0;
    } feed;
    static constexpr int kFeedHostWords = 8;
    unsigned int *hostFeedDone = nullptr, *devFeedHostDone = nullptr; // ... and per word: frames of that launch its monitor wavefront has seen complete
    unsigned int *hostFeed = nullptr, *devFeedHost = nullptr; // kFeedHostWords host-mapped words, handed out in rotation ...
    hipEvent_t feedWordBusy[kFeedHostWords] = {};             // ... a word is reused only when the launch that read it is complete
    int feedWordNext = 0;
    unsigned int *dFeedDev = nullptr;          // 2 x kFeedBcastSlots broadcast slots (per launch stream): the feed word as the launch's monitor last read it
    unsigned long long *dFeedDone = nullptr;   // 2 x 8 cumulative per-frame-slot counters (per launch stream)
    unsigned long long feedDoneBase[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}}; // what the counters read once every closed launch has finished
    hipEvent_t feedResetEvent = nullptr;
    bool feedCountersStale = false;            // a launch was abandoned: its counts are short; re-zero behind a full join before the next fed launch
    int fusedOrphanSlot = -1;                  // a present into this slot was left without a successor frame when its launch was closed: the next join tone-maps it from the image
    int feedIdleStrikes = 0, feedHoldoff = 0;  // fed launches that ended idle in a row; launches for which no fed launch is tried
    unsigned long long statPublishes = 0, statFeedOpens = 0, statFeedIdle = 0;

    // group handle (pt_create_multi): parts[i] renders its share on device_ids[i]; this struct then only carries the root
    // device's streams, the gather buffers and the present slots
    std::vector<pt_renderer *> parts;
    int groupBand = 8;              // 0 = contiguous row blocks; 8 = single tile rows (1080p over 8 devices: largest share 136 rows; 16-row bands: 144)
    bool gatherDirect = true;       // (group) every peer copies to the root over a direct link (hipDeviceCanAccessPeer both ways); pt_multi_gather_is_direct
    hipEvent_t gatherReady = nullptr; // (parts) recorded on the part's stream when its rows may be copied
    void *dGatherFull = nullptr;      // (group) assembled image on the root device (RGBA32F or RGBA8)
    size_t gatherFullBytes = 0;
    void *dGatherStage = nullptr;     // (group) the parts' compact rows side by side, before un-banding
    size_t gatherStageBytes = 0;
    void *dAsyncStage = nullptr;      // (group) the same for the copy stream (asynchronous present)
    size_t asyncStageBytes = 0;

    std::string error;

    bool isGroup() const { return !parts.empty(); }
    bool externalStream() const { return stream != ownStream; }
    float4 *accum() const { return boundAccum ? boundAccum : dAccum; }
    size_t tilePixels() const { return (size_t)rows * (size_t)width; }
};

namespace ptimpl {

inline bool timeline_blocks_pipelining(const pt_renderer *h) { return PT_TIMELINE_BLOCKS(h); }
int fail(pt_handle h, int code, const std::string &msg);
int hip_fail(pt_handle h, hipError_t e, const char *what);

#define PT_CHECK_HANDLE(h)                                                                                             \
    do {                                                                                                               \
        if (!(h) || (h)->magic != ptimpl::kAlive) return ptimpl::fail(nullptr, PT_E_BAD_HANDLE, "bad handle");        \
    } while (0)

#define PT_HIP(h, call)                                                                                                \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) return ptimpl::hip_fail((h), e_, #call);                                                 \
    } while (0)

int bind_device(pt_handle h);
void feed_close(pt_handle h);   // the open frame-fed launch (if any) takes no more frames; bookkeeping of its final frame count
int flush_frames(pt_handle h);  // launch the frames pt_render deferred (a blocking entry point's flush: may wait chain_wait_us per launch)
int flush_frames_bounded(pt_handle h, long waitUs, int maxLaunches);
int batch_limit(pt_handle h);
bool launch_ready(pt_handle h);
// flush + make h->stream wait for every helper stream + (repairNow) enqueue the hand-over repair passes of the launches since the last
// join behind them.  repairNow = false is for callers that synchronise h->stream right away and then call settle_handover(): the
// repair kernels are then only enqueued when the host-visible flag says a launch was abandoned (nothing at all on the usual path)
int join_stripes(pt_handle h, bool repairNow = true);
int settle_handover(pt_handle h); // h->stream synchronised after join_stripes(h, false): repair if the flag is up, forget the launches
hipEvent_t next_launch_event(pt_handle h);
int fix_alpha(pt_handle h, bool repairNow = true); // join + restore alpha = 1 if the image still carries frame tags (before the host observes it)
int ensure_stripe(pt_handle h, int j); // create stripe stream j (j > 0) and its event on first use
hipStream_t stripe_stream(pt_handle h, int j); // stripe 0 runs on the main stream
// tone map this handle's rows into `dst` (RGBA8, compact rows) on h->stream, behind every frame rendered so far
int tone_map_into(pt_handle h, void *dst);
// slot plumbing shared by the single and the group path
int ensure_slot_device(pt_handle h, int slot, size_t pixels);
int ensure_slot_host(pt_handle h, int slot, size_t pixels);
int ensure_slot_events(pt_handle h, int slot);
void free_slots(pt_handle h);
int wait_slot(pt_handle h, PresentSlot &s, bool *abandoned); // host-side wait for the previous present into a slot (event, or a fused present's host word)

// group handles (mi355pt_multi.cpp)
int group_destroy(pt_handle g);
int group_set_size(pt_handle g, int width, int height);
int group_read_result(pt_handle g, float *dst, size_t row_pitch_bytes);
int group_write_result(pt_handle g, const float *src, size_t row_pitch_bytes, int frame_index);
int group_present_rgba8(pt_handle g, uint8_t *dst, size_t row_pitch_bytes);
int group_present_async(pt_handle g, int slot);
int group_result_device_ptr(pt_handle g, void **out_ptr, size_t *out_bytes);
int group_timer_end(pt_handle g, float *out_ms);

} // namespace ptimpl
