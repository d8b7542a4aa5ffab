// pt_kernels.hpp — host-visible launch interface of the HIP kernels (implemented in pt_integrate_persistent.hip, pt_integrate_multisample.hip
// and pt_helper_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pt {

// Everything one dispatch of the integrator needs; passed BY VALUE as the kernel argument so that all of it is
// wave-uniform scalar (SGPR) data.  Mirrors what the reference binds around GL.DispatchCompute
// (src/Render/PathTracer.cs:116-123): UBO 0 (camera), UBO 1 (scene), the 7 uniforms, sampler unit 1, image unit 0.
struct FrameArgs {
    float invProj[16]; // BasicDataUBO.InvProjection, GLSL column-major view of the bytes (compute.glsl:61)
    float invView[16]; // BasicDataUBO.InvView (compute.glsl:62)
    float viewPos[3];  // BasicDataUBO.ViewPos (compute.glsl:63)
    float focalLength, apertureDiameter; // compute.glsl:93-94
    int width, height; // full image size: imageSize(ImgResult) (compute.glsl:103)
    int y0, rows;      // row block owned by this GPU (multi-GPU tiling); y0=0, rows=height on one GPU
    // interleaved (block-cyclic) ownership for load balance across GPUs: when bandRows > 0 this GPU owns the bands
    // b = bandRank, bandRank + bandWorld, ... of bandRows image rows each, stored compactly; local row l (counted from
    // the start of the GPU's storage, localRow0 = first local row of this launch) is image row
    // ((l / bandRows) * bandWorld + bandRank) * bandRows + l % bandRows.  bandRows == 0: image row = y0 + local row.
    int bandRows, bandWorld, bandRank, localRow0;
    int numSpheres, numCuboids; // uboGameObjectsSize (compute.glsl:88)
    int rayDepth, spp; // compute.glsl:90-91
    int frame;         // thisRendererFrame (compute.glsl:96)
    int envSize, envFormat; // cube face size; 0 = RGBA32F, 1 = SRGB8_A8
    const float *objects;   // device copy of the raw 26,624-byte GameObjectsUBO (std140)
    const void *env;        // device cube: [6][envSize][envSize] texels (float4 or uchar4)
    const float *srgbLut;   // 256-entry sRGB8 -> linear table (device)
    float4 *accum;          // rows x width RGBA32F accumulation image (row 0 = image row y0)
    int tilesX, tilesY;     // 8x8-pixel tiles covering width x rows
    int variant;            // kernel variant for A/B runs; all variants are bit-identical in output
    unsigned int *queue;    // global chunk-ticket counter of the persistent kernel (monotonic across launches)
    unsigned int *errorWord; // host-visible flag: a launch of this handle ABANDONED its hand-over (see abandonWord); the host repairs
    unsigned int queueBase; // value of *queue when this launch starts (every launch consumes exactly numChunks tickets)
    int numCUs;             // compute units of the device (grid sizing of persistent variants)
    int queueChunk;         // tiles per global ticket of the persistent kernel's queue
    int drainCompaction;    // 0 = off; else a draining wavefront with at most this many live paths donates them (<= 48)
    int batchFrames;        // frames rendered by this launch (>= 1): frame, frame+1, ... over the same tiles (see the kernel's frame pipelining)
    int tagged;             // 1 = pixels are handed over through alpha tags (device-coherent 16-byte accesses): every launch of more than one
                            // frame, and single frames that may overlap a neighbouring launch; 0 = plain read-modify-write, alpha = 1
    int keepTags;           // 1 = the LAST frame of the launch stores its tag too (the next launch chains on it; the host restores
                            // alpha = 1 before anything observes the image), 0 = it stores the 1 the reference stores
    float chainTag;         // tagged launches: the tag the pixel must show before the launch's FIRST frame may resolve it (the last
                            // frame of the previous, possibly still running launch); 0 = no such predecessor
    int contCapacity;       // set by the launch (spp > 1 batch-pass kernel): parked continuations per wavefront
    int contBatchMin;       // ... and how many of them make a batch pass worth running
    int parkedMax;          // set by the launch (pipelined spp = 1 launches): parked resolves per wavefront (0 = waiting results keep their lanes)
    int materialsInLds;     // set by the launch: 1 = the 64-byte materials are staged in LDS, 0 = read from `objects` (large scenes)
    unsigned long long *timeline; // optional (tuning): per wavefront {start, queue exhausted, end, iterations} timestamps
    // Launch chaining: every workgroup of a tagged launch stores launchSeq into startedFlags[blockIdx.x] (host-visible memory)
    // when it starts, so that the host can tell whether the launch is fully RESIDENT (see launch_frames in mi355pt.cpp)
    unsigned int *startedFlags;
    unsigned int launchSeq; // sequence number of this launch on its handle (every tagged launch has one; never 0)
    // Hand-over bound (round 5).  A result whose pixel does not show the previous frame's tag within waitBudget (wall clock, units of
    // 1,024 ticks of the constant 100 MHz counter) is NOT folded onto a stale pixel: the launch is ABANDONED — abandonWord (device
    // memory, one per handle, ~0u = none) receives the lowest abandoned launchSeq, every launch of that or a later sequence number stops
    // drawing tickets and drops the results that still wait — and the host re-renders exactly the missing (pixel, frame) pairs behind
    // it with pt_repair_kernel (every pixel's alpha tag says which frame it holds), so the image is the one an undisturbed launch
    // produces.  tileFlags (only the first launch of a chain: chainTag == 0): one word per 8x8 tile = launchSeq of the last such launch
    // whose FIRST frame ran its tile pass there (tells an untouched pixel from a finished one where both hold alpha = 1).
    unsigned int *abandonWord;
    unsigned int *tileFlags;
    unsigned int waitBudget, waitCheckInterval; // (same unit; the abandon word is looked at once per waitCheckInterval while a wavefront waits)
    // Sphere grid of large scenes (pt_sphere_grid.hpp; nullptr = none): a uniform grid over the spheres' bounds, per cell the
    // ascending list of the spheres whose (slightly inflated) bounding box touches it.  Packed as uint16 starts[cells + 1]
    // followed by uint8 refs[starts[cells]]; staged into LDS by the kernels that traverse it.
    const unsigned char *grid;
    int gridBytes;          // size of the packed grid
    int gridLdsBytes;       // set by the launch: LDS bytes the staged grid takes (0 = this launch does not traverse it)
    unsigned int tilesFrameMagic, tilesXMagic; // floor(2^32 / (tilesX * tilesY)), floor(2^32 / tilesX): the tile pass splits a ticket's (frame, tile) pair with them
    float invW, invH;       // 1 / width, 1 / height: IEEE quotients, computed once on the host (what the tile pass used to divide out per tile)
    int sceneLdsBytes;      // set by the launch: scene_lds_bytes(...) of this launch = where the frame table starts (one scalar load where a
                            // pixel is resolved, instead of re-deriving it from five other fields)
    int gridDims[3];        // cells per axis (their product is the cell count)
    float gridLo[3], gridHi[3], gridCell[3], gridInvCell[3]; // box, cell size and its reciprocal per axis
    float gridCenter[3];    // a ray uses the grid when its origin lies within sqrt(gridReach2) of the box centre (see ray_trace_t)
    float gridReach2;
    // Sphere runs (in-order sphere loop, ray_trace_t): bit i is set when sphere i does NOT have the same centre x AND z bits as sphere
    // i - 1 (bit 0 is always set).  Inside a run the loop reuses o.x - c.x, o.z - c.z and the two products that start the dot-product
    // chains — the same binary32 values the per-sphere evaluation would produce (pt_sphere_grid.hpp: sphere_runs).
    unsigned long long sphereRunStart[4];
    // Cached tile masks (nullptr = the tile pass culls against its own 64 rays): per 8x8 tile of this launch's rows 4 x 64 bits, the
    // spheres that ANY primary ray of the tile can reach, whatever the jitter and the lens sample — computed once per camera / scene by
    // pt_tile_masks_kernel (tile_cone in pt_device.hpp) and read with scalar loads by every frame's tile pass
    const unsigned long long *tileMasks; // kTileMaskWords words per tile: [0..3] spheres, [4] cuboids
    // Present snapshot (non-blocking present, mi355pt.cpp pt_present_rgba8_async): when set, the resolve of the launch's LAST frame
    // also stores the pixel's new value here (same indexing as accum) — a consistent image of that frame that later frames never
    // touch, so the tone map can read it while the next launch (chained, ordered per pixel by the tags) already overwrites accum.
    float4 *snapshot;
    // Frame-fed launch (round 6; nullptr = a classic launch that knows its frames when it starts).  The launch is started with CAPACITY
    // batchFrames but may only begin the frames the host has PUBLISHED: feedHost (one host-mapped word, written by the host only) holds
    // count << 16 | display slots (below), | kFeedClosed once the host has closed the launch (then the count is final).  The launch's MONITOR
    // wavefront (below) reads it about once per microsecond and broadcasts a new word into feedBcast; a workgroup whose next ticket lies
    // beyond what it knows looks at its broadcast slot.  Wavefronts with nothing
    // to trace and no published work wait; once no new frame has come for feedIdleTicks the launch is ABANDONED with reason "idle"
    // (hand-over bound: the host repairs whatever a racing publish left unrendered), so a host that stops rendering never keeps the GPU.
    // feedDone: 8 cumulative counters (never reset; slot s at word s * kFeedDoneStride) — frame j of the launch adds its resolved pixels
    // to slot j & 7 once their stores are complete — which the launch's monitor wavefront compares with the host's running total (below).
    // FUSED DISPLAY (PostProcessing/fragment.glsl:17-26 for a host that shows every frame, MainWindow.cs:49-64): the tile pass of frame
    // j + 1 reads every pixel's value after frame j anyway (all 64 lanes, coherent) — when frame j is to be shown it tone-maps that value
    // into displayImages[slot - 1] right there (+4 % instructions; a separate tone-map kernel beside the resident wavefronts took 540
    // instead of 14 us, and a snapshot per frame costs 16 more bytes per pixel).  The low 16 bits of the feed word hold, for the 8 most
    // recent frames j, the present slot + 1 of frame j in bits 2 (j & 7) .. +1 (0 = not shown); displayPrev = that of the frame before
    // the launch's first.  The image of frame j is complete when frame j + 1 is (its count in feedDone).
    const unsigned int *feedHost;
    unsigned int *feedBcast;   // kFeedBcastSlots device words, kFeedBcastStride words apart: the feed word as the monitor last read it
    unsigned long long *feedDone;
    unsigned int feedIdleTicks;
    // ... and who tells the host that a frame is complete: ONE wavefront of the launch itself (the last one of workgroup 0: the MONITOR —
    // it takes no tiles) compares the counters with the running totals (feedBase[s] = what slot s reads once every earlier launch has
    // finished; feedPixels = pixels per frame) and stores the number of complete frames into feedHostDone, a host-mapped word the host
    // polls (pt_present_wait).  No second kernel: a gate kernel beside the resident wavefronts was dispatched up to a millisecond late.
    unsigned int *feedHostDone;
    unsigned long long feedBase[8];
    unsigned long long feedPixels;
    uchar4 *displayImages[3]; // (null for a slot that is not bound to device memory: never referenced)
    int displayPrev;
    int displayOn;            // 1 = this launch shows frames (the feed word's slot bits are looked at)
    // Hand-over audit (only compiled into the -DPT_AUDIT build, tools/handover_stress.cpp; nullptr otherwise): one 64-bit word per
    // accumulation pixel = (frames folded so far) << 32 | hash of the colour stored last, maintained with device-scope atomic
    // exchanges next to every read-modify-write of the pixel — an independent, atomics-only record of compute.glsl:126-129's
    // "one ordered read-modify-write per pixel per frame".  Violations go to auditLog (host-mapped): [0] = count, 12 words each.
    unsigned long long *audit;
    unsigned int *auditLog;
    int auditSabotage;      // audit build, tuning knob audit_sabotage = n: every n-th (pixel, frame) folds into a perturbed colour, as a stale or torn read would — proves that the audit sees it
};
// Frame tags (alpha channel inside tagged launches, pt_kernel_common.hpp): tag(f) = 2 + (f & kFrameTagMask), exact in binary32 (< 2^24).
// Two frames that can be in the image together must have different tags, and the repair pass tells "a later launch's tag" from "an
// older one" by the half window: every frame the host can have issued since the oldest launch it still remembers must lie within
// kFrameTagMask / 2 of it.  The host remembers at most kMaxUnverifiedLaunches launches of at most kMaxBatchFrames frames (the bound is
// enforced where launches are remembered, mi355pt.cpp): 128 x 256 = 32,768 frames, a sixteenth of the half window of 524,288.
// (Rounds 2 - 5 used a 1,024-frame window, which 256-frame launches of a small share could outrun: round-5 advisor finding.)
constexpr int kFrameTagMask = 0xFFFFF;
constexpr int kMaxBatchFrames = 256, kMaxUnverifiedLaunches = 128;
static_assert((long long)kMaxBatchFrames * (kMaxUnverifiedLaunches + 2) * 8 <= (kFrameTagMask + 1) / 2, "frame tags would alias within the repair window");
static_assert(kFrameTagMask + 2 < (1 << 24), "frame tags must be exact in binary32");
constexpr unsigned int kFeedClosed = 0x80000000u; // FrameArgs::feedHost: the count in the low bits is final
constexpr int kFeedBcastSlots = 256, kFeedBcastStride = 16; // FrameArgs::feedBcast: 256 words, 64 bytes apart (spread over the memory channels)
constexpr int kFeedDoneStride = 64;               // FrameArgs::feedDone: slot s lives at word s * kFeedDoneStride (512 bytes apart: its own memory channel)
constexpr int kFeedCapacity = 32;                 // frames a fed launch can take.  32, not 64: its frame table is then 256 bytes smaller than a classic launch's,
                                                  // which pays for the kernel's extra static LDS (feed queue + pixel counters: 192 bytes) — the default scene's
                                                  // workgroup sits 96 bytes below the sixth-workgroup-per-CU limit (LDS granules of 1,280 bytes)
constexpr unsigned int kAbandonContended = 1u, kAbandonIdle = 2u; // bits of the host-visible abandon flag (FrameArgs::errorWord)
constexpr int kAuditLogRecords = 1024, kAuditRecordWords = 12;
constexpr int kTileMaskWords = 8; // 64 bytes per tile (FrameArgs::tileMasks)
constexpr int kStartedWords = 4096; // capacity of FrameArgs::startedFlags (a launch with more workgroups does not report in)

// the persistent kernel reads the camera block straight from its kernarg segment (see primary_ray_cam)
static_assert(offsetof(FrameArgs, invProj) == 0 && offsetof(FrameArgs, invView) == 64 && offsetof(FrameArgs, viewPos) == 128 &&
                  offsetof(FrameArgs, focalLength) == 140 && offsetof(FrameArgs, apertureDiameter) == 144,
              "camera block layout");

struct AtmoArgs {
    float invProj[16];
    float invView[6][16];
    float lightPos[3];
    float lightIntensity;
    int size, iSteps, jSteps;
    float4 *out; // [6][size][size]
};

// ticketsConsumed: by how much the launch advances *a.queue (the caller adds it to the next launch's queueBase)
// workgroups: the grid size of the launch (persistent kernels)
// fed (in/out): in = the caller wants a frame-fed launch (a.feedHost etc. set, a.batchFrames = frames published at launch); out = whether one
// was launched (only some kernel configurations have a fed instantiation; otherwise the classic launch of a.batchFrames frames).  For a fed
// launch ticketsConsumed is the count for ZERO frames (workgroups failing tickets): the host adds feed_tickets(frames, ...) when it closes it.
hipError_t launch_integrate(const FrameArgs &a, hipStream_t stream, unsigned int *ticketsConsumed, int *workgroups = nullptr, bool *fed = nullptr);
// Hand-over repair (pt_repair_kernel): enqueued behind a join of the handle's streams, once per tagged launch since the previous join, in
// launch order — a no-op unless a.abandonWord says the launch (or one it builds on) was abandoned; then every pixel's missing frames of the
// launch described by `a` are re-rendered.  ctl = 4 device words (pairs rendered, inconsistent pixels, joins with repairs, spare).
hipError_t launch_repair(const FrameArgs &a, unsigned int *ctl, hipStream_t stream);
// ... followed by ONE of these: abandon word and ticket counters back to what a fresh chain expects
hipError_t launch_repair_done(unsigned int *abandonWord, unsigned int *queueMain, unsigned int expectMain, unsigned int *queueChain,
                              unsigned int expectChain, unsigned int *ctl, hipStream_t stream);
hipError_t launch_atmosphere(const AtmoArgs &a, hipStream_t stream);
// masks[kTileMaskWords * tile + w] for every 8x8 tile of the launch described by `a` (tilesX x tilesY tiles; see FrameArgs::tileMasks)
hipError_t launch_tile_masks(const FrameArgs &a, unsigned long long *masks, hipStream_t stream);
hipError_t launch_clear(float4 *p, size_t n, hipStream_t stream);
// alpha := 1 over n pixels (pt_write_result / pt_bind_result_buffer: alpha is the frame tag inside pipelined launches)
hipError_t launch_set_alpha(float4 *p, size_t n, hipStream_t stream);
// Multi-GPU gather, second step (group handles with block-cyclic bands): `stage` holds the compact rows of part 0, 1, ...
// (part g at pixel offset partOffset[g]); out[y] = row y of the width x height image.  bytesPerPixel = 16 (RGBA32F) or 4
// (RGBA8).  HBM-bound copy: bytesPerPixel read + written per pixel.
struct AssembleArgs {
    const void *stage;
    void *out;
    int width, height, bandRows, world, bytesPerPixel;
    unsigned long long partOffset[16]; // PT_MAX_GROUP_DEVICES
};
hipError_t launch_assemble_bands(const AssembleArgs &a, hipStream_t stream);
// ACES + gamma of PostProcessing/fragment.glsl: n RGBA32F pixels -> n RGBA8 pixels
hipError_t launch_postprocess(const float4 *accum, void *outRgba8, size_t n, hipStream_t stream);
// linearise any environment into RGBA32F for read-back
hipError_t launch_env_to_float(const void *env, int envSize, int envFormat, const float *srgbLut, float4 *out,
                               hipStream_t stream);

} // namespace pt
