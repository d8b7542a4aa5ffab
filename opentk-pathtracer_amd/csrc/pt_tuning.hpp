// pt_tuning.hpp — the library's tuning knobs (A/B runs, stress tools, tests).
//
// The product library reads NO environment variables: a knob only ever changes through pt_debug_set(key, value), an entry point that
// is exported but NOT declared in include/mi355pt.h (like the other pt_debug_* test aids).  One set of knobs per process, read where
// they apply (plain loads of this struct — nothing is parsed per launch); a knob affects renderers / launches created after it is set.
// Knobs are SET-UP state: set them from one thread while no other thread is inside the library (nothing here is atomic; log_launch is
// counted down by whichever thread launches).
//   Python: native.debug_set("no_sphere_grid", 1); bench.py --tune key=value; tools/handover_stress.bin --tune key=value
#pragma once
#include <cstring>
#include <type_traits>

namespace pt {

struct Tuning {
    // renderer creation (mi355pt.cpp: pt_create)
    int drainCompaction = -2;      // >= -1: overrides pt_renderer::drainCompaction (-1 = automatic)
    int batchWorkgroupsPerCU = 0;  // 1..8: workgroups per CU of pipelined launches (default 6)
    int frameBatch = 0;            // 1..64: initial pt_set_frame_batch
    int queueChunk = 0;            // 1..1024: tiles per global ticket (0 = automatic)
    int groupBand = -1;            // group handles: band height (0 = contiguous row blocks; else a multiple of 8)
    int allowStagedGather = 1;     // group handles: 1 = devices without peer access are accepted (gather staged through the host; pt_multi_gather_is_direct says so), 0 = pt_create_multi fails
    // launches (mi355pt.cpp: launch_frames)
    int auditSabotage = 0;         // -DPT_AUDIT builds only: every n-th (pixel, frame) folds into a perturbed colour
    int noSingleTagged = 0;        // 1: single frames never chain
    int serialLaunches = 0;        // 1: every tagged launch goes BEHIND its predecessor on the same stream and pt_render never defers a full batch for
                                   //    residency — launches do not overlap, so rocprofv3's per-launch durations add up to the elapsed time (profiles' cross-check)
    int feed = 1;                  // 0: never a frame-fed launch (every short launch knows its frames when it starts, rounds 1-5)
    long long feedMinTiles = 12000; // images with fewer tiles per frame never get a fed launch (tests / stress runs lower it)
    int feedLog = 0;               // debug: frame-fed launch events (open / publish / close / present) to stderr
    int feedDisplay = 0;           // 1: a host that shows every frame into bound device images gets fed launches with the FUSED DISPLAY (measured slower than the
                                   // per-frame launches + snapshot tone maps of round 3 when the host runs at most two frames ahead: DESIGN.md section 3.1; 2 = debug, no tone map)
    int feedWorkgroupsPerCU = 6;   // workgroups per CU of a fed launch (its present needs no room beside it: the display is fused into the tile pass)
    long feedIdleUs = 150;         // a fed launch whose wavefronts have waited this long for the next frame ends itself (reason "idle")
    int shortWorkgroupsPerCU = 5;  // workgroups per CU of launches of fewer than 8 frames
    long chainWaitUs = 60000;      // back-pressure: how long a launch issued by a BLOCKING entry point waits for its predecessor to become resident
    long renderWaitUs = 2000;      // ... and the most pt_render itself ever waits (only when the host is 16 launches ahead)
    long handoverBudgetMs = 500;   // hand-over bound: a result that has waited this long (wall clock) for its pixel's previous frame abandons its launch
    long handoverCheckUs = 1000;   // ... and how often a waiting wavefront looks at the abandon word and at its own waits
    // kernel selection (pt_integrate_persistent.hip: launch_integrate)
    int parkedMax = -1;            // >= 0: parked resolves per wavefront
    int noBatchPass = 0;           // 1: spp > 1 keeps the in-lane sample chain
    long long batchPassMinTiles = 16384; // pipelined spp > 1 launches over fewer tiles per frame keep the in-lane sample chain
    int parkCapacity = -1;         // >= 0: parked continuations per wavefront of the batch-pass kernel
    int parkMin = 40;              // parked continuations that make a batch pass worth running
    int noSphereGrid = 0;          // 1: large scenes keep the reference's in-order sphere loop
    int forceLeanLds = 0;          // 1: materials are always read from the UBO copy
    int tileMasks = 1;             // 0: the tile pass always culls the spheres against its own 64 rays (no cached per-tile masks)
    int logLaunch = 0;             // n > 0: the next n persistent launches print their kernel choice and LDS budget to stderr
    int carryLast = 1;             // 0: the pixel never travels with its path (every resolve loads it; the kernel of rounds 1-3)
    int gridCarry = 0;             // 1: sphere-grid scenes carry the pixel too, at five workgroups per CU (round 5 experiment)
    // sphere grid build (pt_sphere_grid.hpp)
    int gridMinSpheres = 64;       // scenes with fewer spheres get no grid
    int gridCells = 256;           // cell budget (<= ptgrid::kMaxCells)
    int gridDims[3] = {0, 0, 0};   // all > 0: the grid's resolution
};

inline Tuning &tuning()
{
    static Tuning t;
    return t;
}

// -> false for an unknown key
inline bool tuning_set(const char *key, long long v)
{
    Tuning &t = tuning();
#define PT_KNOB(name, field)                 \
    if (std::strcmp(key, name) == 0) {       \
        t.field = static_cast<std::remove_reference_t<decltype(t.field)>>(v); \
        return true;                         \
    }
    PT_KNOB("drain_compaction", drainCompaction)
    PT_KNOB("batch_wg", batchWorkgroupsPerCU)
    PT_KNOB("frame_batch", frameBatch)
    PT_KNOB("queue_chunk", queueChunk)
    PT_KNOB("group_band", groupBand)
    PT_KNOB("allow_staged_gather", allowStagedGather)
    PT_KNOB("audit_sabotage", auditSabotage)
    PT_KNOB("no_single_tagged", noSingleTagged)
    PT_KNOB("serial_launches", serialLaunches)
    PT_KNOB("short_wg", shortWorkgroupsPerCU)
    PT_KNOB("feed", feed)
    PT_KNOB("feed_idle_us", feedIdleUs)
    PT_KNOB("feed_display", feedDisplay)
    PT_KNOB("feed_log", feedLog)
    PT_KNOB("feed_wg", feedWorkgroupsPerCU)
    PT_KNOB("feed_min_tiles", feedMinTiles)
    PT_KNOB("chain_wait_us", chainWaitUs)
    PT_KNOB("render_wait_us", renderWaitUs)
    PT_KNOB("handover_budget_ms", handoverBudgetMs)
    PT_KNOB("handover_check_us", handoverCheckUs)
    PT_KNOB("parked_max", parkedMax)
    PT_KNOB("no_batch_pass", noBatchPass)
    PT_KNOB("batch_pass_min_tiles", batchPassMinTiles)
    PT_KNOB("park_capacity", parkCapacity)
    PT_KNOB("park_min", parkMin)
    PT_KNOB("no_sphere_grid", noSphereGrid)
    PT_KNOB("force_lean_lds", forceLeanLds)
    PT_KNOB("carry_last", carryLast)
    PT_KNOB("grid_carry", gridCarry)
    PT_KNOB("log_launch", logLaunch)
    PT_KNOB("tile_masks", tileMasks)
    PT_KNOB("grid_min_spheres", gridMinSpheres)
    PT_KNOB("grid_cells", gridCells)
    PT_KNOB("grid_dim_x", gridDims[0])
    PT_KNOB("grid_dim_y", gridDims[1])
    PT_KNOB("grid_dim_z", gridDims[2])
#undef PT_KNOB
    return false;
}

} // namespace pt
