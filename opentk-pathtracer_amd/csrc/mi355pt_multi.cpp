// mi355pt_multi.cpp — group handles: ONE renderer row-tiled over several GPUs of this process (pt_create_multi).
//
// The reference is single-GPU (one GL context, src/Render/PathTracer.cs:95-123); this is the scale-out SURVEY.md
// section 8b/8e specifies behind the C ABI.  The integrator has no inter-pixel dependency — a pixel depends only on
// (x, y, frame, W, H) and read-only inputs (compute.glsl:104-129) — so every device keeps its rows of the accumulation
// image resident and nothing is exchanged per frame.  Only reading / presenting the image communicates: the parts' rows
// are pulled to the root device (device_ids[0]) with hipMemcpyPeerAsync — over xGMI every peer has its own link to the
// root, so the copies of the G-1 peers run concurrently — un-banded there by one small kernel when ownership is
// block-cyclic, and leave through ONE device-to-host copy.  A part is an ordinary single-GPU handle (mi355pt.cpp) with a
// tile; every entry point of the C ABI replicates to the parts (PT_FAN_OUT in mi355pt.cpp).
#include "pt_renderer.hpp"

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using ptimpl::fail;
using ptimpl::hip_fail;
using ptimpl::PresentSlot;

namespace {

int root_device(pt_handle g)
{
    PT_HIP(g, hipSetDevice(g->device));
    return PT_OK;
}

int part_fail(pt_handle g, pt_handle part, int rc) { return fail(g, rc, part->error); }

bool banded(pt_handle g) { return g->parts[0]->bandRows > 0; }

// image rows owned by part `i`, in the order it stores them
void owned_rows(pt_handle g, int i, std::vector<int> &rows)
{
    pt_handle p = g->parts[i];
    rows.clear();
    if (p->bandRows == 0) {
        for (int r = 0; r < p->rows; r++) rows.push_back(p->y0 + r);
        return;
    }
    for (long long b = p->bandRank; b * p->bandRows < g->height; b += p->bandWorld)
        for (long long y = b * p->bandRows; y < (b + 1) * p->bandRows && y < g->height; y++) rows.push_back((int)y);
}

// (Re-)tile the parts: block-cyclic bands of g->groupBand rows, or contiguous row blocks when groupBand == 0 or the image
// is too small for every device to own a band.  Resets the frame counter and zeroes (pt_set_tile semantics).
int apply_partition(pt_handle g)
{
    const int G = (int)g->parts.size();
    if (g->height < G) return fail(g, PT_E_BAD_ARGUMENT, "image has fewer rows than the group has devices");
    if (G == 1) return PT_OK; // the single part owns the whole image (pt_create / pt_set_size state)
    const bool bands = g->groupBand > 0 && (long long)g->groupBand * G <= g->height;
    for (int i = 0; i < G; i++) {
        int rc;
        if (bands) {
            rc = pt_set_interleaved_tile(g->parts[i], i, G, g->groupBand);
        } else {
            int y0 = (int)((long long)i * g->height / G), y1 = (int)((long long)(i + 1) * g->height / G);
            rc = pt_set_tile(g->parts[i], y0, y1 - y0);
        }
        if (rc != PT_OK) return part_fail(g, g->parts[i], rc);
    }
    g->frame = 0;
    return PT_OK;
}

int ensure_buffer(pt_handle g, void **buf, size_t *cap, size_t need, hipStream_t usedOn)
{
    if (need <= *cap) return PT_OK;
    if (*buf) {
        PT_HIP(g, hipStreamSynchronize(usedOn));
        PT_HIP(g, hipFree(*buf));
    }
    *buf = nullptr;
    *cap = 0;
    PT_HIP(g, hipMalloc(buf, need));
    *cap = need;
    return PT_OK;
}

// Pull the parts' rows (src[i] = compact rows of part i on its device, bpp bytes per pixel; part i's rows may be read
// once ready[i] has fired) into the assembled width x height image `full` on the root device, all on root-device
// stream `st`.  `stage` / `stageCap` = the staging buffer to use for block-cyclic ownership.
int gather_on_root(pt_handle g, const std::vector<const void *> &src, const std::vector<hipEvent_t> &ready, int bpp,
                   void *full, void **stage, size_t *stageCap, hipStream_t st)
{
    const int G = (int)g->parts.size();
    if (int rc = root_device(g)) return rc;
    const size_t rowBytes = (size_t)g->width * bpp;
    const bool bands = banded(g);
    if (bands)
        if (int rc = ensure_buffer(g, stage, stageCap, (size_t)g->height * rowBytes, st)) return rc;
    pt::AssembleArgs aa;
    size_t off = 0; // in pixels
    for (int i = 0; i < G; i++) {
        pt_handle p = g->parts[i];
        PT_HIP(g, hipStreamWaitEvent(st, ready[i], 0));
        char *dst = bands ? (char *)*stage + off * bpp : (char *)full + (size_t)p->y0 * rowBytes;
        PT_HIP(g, hipMemcpyPeerAsync(dst, g->device, src[i], p->device, (size_t)p->rows * rowBytes, st));
        aa.partOffset[i] = off;
        off += (size_t)p->rows * g->width;
    }
    if (bands) {
        aa.stage = *stage;
        aa.out = full;
        aa.width = g->width;
        aa.height = g->height;
        aa.bandRows = g->parts[0]->bandRows;
        aa.world = G;
        aa.bytesPerPixel = bpp;
        PT_HIP(g, pt::launch_assemble_bands(aa, st));
    }
    return PT_OK;
}

// every part: launch what is pending, then mark on its stream the point from which its accumulation rows may be read
int parts_ready_accum(pt_handle g, std::vector<const void *> &src, std::vector<hipEvent_t> &ready)
{
    for (pt_handle p : g->parts) {
        if (int rc = ptimpl::bind_device(p)) return part_fail(g, p, rc);
        if (int rc = ptimpl::fix_alpha(p)) return part_fail(g, p, rc); // (chained launches leave frame tags in alpha)
        PT_HIP(g, hipEventRecord(p->gatherReady, p->stream));
        src.push_back(p->accum());
        ready.push_back(p->gatherReady);
    }
    return PT_OK;
}

// (a part's rows were read behind a join of its streams: the hand-over repair passes ran in front of the read — see join_stripes)
int parts_check_handover(pt_handle g)
{
    for (pt_handle p : g->parts) {
        if (int rc = ptimpl::bind_device(p)) return part_fail(g, p, rc);
        PT_HIP(g, hipStreamSynchronize(p->stream));
        if (int rc = ptimpl::settle_handover(p)) return part_fail(g, p, rc);
    }
    return PT_OK;
}

} // namespace

namespace ptimpl {

int group_destroy(pt_handle g)
{
    (void)hipSetDevice(g->device);
    if (g->copyStream) (void)hipStreamSynchronize(g->copyStream);
    if (g->ownStream) (void)hipStreamSynchronize(g->ownStream);
    for (pt_handle p : g->parts)
        if (p) (void)pt_destroy(p);
    g->parts.clear();
    (void)hipSetDevice(g->device);
    free_slots(g);
    if (g->dGatherFull) (void)hipFree(g->dGatherFull);
    if (g->dGatherStage) (void)hipFree(g->dGatherStage);
    if (g->dAsyncStage) (void)hipFree(g->dAsyncStage);
    if (g->copyStream) (void)hipStreamDestroy(g->copyStream);
    if (g->ownStream) (void)hipStreamDestroy(g->ownStream);
    g->magic = 0;
    delete g;
    return PT_OK;
}

int group_set_size(pt_handle g, int width, int height)
{
    if (height < (int)g->parts.size()) return fail(g, PT_E_BAD_ARGUMENT, "image has fewer rows than the group has devices");
    if (int rc = root_device(g)) return rc;
    PT_HIP(g, hipStreamSynchronize(g->copyStream)); // presents of the old size have left the device
    for (PresentSlot &s : g->slots) s.inFlight = false;
    for (pt_handle p : g->parts) {
        int rc = pt_set_size(p, width, height); // PathTracer.cs:131-135 on every device (whole image, frame 0, zeroed)
        if (rc != PT_OK) return part_fail(g, p, rc);
    }
    g->width = width;
    g->height = height;
    g->rows = height;
    g->frame = 0;
    return apply_partition(g);
}

int group_read_result(pt_handle g, float *dst, size_t row_pitch_bytes)
{
    std::vector<const void *> src;
    std::vector<hipEvent_t> ready;
    if (int rc = parts_ready_accum(g, src, ready)) return rc;
    if (int rc = root_device(g)) return rc;
    const size_t rowBytes = (size_t)g->width * 16;
    if (int rc = ensure_buffer(g, &g->dGatherFull, &g->gatherFullBytes, (size_t)g->height * rowBytes, g->stream)) return rc;
    if (int rc = gather_on_root(g, src, ready, 16, g->dGatherFull, &g->dGatherStage, &g->gatherStageBytes, g->stream)) return rc;
    PT_HIP(g, hipMemcpy2DAsync(dst, row_pitch_bytes, g->dGatherFull, rowBytes, rowBytes, (size_t)g->height, hipMemcpyDeviceToHost,
                               g->stream));
    PT_HIP(g, hipStreamSynchronize(g->stream)); // the parts' rows have been read: they may render on
    return parts_check_handover(g);
}

int group_result_device_ptr(pt_handle g, void **out_ptr, size_t *out_bytes)
{
    std::vector<const void *> src;
    std::vector<hipEvent_t> ready;
    if (int rc = parts_ready_accum(g, src, ready)) return rc;
    if (int rc = root_device(g)) return rc;
    const size_t bytes = (size_t)g->height * g->width * 16;
    if (int rc = ensure_buffer(g, &g->dGatherFull, &g->gatherFullBytes, bytes, g->stream)) return rc;
    if (int rc = gather_on_root(g, src, ready, 16, g->dGatherFull, &g->dGatherStage, &g->gatherStageBytes, g->stream)) return rc;
    PT_HIP(g, hipStreamSynchronize(g->stream));
    if (out_ptr) *out_ptr = g->dGatherFull; // a gathered COPY on the root device, valid until the next read / present / resize
    if (out_bytes) *out_bytes = bytes;
    return PT_OK;
}

int group_write_result(pt_handle g, const float *src, size_t row_pitch_bytes, int frame_index)
{
    std::vector<int> rows;
    std::vector<float> compact;
    const size_t rowFloats = (size_t)g->width * 4;
    for (int i = 0; i < (int)g->parts.size(); i++) {
        owned_rows(g, i, rows);
        compact.resize(rows.size() * rowFloats);
        for (size_t r = 0; r < rows.size(); r++)
            std::memcpy(compact.data() + r * rowFloats, (const char *)src + (size_t)rows[r] * row_pitch_bytes, rowFloats * 4);
        int rc = pt_write_result(g->parts[i], compact.data(), 0, frame_index);
        if (rc != PT_OK) return part_fail(g, g->parts[i], rc);
    }
    g->frame = frame_index;
    return PT_OK;
}

int group_present_rgba8(pt_handle g, uint8_t *dst, size_t row_pitch_bytes)
{
    // every device tone-maps its own rows (PostProcessing/fragment.glsl:17-26): the gather moves 4 B per pixel, not 16
    std::vector<const void *> src;
    std::vector<hipEvent_t> ready;
    for (pt_handle p : g->parts) {
        void *d = nullptr;
        int rc = pt_postprocess_device(p, &d, nullptr);
        if (rc != PT_OK) return part_fail(g, p, rc);
        PT_HIP(g, hipEventRecord(p->gatherReady, p->stream));
        src.push_back(d);
        ready.push_back(p->gatherReady);
    }
    if (int rc = root_device(g)) return rc;
    const size_t rowBytes = (size_t)g->width * 4;
    if (int rc = ensure_buffer(g, &g->dGatherFull, &g->gatherFullBytes, (size_t)g->height * rowBytes, g->stream)) return rc;
    if (int rc = gather_on_root(g, src, ready, 4, g->dGatherFull, &g->dGatherStage, &g->gatherStageBytes, g->stream)) return rc;
    PT_HIP(g, hipMemcpy2DAsync(dst, row_pitch_bytes, g->dGatherFull, rowBytes, rowBytes, (size_t)g->height, hipMemcpyDeviceToHost,
                               g->stream));
    PT_HIP(g, hipStreamSynchronize(g->stream));
    return parts_check_handover(g);
}

int group_present_async(pt_handle g, int slot)
{
    if (int rc = root_device(g)) return rc;
    PresentSlot &gs = g->slots[slot];
    const size_t pixels = (size_t)g->width * g->height;
    if (int rc = ensure_slot_events(g, slot)) return rc;
    if (int rc = ensure_slot_device(g, slot, pixels)) return rc; // the assembled RGBA8 image on the root device
    if (int rc = ensure_slot_host(g, slot, pixels)) return rc;
    std::vector<const void *> src;
    std::vector<hipEvent_t> ready;
    for (pt_handle p : g->parts) {
        if (int rc = ptimpl::bind_device(p)) return part_fail(g, p, rc);
        PresentSlot &ps = p->slots[slot];
        if (int rc = ensure_slot_events(p, slot)) return part_fail(g, p, rc);
        if (int rc = ensure_slot_device(p, slot, p->tilePixels())) return part_fail(g, p, rc);
        if (int rc = ptimpl::join_stripes(p)) return part_fail(g, p, rc);
        // the root's previous pull from this part's slot image must have finished before it is overwritten
        if (gs.inFlight) PT_HIP(g, hipStreamWaitEvent(p->stream, gs.copied, 0));
        PT_HIP(g, pt::launch_postprocess(p->accum(), ps.dRgba8, p->tilePixels(), p->stream));
        PT_HIP(g, hipEventRecord(ps.toneMapped, p->stream));
        src.push_back(ps.dRgba8);
        ready.push_back(ps.toneMapped);
    }
    // gather + un-band + device-to-host copy on the root's COPY stream: the parts render on meanwhile
    if (int rc = gather_on_root(g, src, ready, 4, gs.dRgba8, &g->dAsyncStage, &g->asyncStageBytes, g->copyStream)) return rc;
    PT_HIP(g, hipMemcpyAsync(gs.host, gs.dRgba8, pixels * 4, hipMemcpyDeviceToHost, g->copyStream));
    PT_HIP(g, hipEventRecord(gs.copied, g->copyStream));
    gs.inFlight = true;
    gs.valid = false;
    gs.frame = g->parts[0]->frame;
    gs.rows = g->height;
    gs.width = g->width;
    return PT_OK;
}

int group_timer_end(pt_handle g, float *out_ms)
{
    float worst = 0.0f;
    for (pt_handle p : g->parts) {
        float ms = 0.0f;
        int rc = pt_timer_end(p, &ms);
        if (rc != PT_OK) return part_fail(g, p, rc);
        if (ms > worst) worst = ms;
    }
    *out_ms = worst; // the frame is done when the slowest device is
    return PT_OK;
}

} // namespace ptimpl

extern "C" {

PT_API int pt_create_multi(const int *device_ids, int n_devices, int width, int height, pt_handle *out)
{
    if (!out) return fail(nullptr, PT_E_BAD_ARGUMENT, "out == NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > PT_MAX_GROUP_DEVICES)
        return fail(nullptr, PT_E_BAD_ARGUMENT, "need 1..PT_MAX_GROUP_DEVICES device ids");
    if (width <= 0 || height <= 0) return fail(nullptr, PT_E_BAD_ARGUMENT, "width/height must be positive");
    if (height < n_devices) return fail(nullptr, PT_E_BAD_ARGUMENT, "image has fewer rows than the group has devices");
    pt_renderer *g = new (std::nothrow) pt_renderer();
    if (!g) return fail(nullptr, PT_E_OUT_OF_MEMORY, "host allocation failed");
    g->width = width;
    g->height = height;
    g->rows = height;
    g->device = device_ids[0];
    for (int i = 0; i < n_devices; i++) {
        pt_handle p = nullptr;
        int rc = pt_create(device_ids[i], width, height, &p); // validates the id, the size limits and the device
        if (rc != PT_OK) {
            std::string msg = pt_last_error(nullptr);
            ptimpl::group_destroy(g);
            return fail(nullptr, rc, msg);
        }
        g->parts.push_back(p);
    }
#define PT_GROUP_HIP(call)                                                                                             \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            int rc_ = hip_fail(nullptr, e_, #call);                                                                    \
            ptimpl::group_destroy(g);                                                                                  \
            return rc_;                                                                                                \
        }                                                                                                              \
    } while (0)
    PT_GROUP_HIP(hipSetDevice(g->device));
    PT_GROUP_HIP(hipStreamCreateWithFlags(&g->ownStream, hipStreamNonBlocking));
    g->stream = g->ownStream;
    PT_GROUP_HIP(hipStreamCreateWithFlags(&g->copyStream, hipStreamNonBlocking));
#undef PT_GROUP_HIP
    // Direct xGMI copies between the root and every peer.  Without peer access hipMemcpyPeerAsync stages the gather through host memory:
    // the group still WORKS (a CI box or a workstation without P2P renders the same bits), but it is not what a SCALE number may come
    // from — the handle remembers it (pt_multi_gather_is_direct) and bench.py / tools/multi_gpu_check.sh refuse a staged group.  Tuning
    // knob allow_staged_gather = 0 turns the fallback into the hard error it was in round 5.
    for (int i = 1; i < n_devices; i++) {
        const int d = device_ids[i];
        if (d == g->device) continue;
        int canRootToPeer = 0, canPeerToRoot = 0;
        const hipError_t e1 = hipDeviceCanAccessPeer(&canRootToPeer, g->device, d), e2 = hipDeviceCanAccessPeer(&canPeerToRoot, d, g->device);
        if (e1 != hipSuccess || e2 != hipSuccess || !canRootToPeer || !canPeerToRoot) {
            (void)hipGetLastError();
            g->gatherDirect = false;
            if (pt::tuning().allowStagedGather == 0) {
                const std::string msg = "pt_create_multi: no peer access between device " + std::to_string(g->device) + " and device " + std::to_string(d) +
                                        " (hipDeviceCanAccessPeer = 0): the gather would be staged through the host";
                ptimpl::group_destroy(g);
                return fail(nullptr, PT_E_HIP, msg);
            }
        }
        if (canRootToPeer) {
            (void)hipSetDevice(g->device);
            (void)hipDeviceEnablePeerAccess(d, 0); // hipErrorPeerAccessAlreadyEnabled is fine
        }
        if (canPeerToRoot) {
            (void)hipSetDevice(d);
            (void)hipDeviceEnablePeerAccess(g->device, 0);
        }
        (void)hipGetLastError();
    }
    if (const int v = pt::tuning().groupBand; v == 0 || (v >= 8 && (v & 7) == 0)) g->groupBand = v; // (tuning knob, pt_tuning.hpp)
    int rc = apply_partition(g);
    if (rc != PT_OK) {
        std::string msg = g->error;
        ptimpl::group_destroy(g);
        return fail(nullptr, rc, msg);
    }
    *out = g;
    return PT_OK;
}

PT_API int pt_multi_gather_is_direct(pt_handle h, int *out_direct)
{
    PT_CHECK_HANDLE(h);
    if (!out_direct) return fail(h, PT_E_BAD_ARGUMENT, "out == NULL");
    *out_direct = (!h->isGroup() || h->gatherDirect) ? 1 : 0;
    return PT_OK;
}

PT_API int pt_multi_set_partition(pt_handle h, int band_rows)
{
    PT_CHECK_HANDLE(h);
    if (!h->isGroup()) return fail(h, PT_E_BAD_ARGUMENT, "not a group handle (pt_create_multi)");
    if (band_rows < 0 || (band_rows & 7)) return fail(h, PT_E_BAD_ARGUMENT, "band_rows must be 0 or a positive multiple of 8");
    h->groupBand = band_rows;
    // presents of the old partition must have left the parts' slot images before their buffers are re-tiled (a part's row count
    // can grow, and ensure_slot_device would free an image the root's copy stream is still pulling from)
    if (int rc = root_device(h)) return rc;
    PT_HIP(h, hipStreamSynchronize(h->copyStream));
    for (PresentSlot &s : h->slots) s.inFlight = false;
    if ((int)h->parts.size() == 1) { // the single part keeps the whole image: frame counter = 0 and zeroed, as documented
        int rc = pt_set_tile(h->parts[0], 0, h->height);
        if (rc != PT_OK) return part_fail(h, h->parts[0], rc);
        h->frame = 0;
        return PT_OK;
    }
    return apply_partition(h);
}

} // extern "C"
