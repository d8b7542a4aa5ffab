/*
 * mi355pt.h — C ABI of libmi355pt.so: the MI355X (gfx950) replacement for the reference's path-tracing
 * compute dispatch.  Plain C, cdecl, blittable arguments only, so the reference's C# host can bind every
 * entry point with [DllImport("mi355pt")] (binding shown in INTEGRATION.md).
 *
 * The reference has NO plugin/FFI interface for this path: the hot path is hard-wired GL calls inside the
 * C# class PathTracer.  Each entry point below therefore replaces one thing the host does to the shader
 * today; citations are relative to /root/reference/OpenTK-PathTracer/.
 *
 * Conventions
 *   - every function returns PT_OK (0) or a negative PT_E_* code; nothing throws across the boundary
 *     (reference: C# exceptions + console prints, src/Render/Objects/ShaderProgram.cs:25-27,70-74);
 *     pt_last_error() returns a human-readable message for the last failure on that handle
 *     (or for the last failed pt_create when handle == NULL);
 *   - the host owns every source/destination array for the duration of the call only (as with
 *     GL.NamedBufferSubData, src/Render/Objects/BufferObject.cs:37-48); the library owns all device memory;
 *   - one handle = one renderer on one GPU (pt_create) or on a GROUP of GPUs (pt_create_multi: the image is
 *     row-tiled across the devices inside the library and gathered over xGMI only when it is read or presented),
 *     callable from one thread at a time (the reference calls everything from the GameWindow thread,
 *     src/MainWindow.cs:40,72,146).  Work is enqueued on the handle's HIP stream: pt_upload_*, pt_set_* and
 *     pt_render are stream-ordered; only pt_read_*, pt_present_rgba8, pt_present_wait, pt_synchronize,
 *     pt_timer_end and pt_destroy block;
 *   - limits (PT_E_OUT_OF_RANGE beyond them): width, height <= PT_MAX_IMAGE_DIM; ray_depth <= PT_MAX_RAY_DEPTH;
 *     spp <= PT_MAX_SPP (the reference GUI offers rayDepth 1..50 and SPP 1..10, src/Render/Gui.cs:40,48);
 *   - image rows: row 0 is the BOTTOM of the image (NDC y = -1), as in the GL image the reference writes
 *     (res/shaders/PathTracing/compute.glsl:104,114); pixels are RGBA32F, alpha = 1.
 */
#ifndef MI355PT_H
#define MI355PT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define PT_API __declspec(dllexport)
#else
#define PT_API __attribute__((visibility("default")))
#endif

typedef struct pt_renderer *pt_handle;

enum {
    PT_OK = 0,
    PT_E_BAD_HANDLE = -1,
    PT_E_BAD_ARGUMENT = -2,  /* NULL pointer, bad dimensions, bad enum */
    PT_E_OUT_OF_RANGE = -3,  /* byteOffset/size outside the 144 B / 26,624 B / 464 B blobs (GL would raise INVALID_VALUE) */
    PT_E_NO_ENVIRONMENT = -4,/* pt_render before any environment map was set */
    PT_E_HIP = -5,           /* a HIP runtime call failed; message has the hipError string */
    PT_E_NO_DEVICE = -6,     /* no usable gfx950 device: the library has NO CPU fallback */
    PT_E_OUT_OF_MEMORY = -7
};

enum { PT_ENV_RGBA32F = 0, PT_ENV_SRGB8_A8 = 1 };

#define PT_BASIC_DATA_UBO_SIZE 144      /* src/MainWindow.cs:195-197 ; compute.glsl:59-64  */
#define PT_GAME_OBJECTS_UBO_SIZE 26624  /* src/MainWindow.cs:17,199-201 ; compute.glsl:66-70 */
#define PT_ATMOSPHERE_UBO_SIZE 464      /* src/Render/AtmosphericScatterer.cs:72 ; AtmosphericScattering/compute.glsl:11-15 */
#define PT_MAX_SPHERES 256
#define PT_MAX_CUBOIDS 64
#define PT_MAX_IMAGE_DIM 32767   /* pixel coordinates travel in 16-bit fields inside the kernels */
#define PT_MAX_RAY_DEPTH 4095    /* bounce counters travel in 12-bit fields */
#define PT_MAX_SPP 4095
#define PT_MAX_GROUP_DEVICES 16
#define PT_PRESENT_SLOTS 3       /* pinned host images of the non-blocking present path */

/* ---- lifetime -------------------------------------------------------------------------------------------- */

/* new PathTracer(env, width, height, ...) — src/Render/PathTracer.cs:95-110 — on HIP device `device_id`.
 * Allocates the RGBA32F accumulation image (the reference's `Result` texture, PathTracer.cs:97-99) ZEROED
 * (the reference leaves it undefined, src/Render/Objects/Texture.cs:169-197; frame 0 multiplies it by 0). */
PT_API int pt_create(int device_id, int width, int height, pt_handle *out);
PT_API int pt_destroy(pt_handle h);

/* The same renderer on n_devices GPUs of this process (no reference counterpart, the reference
 * owns one GL context): the image is tiled across device_ids[0..n) in block-cyclic bands of 8 rows = single tile rows (balanced: rows
 * near the floor cost ~2x sky rows; pt_multi_set_partition changes the band height or selects contiguous row
 * blocks), every device keeps its rows resident and renders them with global pixel coordinates, so the image is
 * bit-identical to the single-GPU one.  Nothing is exchanged per frame.  Every entry point accepts the group handle:
 * uploads / parameters / environment are replicated to all devices, pt_render enqueues on all of them, and
 * pt_read_result / pt_present_rgba8 / pt_present_rgba8_async GATHER the rows on device_ids[0] with peer copies over
 * xGMI (hipMemcpyPeerAsync; each peer has its own link to the root) before the single device-to-host copy.
 * device_ids may name the same device more than once (used to test the group path on a one-GPU box).
 * Not available on a group handle (PT_E_BAD_ARGUMENT): pt_set_tile, pt_set_interleaved_tile, pt_bind_result_buffer,
 * pt_set_stream, pt_postprocess_device.
 * Peer access: the gather wants hipDeviceCanAccessPeer == 1 in both directions between device_ids[0] and every other
 * device.  Where it is 0 (no xGMI / P2P disabled) the group is still created and renders the same bits, but its gather
 * is staged through host memory by the HIP runtime; pt_multi_gather_is_direct reports which one the handle got, and a
 * scaling measurement must refuse a staged group (bench.py and tools/multi_gpu_check.sh do). */
PT_API int pt_create_multi(const int *device_ids, int n_devices, int width, int height, pt_handle *out);
/* *out_direct = 1 when every gather copy of this handle goes over a direct peer link (always 1 for a one-GPU handle), 0 when the
 * runtime stages it through the host (see pt_create_multi). */
PT_API int pt_multi_gather_is_direct(pt_handle h, int *out_direct);
/* band_rows = 0: contiguous row blocks (device g owns rows [g*H/G, (g+1)*H/G)); else a multiple of 8: block-cyclic
 * bands of that many rows (default 8).  Resets the frame counter and zeroes, like pt_set_tile. */
PT_API int pt_multi_set_partition(pt_handle h, int band_rows);
/* Number of devices behind the handle (1 for pt_create handles). */
PT_API int pt_device_count_of(pt_handle h, int *out_n_devices);

/* PathTracer.SetSize — PathTracer.cs:131-135: frame counter = 0, image reallocated (and zeroed). The tile is
 * reset to the whole image. */
PT_API int pt_set_size(pt_handle h, int width, int height);

/* Multi-GPU row-block tiling (no reference counterpart: the reference is single-GPU).  This handle renders
 * and stores only rows [y0, y0+rows) of the width x height image; seeds and NDC still use global pixel
 * coordinates, so tiled and untiled renders are bit-identical.  Resets the frame counter and zeroes. */
PT_API int pt_set_tile(pt_handle h, int y0, int rows);

/* Block-cyclic variant of pt_set_tile for load balance across GPUs (rows near the floor cost ~2x sky rows): this handle
 * owns the bands rank, rank + world, rank + 2*world, ... of band_rows image rows each (band_rows a multiple of 8) and
 * stores them compactly, band after band.  Same bit-identical pixels; resets the frame counter and zeroes. */
PT_API int pt_set_interleaved_tile(pt_handle h, int rank, int world, int band_rows);

/* PathTracer.ResetRenderer — PathTracer.cs:137-140: frame counter = 0 (image contents are irrelevant then). */
PT_API int pt_reset(pt_handle h);

/* ---- inputs ------------------------------------------------------------------------------------------------ */

/* The six property setters NumSpheres/NumCuboids/RayDepth/SPP/FocalLength/ApertureDiameter —
 * PathTracer.cs:11-83 (GLSL uniforms compute.glsl:88-94). */
PT_API int pt_set_params(pt_handle h, int num_spheres, int num_cuboids, int ray_depth, int spp,
                         float focal_length, float aperture_diameter);

/* BasicDataUBO.SubData(offset, size, data) — src/MainWindow.cs:131-132,278-279: InvProjection@0, InvView@64,
 * ViewPos@128 (OpenTK row-major bytes, consumed as GLSL column-major). 0 <= offset, offset+size <= 144. */
PT_API int pt_upload_basic_data(pt_handle h, int byte_offset, int size, const void *src);

/* GameObjectsUBO.SubData(BufferOffset, size, data) — src/BaseSTD140Compatible.cs:12-16: std140 Spheres[256]@0
 * (80 B each), Cuboids[64]@20480 (96 B each). 0 <= offset, offset+size <= 26,624.  (The library shadows the block on the
 * host: scenes with >= 64 spheres get a sphere grid for their secondary rays, rebuilt before the next launch; results are
 * bit-identical to the reference's loop over all spheres.) */
PT_API int pt_upload_game_objects(pt_handle h, int byte_offset, int size, const void *src);

/* PathTracer.EnvironmentMap = <cube texture> — PathTracer.cs:85,118; cube creation src/MainWindow.cs:177-187,
 * src/Helper.cs:41-48.  faces[0..5] = +X,-X,+Y,-Y,+Z,-Z, each face_size^2 tightly packed RGBA texels
 * (float32 x4 for PT_ENV_RGBA32F, uint8 x4 sRGB-encoded for PT_ENV_SRGB8_A8), row 0 = t 0 as uploaded by
 * glTextureSubImage3D.  Sampling = GL LINEAR magnification with seamless cube edges (MainWindow.cs:168,178). */
PT_API int pt_set_environment(pt_handle h, int face_size, int format, const void *const faces[6]);

/* ---- the hot path ------------------------------------------------------------------------------------------ */

/* PathTracer.Render() — PathTracer.cs:114-123: enqueue one dispatch of the integrator with the current frame
 * index, then post-increment it. Asynchronous. *out_total_samples (optional) = frames * SPP after this call
 * (PathTracer.Samples, PathTracer.cs:112).
 * An upload / pt_set_params that repeats what the renderer already holds is NOT "something in between" (the reference re-uploads the
 * camera on every update, MainWindow.cs:131-132): it returns at once and changes nothing.
 * Frames of consecutive pt_render calls with nothing in between are launched as ONE pipelined kernel (up to
 * pt_set_frame_batch frames).  A frame is only held back while earlier frames of this handle are still running on the GPU
 * (so deferral never idles the device); the launch happens when the GPU has drained, when the batch is full, or at the next call of any other entry point
 * (uploads and parameter changes apply to LATER frames only, exactly as with one launch per call; every read,
 * pt_synchronize and pt_timer_* first launch what is pending).  The image is bit-identical either way.
 * Back-pressure: consecutive launches overlap on two internal streams (the second moves into the wavefront slots the first one's
 * drain frees), which needs the first one resident.  pt_render does not wait for that: a full batch whose predecessor is not resident
 * yet simply stays pending (the GPU has two launches queued, nothing idles) and is launched by a later call; only a host that runs
 * more than 16 launches ahead is held, for at most 2 ms per call.  Calls that block anyway (pt_synchronize, reads, presents) launch
 * what is pending and may wait for residency in between.  A single frame is launched that way too whenever the GPU still runs the
 * previous one.
 * pt_render cannot fail for a reason of the frame pipelining (PathTracer.cs:114-123 cannot either): a launch whose hand-over of a
 * pixel between two frames does not complete in time (a GPU shared with other work) abandons itself instead of producing a wrong
 * pixel, and the library re-renders exactly what is missing behind it before anything can observe the image. */
PT_API int pt_render(pt_handle h, int *out_total_samples);
/* Largest number of frames one launch may pipeline (1..64; 0 = back to automatic).  1 = NO deferral: every pt_render hands its frame to the
 * GPU at once (lowest latency: the interactive setting).  Since round 6 that does not mean one launch per frame: the first such
 * pt_render starts a FRAME-FED launch with room for 32 frames, and the calls that follow publish their frame into it (one store to a
 * host-mapped word; the resident wavefronts begin it as soon as they run out of earlier work), so frames of a host that renders faster
 * than the GPU pipeline like a batch (0.114 instead of 0.137 ms per 1080p frame).  Only full-size images of scenes whose kernel has a fed
 * instantiation (one sample per pixel, materials in LDS, no sphere grid) take that path; anything that changes an input or observes the
 * image closes the launch first, and a launch that waits 150 us for its next frame ends itself (the GPU is never held by a host that
 * has stopped rendering).  A handle on which this
 * was never called (or was last called with 0) pipelines up to 64 frames per launch — and up to 256 when it owns fewer than 12,000
 * tiles (a 1/8 share of a 1080p image: the fixed cost of a launch is 7 % of a 64-frame launch there); a limit set here is kept
 * exactly as given. */
PT_API int pt_set_frame_batch(pt_handle h, int max_frames);

/* What ScreenEffect.Render reads (src/MainWindow.cs:51): blocks until the stream is idle and copies this
 * handle's rows [y0, y0+rows) into dst (row_pitch_bytes >= width*16; 0 means tightly packed). */
PT_API int pt_read_result(pt_handle h, float *dst_rgba32f, size_t row_pitch_bytes);

/* The step right after the path ("next" row 1): ScreenEffect.Render(PathTracer.Result) —
 * src/Render/ScreenEffect.cs:29-37 with res/shaders/PostProcessing/fragment.glsl:17-44 — ACES tone map + gamma 2.4 into
 * an RGBA8 image (alpha 255), fused with the read-back: 4 B/pixel cross PCIe instead of 16.  Blocks like pt_read_result;
 * row_pitch_bytes >= width*4 (0 = tight).  pt_postprocess_device runs the same pass and returns the device copy (stream-
 * ordered, no host copy) so that a multi-GPU harness can gather RGBA8 rows instead of RGBA32F. */
PT_API int pt_present_rgba8(pt_handle h, uint8_t *dst_rgba8, size_t row_pitch_bytes);
PT_API int pt_postprocess_device(pt_handle h, void **out_device_ptr, size_t *out_bytes);

/* Non-blocking present for the reference's per-frame loop (src/MainWindow.cs:49-56: Render() -> PostProcesser.Render
 * (PathTracer.Result) -> blit, every frame).  pt_present_rgba8_async snapshots the image as it is after the frames
 * rendered so far: tone map (ACES + gamma, as pt_present_rgba8) into a device-side RGBA8 image of slot `slot`
 * (0 <= slot < PT_PRESENT_SLOTS), stream-ordered behind those frames, then a device-to-host copy into the library-owned
 * PINNED host image of that slot on a separate copy stream.  It returns at once; later pt_render calls only wait for
 * the (microseconds-long) tone-map pass, so frame f+1 renders while frame f's 4 B/pixel cross PCIe.
 * pt_present_wait blocks until the slot's copy has landed and returns the pinned image (tightly packed rows, row 0 =
 * bottom, valid until the next pt_present_rgba8_async on the same slot or pt_set_size/pt_destroy) and the frame index
 * (= number of accumulated frames) it shows.  Typical loop: render; present_async(f % 2); present_wait((f + 1) % 2) ->
 * upload to the GL texture -> swap.  Works on group handles (the gather of the RGBA8 rows happens on the copy stream).
 * On a single-GPU handle the presented frame comes from a SNAPSHOT that the integrator launch writes while it resolves that
 * frame's pixels, so the next pt_render does not wait for the tone map; once a host is seen to present every frame, pt_render
 * launches each frame with the snapshot attached (the image is unchanged; three internal buffers of width x rows x 16 bytes). */
PT_API int pt_present_rgba8_async(pt_handle h, int slot);
PT_API int pt_present_wait(pt_handle h, int slot, const uint8_t **out_host_rgba8, size_t *out_row_pitch_bytes,
                           int *out_frame_index);
/* Interop-style present (the library side of SURVEY section 8f row 4; no reference counterpart — the reference samples its GL
 * texture in place, src/Render/ScreenEffect.cs:29-37): make caller-owned DEVICE memory (>= rows*width*4 bytes, e.g. a GL buffer
 * registered with hipGraphicsGLRegisterBuffer and mapped) the image of present slot `slot`; NULL restores the library's own images.
 * pt_present_rgba8_async(slot) then tone-maps the frame straight into that memory and copies nothing to the host — the 8.3 MB
 * that a 1080p frame otherwise sends over PCIe; pt_present_wait(slot) returns once the image is complete there (out_host_rgba8
 * receives NULL, the frame index as usual).  Single-GPU handles only. */
PT_API int pt_present_bind_device_image(pt_handle h, int slot, void *device_rgba8, size_t bytes);

/* Resume support (no reference counterpart; the reference discards accumulation on every event): replace the
 * accumulation image of this tile and set the frame counter.  Alpha is stored as 1 whatever the source holds (the
 * reference always stores 1, compute.glsl:129; inside a pipelined launch the library uses alpha as a frame tag). */
PT_API int pt_write_result(pt_handle h, const float *src_rgba32f, size_t row_pitch_bytes, int frame_index);

PT_API int pt_get_frame_index(pt_handle h, int *out_frame_index);
/* Launch what is pending and wait for the handle's stream.  (A pipelined launch that abandoned a frame hand-over — see pt_render — has
 * been repaired when this returns: the image is the one an undisturbed run leaves; no error is reported for it.) */
PT_API int pt_synchronize(pt_handle h);

/* ---- atmosphere environment (secondary kernel) ------------------------------------------------------------- */

/* AtmosphericScatterer: UBO upload (src/Render/AtmosphericScatterer.cs:72-89: InvProjection@0 + 6 InvView@64),
 * the four uniforms (:11-57: ISteps, JSteps, lightPos from Time, LightIntensity) and Render() (:102-113),
 * into a size^2 x 6 RGBA32F cube that then becomes the environment (MainWindow.cs:189). */
PT_API int pt_atmosphere_upload_data(pt_handle h, int byte_offset, int size, const void *src);
PT_API int pt_atmosphere_render(pt_handle h, int size, int i_steps, int j_steps, const float light_pos[3],
                                float light_intensity);

/* Read back the current environment cube as RGBA32F (6 * face_size^2 * 4 floats), e.g. the atmosphere result
 * (the GUI shows / the tests check it).  sRGB environments are returned linearised. */
PT_API int pt_read_environment(pt_handle h, float *dst_rgba32f, int *out_face_size);

/* ---- plumbing for harnesses (torch.distributed, benchmarking); not part of the reference surface ----------- */

/* Device pointer + byte size of this tile's accumulation image (for zero-copy wrapping, e.g. the RCCL gather at
 * present time in multi-GPU runs). */
PT_API int pt_result_device_ptr(pt_handle h, void **out_device_ptr, size_t *out_bytes);
/* Render into caller-owned device memory (>= rows*width*16 bytes) instead of the internal image; NULL restores.
 * The buffer's RGB contents are taken as the accumulation so far (zero them for a fresh render: frame 0 multiplies
 * them by 0, and 0 * NaN is NaN as in the reference); its alpha channel is set to 1 at bind time (stream-ordered).
 * On the library's own stream the buffer must be observed through the library: while pt_render calls are outstanding its
 * alpha channel carries the frame tags of the pipelined launches, and pt_synchronize / pt_read_result /
 * pt_result_device_ptr restore alpha = 1 (the value the reference stores, compute.glsl:129) before they return.  On a
 * caller-owned stream (pt_set_stream) every frame stores alpha = 1 and stream order alone is enough. */
PT_API int pt_bind_result_buffer(pt_handle h, void *device_ptr, size_t bytes);
/* Use an existing hipStream_t (passed as void*) instead of the handle's own stream; NULL restores.  With a caller
 * stream every pt_render is enqueued on THAT stream before it returns (one launch per frame, no deferral, no
 * internal helper streams), so synchronising the stream is enough to observe the image. */
PT_API int pt_set_stream(pt_handle h, void *hip_stream);
/* hipEvent pair recorded on the handle's stream: elapsed GPU milliseconds between begin and end. */
PT_API int pt_timer_begin(pt_handle h);
PT_API int pt_timer_end(pt_handle h, float *out_milliseconds);
/* Kernel variant selector for A/B measurements; all variants produce bit-identical images.
 * 0 = default (persistent queue kernel), 1 = one wavefront per 8x8 tile, 2..6 = wave-local pools, 10+k = persistent
 * kernel with k+1 workgroups per CU. */
PT_API int pt_set_variant(pt_handle h, int variant);

PT_API const char *pt_last_error(pt_handle h);
PT_API const char *pt_version(void);
PT_API int pt_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355PT_H */
