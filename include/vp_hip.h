/*
 * vp_hip.h -- C ABI of the MI355X-native blob-detection hot path (libvp_hip.so).
 *
 * This is the drop-in boundary for the reference's per-frame GPU chain
 *     raw2quad -> resampling -> gradientDot -> satHorizontal/satVertical -> satBlobCenter -> blobList
 * (SURVEY.md section 8).  Every entry point cites the reference interface it replaces (file:line under the reference
 * tree).  Signatures use only plain pointers, sizes and PODs; all image pointers are DEVICE pointers to tightly
 * packed row-major data; `batch` frames are stored back to back.  All launches are asynchronous on the context's
 * HIP stream `stream` (0 <= stream < n_streams); nothing here synchronises except vp_sync and the map helpers.
 *
 * Errors: every function returns VP_OK (0) or a negative VpStatus and stores a message retrievable with
 * vp_last_error() (thread-local).  The C++ facade above this ABI turns any non-zero status into the reference's
 * FATAL() behaviour (log.h:21: stderr + exit(1)).
 *
 * Threads: a context is thread-COMPATIBLE, not thread-safe -- launches, tuning and synchronisation of one VpContext come from one thread at a
 * time (the reference launches from its main thread only); different contexts may be driven by different threads at once
 * (host/feeds_multi.cpp: one per device).  The exceptions are the blocking map / unmap helpers of VpBuffer / VpImage and vp_image_acquire /
 * vp_image_release, which other threads may call while the owner launches (the reference's RTP and snapshot threads do, rtpstreamer.cpp:177-181,
 * snapshotwriter.cpp:52-54).
 *
 * The library fails loudly when no HIP device is usable; there is no CPU fallback of any kind.
 */
#ifndef VP_HIP_H
#define VP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	VP_OK = 0,
	VP_ERR_INVALID = -1,     /* bad argument / unsupported shape */
	VP_ERR_HIP = -2,         /* a HIP runtime call failed (message holds hipGetErrorString) */
	VP_ERR_NOMEM = -3,
	VP_ERR_TIMEOUT = -4,     /* a bounded in-kernel wait expired (single-pass SAT look-back) */
	VP_ERR_NODEVICE = -5
} VpStatus;

/* PixelFormat::RGGB8 / GRBG8 / BGR8 (opencl.h:41-45, opencl.cpp:29-31: the -DRGGB/-DGRBG/-DBGR kernel options) */
typedef enum { VP_LAYOUT_RGGB = 0, VP_LAYOUT_GRBG = 1, VP_LAYOUT_BGR = 2 } VpLayout;

/* Behaviour of read_imageui through CLK_FILTER_LINEAR on an integer image (resampling.cl:50), which OpenCL leaves
 * undefined.  NEAREST: texel floor(u), clamp-to-edge.  BILINEAR_RTZ: spec 8.2 linear formula in binary32,
 * truncated toward zero.  Recorded with every parity result. */
typedef enum { VP_SAMPLER_NEAREST = 0, VP_SAMPLER_BILINEAR_RTZ = 1 } VpSamplerMode;

/* Summed-area-table implementations.  All produce the same int32 (wrap-around) table; box sums taken from it are
 * exact, and equal the reference's binary32 SAT whenever max|SAT| < 2^24.
 *   TILED    : single pass, LDS-tiled local scans + decoupled look-back of tile carries (8 B/px HBM traffic)
 *   TWOPASS  : global row pass then column pass ("global-recursive" comparison variant, 16 B/px)
 *   LOCAL    : vp_frame / vp_flat_bloblist only -- no table at all: gradientDot, sliding box sums of it (what circle takes from the SAT), circle and blobList in ONE kernel that
 *              reads flat once (the reference keeps its SAT private to rgba2blobCenter, Resources.cpp:155-163).  VpFrameBuffers.sat, .graddot
 *              and .blobcenter must be NULL.  Same blob list, bit for bit; TILED stays the default and the pass the roofline is quoted on. */
typedef enum { VP_SAT_TILED = 0, VP_SAT_TWOPASS = 1, VP_SAT_LOCAL = 2 } VpSatVariant;

/* Order of the blob list.  ATOMIC = append order of the device atomics, as in blobList.cl:87 (run-dependent);
 * RASTER = records sorted by pixel index (deterministic whenever counter[0] <= max_matches). */
typedef enum { VP_ORDER_ATOMIC = 0, VP_ORDER_RASTER = 1 } VpListOrder;

/* CLCameraModel, Perspective.h:22-29 == resampling.cl:20-27.  72 bytes, packed, passed by value to the kernel.
 * `f` is the focal length in quad-plane pixels (Perspective.cpp:140); shape is carried but unused (resampling.cl:29-47). */
typedef struct __attribute__((packed)) {
	int32_t shape[2];
	float f;
	float p[2];
	float d;
	float r[9]; /* row-major field->image rotation */
	float c[3]; /* camera position, field coordinates */
} VpCameraModel;

/* CLMatch, main.cpp:33-41 == Match in blobList.cl:26-32.  22 bytes, packed (floats at offsets 0,4,14,18 are unaligned). */
typedef struct __attribute__((packed)) {
	float x, y;        /* sub-pixel flat coordinates */
	uint8_t color[3];  /* mean dRGB over the disc */
	uint8_t center[3]; /* dRGB at the peak pixel */
	float circ;
	float score;
} VpMatch;

typedef struct VpContext VpContext;

/* ABI revision of this header and of the library that was built from it: bumped whenever a struct the library WRITES into caller memory grows (VpPlacement
 * gained `fast_gbs` in revision 2) or an entry point changes meaning.  A caller compiled against an older header checks vp_abi_version() == VP_ABI_VERSION
 * before it hands the library arrays of such structs; vp_abi_sizeof names the structs' sizes as the library sees them ("VpPlacement", "VpFrameParams",
 * "VpFrameBuffers", "VpCameraModel", "VpMatch", "VpStageTime"; 0 = unknown name). */
#define VP_ABI_VERSION 3
int vp_abi_version(void);
size_t vp_abi_sizeof(const char* struct_name);

/* ---- context / runtime: replaces OpenCL::OpenCL, wait, printRuntimes, clearEvents (opencl.h:72-112, opencl.cpp:34-101)
 * vp_ctx_create picks HIP device `device`, creates n_streams in-order streams (the reference uses one in-order queue,
 * opencl.cpp:48).  vp_ctx_adopt_stream wraps an existing hipStream_t (e.g. a PyTorch stream) as stream 0. */
int vp_ctx_create(int device, int n_streams, VpContext** out);
int vp_ctx_adopt_stream(int device, void* hip_stream, VpContext** out);
void vp_ctx_destroy(VpContext* ctx);
int vp_sync(VpContext* ctx, int stream);              /* OpenCL::wait (opencl.cpp:87-92); stream < 0 = all */
/* clFlush: makes sure everything enqueued on the stream so far has been handed to the device, without waiting for it (hipStreamQuery).  Returns VP_OK
 * whether or not the work has finished; stream < 0 = all. */
int vp_flush(VpContext* ctx, int stream);
/* Ordering against streams the library does not own -- the device-side counterpart of the cl::Event every OpenCL::run returns (opencl.h:76-86), which a
 * cl::EnqueueArgs can wait for; the reference itself has one in-order queue (opencl.cpp:48) plus map / unmap from its RTP and snapshot threads.  A context's own streams are
 * non-blocking, so a buffer the caller fills on another stream (hipMemcpyAsync, a kernel, torch) is NOT ordered before a vp_* call on a library stream.
 * vp_stream_wait_for: everything enqueued so far on `producer_stream` (a hipStream_t; NULL = the legacy default stream of the context's device)
 * completes before anything enqueued from now on on the context's `stream`.  vp_stream_release_to: the reverse edge, for a consumer stream that reads
 * the results.  One event each, no host wait; a no-op when both are the same stream (vp_ctx_adopt_stream).  Streams of another device are fine. */
int vp_stream_wait_for(VpContext* ctx, int stream, void* producer_stream);
int vp_stream_release_to(VpContext* ctx, int stream, void* consumer_stream);
const char* vp_last_error(void);
int vp_device_count(void);
const char* vp_device_name(VpContext* ctx);
/* Result-neutral tuning (every value gives bit-identical output; the library reads no environment variable):
 *   "sat_workgroups_per_cu" 2..8 (0 = automatic, the default), "resample_xcd_bands" -1 automatic / 0 / 1, "resample_groups_per_wg" 0..64 (default 0 = automatic: 16-frame groups a resampling workgroup takes with one set-up), "resample_float_stage" 0 / 1 / 2 (default 2: the staged Bayer kernel parks the footprint box in LDS as floats, two frames side by side; 1: floats; 0: bytes),
 *   "resample_bgr_staged" 0 / 1 (default 1), "candidate_capacity" n (0 = max(4 * max_matches, 4096); a frame with more candidates
 *   is rescanned, still exact), "tail_segment_rows" n (0 = automatic), "tail_emit_kernel" 0 / 1 (default 1: the walkers' survivors become records in a second
 *   launch, one workgroup per frame; 0: inside the walker), "sat_halo_strips" 0 / 1 (default 1: vp_frame lets the resampling kernel
 *   write the dense halo strips the tiled SAT pass reads), "placement_max_mib" n (default 0: MiB a placement search may hold at once, 0 = half of the free memory but at most a quarter of the device; -1 = the calling process owns the GPU: nine tenths of the free memory), "placement_candidates" 0..256 (default 96: blocks vp_malloc_device_placed may probe per
 *   request; 0 or 1 = take what hipMalloc gives), "frame_graph" 0 / 1 / 2 (default 0; 1: vp_frame captures the launches of a (stream, parameters, batch,
 *   buffers) call into a hipGraph when the call follows itself on the stream and replays the graph from then on; a stream whose state belongs to another
 *   tuple -- another launch shape, another geometry -- takes the ordinary path, which repairs it; calls with stage events enabled or debug views always
 *   do.  2 = the same without the pre-flight check: every repeated tuple attempts a capture, and a capture that meets a repair is refused and undone --
 *   slower, for tests of that path), "sat_spin_limit" n (0 = 2^22: polls a look-back wait of the single-pass SAT may take before the launch gives up --
 *   vp_sync then returns VP_ERR_TIMEOUT and the stream's next SAT launch resets the carry state; seconds by default, never reached in a healthy run).
 *   Applies to calls issued after it.
 * NOT result-neutral, for tests of the recovery path only: "sat_debug_drop_tile" t (-1 = off): the single-pass SAT withholds the column carry of tile t
 *   (frame * tiles per frame + row * tiles per row + column) of its launches, so that the tile below waits for the limit above and the launch times out. */
int vp_ctx_tune(VpContext* ctx, const char* key, int value);
/* "frame_graph" bookkeeping since the context was created: out = {captures made, replays launched, captures refused and undone, graphs dropped because the
 * stream's state had moved}. */
int vp_frame_graph_stats(VpContext* ctx, int64_t out[4]);

/* per-launch timing, the analogue of the stored cl::Event list (opencl.h:84, opencl.cpp:94-105).  When enabled every
 * stage launch is bracketed by hipEvents; vp_event_times syncs the stream and returns up to `cap` {tag, ms} pairs. */
typedef struct { const char* stage; float ms; } VpStageTime;
int vp_events_enable(VpContext* ctx, int on);
int vp_event_times(VpContext* ctx, VpStageTime* out, int cap);  /* printRuntimes */
int vp_events_clear(VpContext* ctx);                            /* clearEvents   */

/* ---- memory: replaces clAlloc / CLArray / CLImage allocation and the blocking map/unmap RAII (opencl.cpp:137-159,
 * opencl.h:115-165,215-262).  Device memory is plain hipMalloc; host memory is pinned (the CL_MEM_ALLOC_HOST_PTR analogue).
 * The copies are asynchronous on `stream`; the *_sync forms also wait, like a blocking clEnqueueMapBuffer. */
int vp_malloc_device(VpContext* ctx, size_t bytes, void** dptr);
int vp_free_device(VpContext* ctx, void* dptr);
int vp_malloc_host(VpContext* ctx, size_t bytes, void** hptr);
int vp_free_host(VpContext* ctx, void* hptr);
/* Placement probe: mean microseconds over `reps` launches (after one untimed) of a pure 16-byte-per-lane read stream (write = 0) or zero-fill
 * stream (write = 1: the contents are lost) over [ptr, ptr + bytes); ptr 16-byte aligned.  Synchronises the stream. */
int vp_probe_stream(VpContext* ctx, int stream, void* ptr, size_t bytes, int write, int reps, float* us);
/* Placement-checked device memory.  On MI355X a large hipMalloc block takes write streams at one of two rates for as long as it lives (4.95-5.4 or
 * 3.8-4.4 TB/s zero-fill; reads are not affected), decided by where the allocation landed.  The reference allocates its images once and keeps them
 * (OpenCL::acquire pool, opencl.cpp:108-122; CLArray, opencl.cpp:137-159), so the draw is made once: vp_malloc_device_placed allocates up to
 * "placement_candidates" (vp_ctx_tune, default 96) blocks without freeing, times a zero-fill of each (vp_probe_stream), hands out the `count`
 * fastest (dptrs[0] the fastest) and frees the rest.  It stops early once `count` blocks (vp_frame_buffers_create: ONE, the table's -- flat is written by a kernel that is not HBM-bound and only read by the
 * SAT pass, and reads do not depend on the placement) write at 0.70 or more of the read-stream rate it measured on
 * the first candidate (a ratio of two measurements of this device, not an MI355X constant: fast blocks reach 0.74-0.81, slow ones 0.57-0.66), or once
 * sixteen candidates lie within 6 % of each other; it never holds more than half of the free device memory and a quarter of the device, or "placement_max_mib" MiB (vp_ctx_tune,
 * default 0 = that rule), and it asks the device before every candidate and stops while an eighth of the device is still free, so that a second context or
 * process on the GPU keeps room (its hipMalloc must not fail because this search is shopping).  A process that owns its GPU says so with
 * "placement_max_mib" = -1: nine tenths of the free memory, no reserve (bench.py, vp_feeds).  Blocks below 768 MiB are plain hipMalloc (their writes live in the 256 MiB Infinity Cache).  Free with vp_free_device. */
typedef struct {
	float write_gbs;            /* zero-fill rate of this block, GB/s (0 when nothing was probed) */
	float best_gbs, worst_gbs;  /* over the candidates probed for the request */
	int32_t candidates;         /* blocks allocated and probed */
	int32_t selected;           /* 0 = plain hipMalloc; chosen among candidates: 1 = the search met its stop criterion, 2 = it ran out of candidates, bytes or memory first */
	float fast_gbs;             /* the write rate that counts as fast on this device (0.70 x its read-stream rate; 0 when nothing was probed): write_gbs below it = a slow block */
} VpPlacement;
int vp_malloc_device_placed(VpContext* ctx, size_t bytes, int count, void** dptrs, VpPlacement* info /* count entries, or NULL */);
int vp_copy_h2d(VpContext* ctx, int stream, void* dst_d, const void* src_h, size_t bytes);
int vp_copy_d2h(VpContext* ctx, int stream, void* dst_h, const void* src_d, size_t bytes);
int vp_copy_d2h_sync(VpContext* ctx, int stream, void* dst_h, const void* src_d, size_t bytes);
int vp_memset_device(VpContext* ctx, int stream, void* dst_d, int value, size_t bytes);

/* ---- buffer and image objects: what CLArray / RawImage and CLImage are made of ---------------------------------------------------
 * VpBuffer = CLArray (opencl.h:154-165, opencl.cpp:146-147): a device buffer; `init` non-NULL copies host data in at creation
 * (the CL_MEM_COPY_HOST_PTR constructor).  map / unmap are the blocking CLMap of opencl.h:115-152: map(read_back) returns a pinned
 * host shadow holding the device contents (or just waits for the stream when read_back = 0), unmap(write_back) copies it back.
 * VpImage = CLImage from the pool of OpenCL::acquire (opencl.cpp:108-122): an idle image of the same format and size is handed
 * out again, otherwise one is allocated; release returns it to the pool (the reference does that when the last shared_ptr copy
 * goes away); the pool never frees before vp_ctx_destroy.  Rows are tightly packed; VP_FMT_NV12 allocates 2 bytes per pixel like
 * the reference (opencl.cpp:27). */
typedef struct VpBuffer VpBuffer;
typedef struct VpImage VpImage;
typedef enum { VP_FMT_RGBA8 = 0, VP_FMT_U8 = 1, VP_FMT_F32 = 2, VP_FMT_NV12 = 3 } VpPixelFormat;
int vp_buffer_create(VpContext* ctx, size_t bytes, const void* init, VpBuffer** out);
void* vp_buffer_device(const VpBuffer* buffer);
size_t vp_buffer_size(const VpBuffer* buffer);
int vp_buffer_map(VpContext* ctx, int stream, VpBuffer* buffer, int read_back, void** host);
int vp_buffer_unmap(VpContext* ctx, int stream, VpBuffer* buffer, int write_back);
int vp_buffer_destroy(VpContext* ctx, VpBuffer* buffer);
int vp_image_acquire(VpContext* ctx, int format, int w, int h, VpImage** out);
int vp_image_release(VpContext* ctx, VpImage* image);
void* vp_image_device(const VpImage* image);
int vp_image_info(const VpImage* image, int* format, int* w, int* h, size_t* bytes);
int vp_image_map(VpContext* ctx, int stream, VpImage* image, int read_back, void** host);
int vp_image_unmap(VpContext* ctx, int stream, VpImage* image, int write_back);

/* ---- stage kernels ------------------------------------------------------------------------------------------------ */

/* raw2quad (raw2quad.cl:21-39; launched by Resources::raw2quad, Resources.cpp:138-143).
 * raw: batch x (Bayer: 2qh x 2qw bytes | BGR: qh x qw x 3 bytes).  planes: batch x 4 x qh x qw (plane 3 untouched for BGR). */
int vp_raw2quad(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, int batch, uint8_t* planes);

/* The same two stages for four separately allocated plane images, as Resources::raw2quad hands them out
 * (four OpenCL::acquire calls, Resources.cpp:139-140); one frame per call.  plane3 may be NULL for BGR. */
int vp_raw2quad4(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, uint8_t* plane0, uint8_t* plane1, uint8_t* plane2, uint8_t* plane3);
int vp_resample4(VpContext* ctx, int stream, const uint8_t* plane0, const uint8_t* plane1, const uint8_t* plane2, const uint8_t* plane3, int layout, int qw, int qh,
		const VpCameraModel* model, float max_robot_height, float field_scale, float off_x, float off_y, int sampler_mode, int fw, int fh, uint8_t* flat);

/* resampling + field2image (resampling.cl:29-100; Resources.cpp:159).  planes as written by vp_raw2quad.
 * flat: batch x fh x fw RGBA8 (delta-RGB, alpha 255). */
int vp_resample(VpContext* ctx, int stream, const uint8_t* planes, int layout, int qw, int qh, const VpCameraModel* model,
		float max_robot_height, float field_scale, float off_x, float off_y, int sampler_mode, int fw, int fh, int batch, uint8_t* flat);

/* Fused raw2quad+resampling: same result as vp_raw2quad followed by vp_resample, reading the raw buffer directly
 * (the planes are never materialised). */
int vp_resample_raw(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, const VpCameraModel* model,
		float max_robot_height, float field_scale, float off_x, float off_y, int sampler_mode, int fw, int fh, int batch, uint8_t* flat);

/* Coordinate map: field2image() of every flat pixel (resampling.cl:29-47 evaluated at :53).  It depends on the geometry only
 * (Perspective::geometryCheck runs once per geometry version, Perspective.cpp:37-38), not on the frame, so the 18 IEEE
 * divisions per pixel can be paid once.  map: fh x fw pairs (px, py) of binary32 quad-plane coordinates. */
int vp_resample_map(VpContext* ctx, int stream, const VpCameraModel* model, float max_robot_height, float field_scale, float off_x, float off_y,
		int fw, int fh, float* map);

/* vp_resample_raw with the coordinates read from a map built by vp_resample_map for the same geometry: identical output. */
int vp_resample_raw_mapped(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, const float* map, int sampler_mode,
		int fw, int fh, int batch, uint8_t* flat);

/* gradient_dotproduct (gradientDot.cl:22-30; Resources.cpp:160).  graddot: batch x h x w float (exact integers). */
int vp_graddot(VpContext* ctx, int stream, const uint8_t* flat, int w, int h, int offset, int batch, float* graddot);

/* sat_horizontal + sat_vertical (satHorizontal.cl:22-32, satVertical.cl:22-32; Resources.cpp:161-162) on a gradDot image. */
int vp_sat(VpContext* ctx, int stream, const float* graddot, int w, int h, int batch, int variant, int32_t* sat);

/* Fused gradient_dotproduct + SAT: the pass BASELINE.json's HBM-roofline figure is quoted on (8 B/px algorithmic:
 * read flat once, write SAT once).  graddot_out may be NULL; if not it also receives the F32 gradDot debug view. */
int vp_graddot_sat(VpContext* ctx, int stream, const uint8_t* flat, int w, int h, int offset, int batch, int variant,
		int32_t* sat, float* graddot_out);

/* Halo strips: the first and the last `offset` flat pixels of every 64-pixel tile row, one record of 2 * offset pixels per tile column and
 * row: strips[((tile column * hp + row) * 2 + side) * offset + k], side 0 = first / 1 = last pixels, hp = h rounded up to a multiple of 4.
 * The single-pass SAT needs an `offset` wide horizontal halo on both sides of a tile; fetched from `flat` that halo costs one 128-byte line
 * per row and side (measured: 1.51x the algorithmic bytes of the pass), fetched from the neighbours' records it is dense (1.16x).  vp_frame
 * lets the resampling kernel emit the records while it writes flat (the four rows of a workgroup are one contiguous 64-byte store at
 * offset 2); vp_halo_strips builds them from an existing flat.  The layout is private to the library: treat the buffer as opaque.
 * Requires w % 64 == 0 and 1 <= offset <= 4 (vp_halo_strip_bytes returns 0 otherwise and the strips are simply not used).
 * Size per frame: 2 * (w / 64) * hp * offset * 4 bytes. */
size_t vp_halo_strip_bytes(int w, int h, int offset);
int vp_halo_strips(VpContext* ctx, int stream, const uint8_t* flat, int w, int h, int offset, int batch, uint32_t* strips);
int vp_graddot_sat_strips(VpContext* ctx, int stream, const uint8_t* flat, const uint32_t* strips, int w, int h, int offset, int batch, int variant,
		int32_t* sat, float* graddot_out);

/* circle (satBlobCenter.cl:34-42; Resources.cpp:163) on the int32 SAT.  blobcenter: batch x h x w float. */
int vp_blobcenter(VpContext* ctx, int stream, const int32_t* sat, int w, int h, int radius, int batch, float* blobcenter);

/* matches (blobList.cl:36-103; launched by main.cpp:289 with the counter protocol of main.cpp:283-288,299-316).
 * matches: batch x max_matches records; idx: batch x max_matches pixel indices (y*w+x) parallel to matches (may be NULL);
 * counter: batch x 3 ints {appended (may exceed max_matches), score-rejected, non-peak-rejected}; zeroed by this call. */
int vp_bloblist(VpContext* ctx, int stream, const uint8_t* flat, const float* blobcenter, int w, int h, int batch,
		float circ_threshold, float min_score, int radius, int max_matches, int order,
		VpMatch* matches, uint32_t* idx, int32_t* counter);

/* Fused circle + matches straight from the SAT (blobCenter stays in LDS).  blobcenter_out may be NULL. */
int vp_blobcenter_list(VpContext* ctx, int stream, const uint8_t* flat, const int32_t* sat, int w, int h, int batch,
		int circle_radius, float circ_threshold, float min_score, int list_radius, int max_matches, int order,
		VpMatch* matches, uint32_t* idx, int32_t* counter, float* blobcenter_out);

/* gradient_dotproduct + sat_horizontal/vertical + circle + matches in one kernel, no global table (VP_SAT_LOCAL), for the (circle_radius,
 * grad_offset) pairs a geometry can form with circle radius 2..12; any other pair gives the same list through a table private to the library
 * (single-pass SAT into a workspace, then the fused circle + matches): never refused, merely no faster than VP_SAT_TILED. */
int vp_flat_bloblist(VpContext* ctx, int stream, const uint8_t* flat, int w, int h, int batch, int grad_offset, int circle_radius, float circ_threshold, float min_score,
		int list_radius, int max_matches, int order, VpMatch* matches, uint32_t* idx, int32_t* counter);

/* matches (blobScore.cl:23-67; compiled at blob_benchmark.cpp:117, launch commented out :154-155): score image. */
int vp_blobscore(VpContext* ctx, int stream, const uint8_t* flat, const float* blobcenter, int w, int h, int batch,
		float circ_threshold, int radius, float* score);

/* circularize (blobCenter.cl:29-63; never compiled by the reference): non-SAT quadrant ringness, for cross-checks. */
int vp_circularize(VpContext* ctx, int stream, const float* graddot, int w, int h, int batch, int min_radius, int max_radius, float* out);

/* ---- debug / stream colour kernels ------------------------------------------------------------------------
 * NV12 frame layout (what RTPStreamer::sendFrame receives, rtpstreamer.cpp:177-181): w*h luma bytes followed by ceil(h/2) rows of
 * w interleaved U,V bytes = vp_nv12_bytes(w, h) per frame; w must be even.  The reference lets all four pixels of a 2x2 block
 * write the block's UV pair (quad2nv12.cl:54-57, rgba2nv12.cl:28-30: a data race); here the block's last pixel in raster order
 * defines it, which is what a sequential execution of the reference kernel yields. */
size_t vp_nv12_bytes(int w, int h);

/* quad2rgba (quad2rgba.cl:24-53; Resources::quad2rgba, Resources.cpp:145-149): demosaiced RGBA8 at quad resolution, no dRGB.
 * planes: batch x 4 x qh x qw; rgba: batch x qh x qw x 4.  vp_quad2rgba4: four separately acquired plane images, one frame.
 * vp_raw2rgba: the same from the raw sensor buffer with raw2quad fused (planes never materialised). */
int vp_quad2rgba(VpContext* ctx, int stream, const uint8_t* planes, int layout, int qw, int qh, int sampler_mode, int batch, uint8_t* rgba);
int vp_quad2rgba4(VpContext* ctx, int stream, const uint8_t* plane0, const uint8_t* plane1, const uint8_t* plane2, const uint8_t* plane3, int layout, int qw, int qh,
		int sampler_mode, uint8_t* rgba);
int vp_raw2rgba(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, int sampler_mode, int batch, uint8_t* rgba);

/* quad2nv12 (quad2nv12.cl:24-58; Resources::streamQuad, Resources.cpp:166-170): the same colour, BT.601 limited-range NV12. */
int vp_quad2nv12(VpContext* ctx, int stream, const uint8_t* planes, int layout, int qw, int qh, int sampler_mode, int batch, uint8_t* nv12);
int vp_quad2nv124(VpContext* ctx, int stream, const uint8_t* plane0, const uint8_t* plane1, const uint8_t* plane2, const uint8_t* plane3, int layout, int qw, int qh,
		int sampler_mode, uint8_t* nv12);
int vp_raw2nv12(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, int sampler_mode, int batch, uint8_t* nv12);

/* rgb2nv12 (rgba2nv12.cl:22-31) and f2nv12 (f2nv12.cl:22-26): Resources::streamImage for RGBA8 / F32 images (Resources.cpp:172-186). */
int vp_rgba2nv12(VpContext* ctx, int stream, const uint8_t* rgba, int w, int h, int batch, uint8_t* nv12);
int vp_f2nv12(VpContext* ctx, int stream, const float* img, int w, int h, int batch, uint8_t* nv12);

/* ---- whole frame: Resources::raw2quad + Resources::rgba2blobCenter + the blobList launch of main.cpp:283-289 ---------- */
typedef struct {
	int32_t layout, qw, qh;            /* RawImage format/width/height (opencl.h:168-188; Bayer: 2x2-cell counts) */
	VpCameraModel model;               /* Perspective::getCLCameraModel (Perspective.cpp:136-150) */
	float max_robot_height;            /* gcSocket->maxBotHeight (Resources.cpp:159) */
	float field_scale, off_x, off_y;   /* perspective->fieldScale, visibleFieldExtent[0], visibleFieldExtent[2] */
	int32_t fw, fh;                    /* perspective->reprojectedFieldSize */
	int32_t grad_offset;               /* (int)ceilf(maxBlobRadius/fieldScale)/3   (Resources.cpp:160) */
	int32_t circle_radius;             /* (int)ceilf(minBlobRadius/fieldScale)     (Resources.cpp:163) */
	int32_t list_radius;               /* (int)floorf(minBlobRadius/fieldScale)    (main.cpp:289) */
	float circ_threshold;              /* (float)minCircularity                    (main.cpp:289) */
	float min_score;                   /* 0.0f at the reference call site          (main.cpp:289) */
	int32_t max_matches;               /* maxBlobs */
	int32_t sampler_mode, sat_variant, order;
	int32_t use_coord_map;             /* 1: vp_frame keeps a per-stream coordinate map (rebuilt when the geometry scalars change) */
} VpFrameParams;

typedef struct {
	uint8_t* flat;        /* required: batch x fh x fw x 4 */
	int32_t* sat;         /* required: batch x fh x fw (NULL with VP_SAT_LOCAL) */
	VpMatch* matches;     /* required: batch x max_matches */
	uint32_t* idx;        /* optional */
	int32_t* counter;     /* required: batch x 3 */
	uint8_t* planes;      /* optional debug/stream view: batch x 4 x qh x qw (Resources::raw2quad output) */
	float* graddot;       /* optional debug view (main.cpp:291-295, 377-394) */
	float* blobcenter;    /* optional debug view / blob_benchmark input (blob_benchmark.cpp:162) */
} VpFrameBuffers;

/* Every buffer vp_frame writes for `batch` frames of this geometry, allocated once like the reference's image pool and CLArrays (opencl.cpp:108-122,
 * main.cpp:257-258): flat, the table (none with VP_SAT_LOCAL), matches, idx, counter, plus the optional views named in `views`.  The table and flat
 * -- the two image-sized buffers written every frame -- come from vp_malloc_device_placed (the fastest-writing block becomes the table).
 * vp_frame_buffers_placement reports what was drawn; vp_frame_buffers_destroy waits for the context's streams and frees everything. */
typedef struct VpFrameSlab VpFrameSlab;
typedef enum { VP_VIEW_PLANES = 1, VP_VIEW_GRADDOT = 2, VP_VIEW_BLOBCENTER = 4 } VpFrameViews;
int vp_frame_buffers_create(VpContext* ctx, const VpFrameParams* params, int batch, int views, VpFrameBuffers* buffers, VpFrameSlab** slab);
int vp_frame_buffers_placement(const VpFrameSlab* slab, VpPlacement* sat, VpPlacement* flat);
int vp_frame_buffers_destroy(VpContext* ctx, VpFrameSlab* slab);

/* raw -> blob list for `batch` frames on one stream, no host synchronisation inside. */
int vp_frame(VpContext* ctx, int stream, const VpFrameParams* params, const uint8_t* raw, int batch, const VpFrameBuffers* buffers);

/* ---- transport form of a step's blob lists ------------------------------------------------------------------------------------------
 * vp_frame fills fixed-capacity arrays (max_matches records per frame, main.cpp:257: 52 KB per frame at maxBlobs 2000) of which a frame uses a few
 * per cent; what the host loop reads back is counter[3] and min(counter[0], maxBlobs) records (main.cpp:299-301).  vp_records_compact copies exactly
 * that into one slot per frame: [counter 3 x int32 | n int32 | cap x VpMatch | pad | cap x uint32 pixel index], slot size padded to 16 bytes, with
 * n = min(counter[0], max_matches, cap) records and indices present (idx may be NULL).  vp_record_layout_compact returns the size of `batch` slots
 * (a multiple of 256) and the slot size / offsets.  This is what vp_feeds reads back and what its multi-GPU gather carries (include/vp_multi.h). */
size_t vp_record_layout_compact(int batch, int cap, size_t* slot_bytes, size_t* off_matches, size_t* off_idx);
int vp_records_compact(VpContext* ctx, int stream, const int32_t* counter, const VpMatch* matches, const uint32_t* idx, int batch, int max_matches, int cap, uint8_t* out);
/* The same with a completion WORD for host loops that read the records back (main.cpp:297-301 maps and waits): `out` and `done_word` live in pinned host memory
 * (vp_malloc_host; device-visible), the kernel stores the slots there itself and its last workgroup then stores `done_value` into *done_word at system scope.
 * The host waits with vp_wait_word (a spin on its own memory: VP_OK, or VP_ERR_TIMEOUT after timeout_s seconds, 0 = wait for ever) instead of vp_sync: the
 * completion reaches it without a copy command, a marker packet or an interrupt (no faster than copy + vp_sync where the step is bound by the kernels or by the
 * host's own stage: profiles/r06_vp_feeds.jsonl; the form for loops that must not block in the runtime).  Use a value that changes from step to step (a step
 * counter).  An error raised by the stream's kernels is still reported by vp_sync only. */
int vp_records_compact_signal(VpContext* ctx, int stream, const int32_t* counter, const VpMatch* matches, const uint32_t* idx, int batch, int max_matches, int cap, uint8_t* out,
		uint32_t* done_word, uint32_t done_value);
int vp_wait_word(const volatile uint32_t* word, uint32_t value, double timeout_s);

/* ---- synthetic input (libvp_host.so; host-side, no GPU) --------------------------------------------------------------
 * Deterministic test/benchmark frames; not part of the reference. */
int vp_synth_frame(const VpCameraModel* model, int layout, int qw, int qh, float plane_height, float field_len, float field_wid,
		uint32_t seed, int n_robots, int n_balls, float noise_sigma, uint8_t* raw);
void vp_synth_camera(int qw, int qh, float cam_height, float plane_height, float visible_len, float distortion, VpCameraModel* out);
/* ground truth of vp_synth_frame(seed, ...): n x {x, y, radius, r, g, b} discs in field mm (5 per robot, then the balls); returns n */
int vp_synth_truth(uint32_t seed, int n_robots, int n_balls, float field_len, float field_wid, float* out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* VP_HIP_H */
