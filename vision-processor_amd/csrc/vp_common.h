// Internal header of libvp_hip.so (gfx950 only): context, error plumbing, launch bookkeeping, device helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "vp_hip.h"

#define VP_MAX_STREAMS 16
#define VP_WS_SLOTS 6

struct VpEventRec {
	const char* stage;
	hipEvent_t start, stop;
};

struct VpWorkspace {
	void* ptr = nullptr;
	size_t bytes = 0;
};

// device buffer with a lazily allocated pinned shadow for the blocking map / unmap of opencl.h:115-152.  The reference maps from
// its RTP encoder and snapshot threads while the main thread launches (rtpstreamer.cpp:177-181, snapshotwriter.cpp:52-54): map /
// unmap of one object are serialised by its mutex, different objects never share state.
struct VpBuffer {
	void* dev = nullptr;
	void* host = nullptr;
	size_t bytes = 0;
	std::mutex mu;
};
// pooled 2-D image (opencl.cpp:108-122): tightly packed rows
struct VpImage {
	VpBuffer buf;
	int format = 0, w = 0, h = 0;
	bool in_use = false;
};

// Result-neutral tuning knobs (vp_ctx_tune): every value gives bit-identical output.  The library reads no environment variable.
struct VpTuning {
	int sat_workgroups_per_cu = 0; // resident workgroups per CU of the single-pass SAT (2..8; 5 and 6 have their own register budget); 0 = 5, and 4 below 64 frames per launch
	int resample_xcd_bands = -1;   // staged resampling: XCD-contiguous tile bands; -1 = automatic (on for >= 4096 tiles per frame group)
#ifndef VP_RS_STAGE_DEFAULT
#define VP_RS_STAGE_DEFAULT 2
#endif
	int resample_float_stage = VP_RS_STAGE_DEFAULT;  // staged Bayer resampling, BILINEAR_RTZ: box staged as floats of two frames side by side (2, default), floats (1) or bytes (0)
#ifndef VP_RS_GPW_DEFAULT
#define VP_RS_GPW_DEFAULT 0
#endif
	int resample_groups_per_wg = VP_RS_GPW_DEFAULT; // staged Bayer resampling (frame-pair stage): consecutive 16-frame groups one workgroup takes with ONE set-up (coordinate, footprints, box, weights); 0 = automatic (4)
	int resample_bgr_staged = 1;   // BGR8 sensors: staged kernel (1) or the generic mapped kernel (0)
	int candidate_capacity = 0;    // blob list: candidate slots per frame; 0 = max(4 * max_matches, 4096).  Overflow is handled exactly (rescan)
	int sat_halo_strips = 1;       // vp_frame: resampling also writes the dense halo strips the tiled SAT pass reads (0: the pass reads its halo columns from flat)
	int tail_segment_rows = 0;     // circle + detect: image rows per wavefront; 0 = chosen from the launch size (32 / 64 / 128)
	int frame_graph = 0;           // vp_frame replays a captured hipGraph of its launches for a repeated (stream, parameters, batch, buffers) tuple (2: no pre-flight, see vp_frame.hip)
	int tail_emit_kernel = 1;      // walkers: records by a second launch from per-frame survivor queues (1, default) or inside the walker (0)
	int sat_spin_limit = 0;        // single-pass SAT: polls a look-back wait may take before the launch gives up with VP_ERR_TIMEOUT (0 = 2^22, seconds); results do not depend on it
	int sat_debug_drop_tile = -1;  // FAULT INJECTION for the tests (not result-neutral, by design): this tile's column carry is never published, see vp_sat_tiled.hip TiledArgs::drop_tile
	int placement_max_mib = 0;     // vp_malloc_device_placed: MiB the search may hold at once (0 = half of the free device memory, at most a quarter of the device, and it stops while an eighth of the device is still free; -1 = the process owns the GPU: nine tenths of the free memory, a reserve of 1 GiB / 2 % of the device)
	int placement_candidates = 96; // vp_malloc_device_placed / vp_frame_buffers_create: allocations probed at most per request (0 or 1: take what hipMalloc gives); 48 ran
	                               // out twice in six fresh processes on a box whose fast blocks were scarce (profiles/r04_bench_spread.json); the bytes held stay capped
};

// one captured vp_frame launch sequence (vp_frame.hip, vp_ctx_tune "frame_graph")
struct VpFrameGraph {
	std::vector<uint8_t> key; // stream, batch, parameters, raw pointer, buffers, tuning
	hipGraph_t graph = nullptr;
	hipGraphExec_t exec = nullptr;
	int refused = 0;           // captures the runtime could not make; after three the tuple stays on the ordinary path
	int refused_bodies = 0;    // consecutive captures whose launch sequence itself failed (a refused repair, a launch error); after eight likewise
	bool noted = false;        // the fields below hold the stream state the tuple's launches need (recorded after each ordinary call / capture)
	uint64_t ws_generation = 0, sat_shape = 0;
	uint32_t sat_tickets = 0;
	unsigned char map_key[96] = {};
};

struct VpContext {
	int device = 0;
	int n_streams = 0;
	bool own_streams = true;
	hipStream_t streams[VP_MAX_STREAMS] = {};
	VpWorkspace ws[VP_WS_SLOTS][VP_MAX_STREAMS]; // slot 5: private table of VP_SAT_LOCAL calls whose (radius, offset) pair has no walker; slot 0: per-call scratch (candidate lists, append-order records); slot 1: persistent SAT tile carries; slot 2: coordinate map; slot 3: halo strips;
	                                     // slot 4: self-cleaning counters of the blob list (zero between launches: nothing else may ever be placed there)
	unsigned char map_key[VP_MAX_STREAMS][96] = {}; // geometry the cached coordinate map of each stream was built for
	uint32_t sat_epoch[VP_MAX_STREAMS] = {};       // single-pass SAT launches on this stream since its carry state was last reset (the kernels derive the same number from their tickets)
	uint64_t sat_shape[VP_MAX_STREAMS] = {};       // tiling + batch + tickets per launch those launches were made with
	uint32_t sat_tickets[VP_MAX_STREAMS] = {};     // tickets per launch and frame queue of that shape
	uint64_t ws_generation[VP_MAX_STREAMS] = {};   // bumped whenever a workspace of the stream is (re)allocated: captured graphs hold the old pointers
	bool capturing[VP_MAX_STREAMS] = {};           // a vp_frame launch sequence is being captured on the stream: nothing may synchronise, allocate or reset
	std::vector<VpFrameGraph> graphs;
	int64_t graph_stats[4] = {};                   // captures, replays, refused captures, dropped graphs (vp_frame_graph_stats)
	int* host_flag[VP_MAX_STREAMS] = {};           // pinned, device-mapped word the SAT kernel raises on a look-back timeout
	uint32_t* compact_arrived[VP_MAX_STREAMS] = {}; // vp_records_compact_signal: arrival counter of the launch's workgroups (device, self-cleaning)
	std::string device_name;
	int cu_count = 0;
	VpTuning tune;
	bool events_on = false;
	std::mutex mu;
	std::vector<VpEventRec> events;
	std::vector<VpImage*> image_pool; // every image ever acquired; idle ones are handed out again (the pool never frees)
};

void vp_set_error(const char* fmt, ...);
void vp_clear_error(); // a call that succeeded after an internal retry leaves no stale message behind

#define VP_CHECK_HIP(expr)                                                                                     \
	do {                                                                                                         \
		hipError_t e_ = (expr);                                                                                  \
		if(e_ != hipSuccess) {                                                                                   \
			vp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);           \
			return VP_ERR_HIP;                                                                                   \
		}                                                                                                        \
	} while(0)

#define VP_REQUIRE(cond, ...)                                                                                  \
	do {                                                                                                         \
		if(!(cond)) {                                                                                            \
			vp_set_error(__VA_ARGS__);                                                                           \
			return VP_ERR_INVALID;                                                                               \
		}                                                                                                        \
	} while(0)

#define VP_TRY(expr)                       \
	do {                                     \
		int rc_ = (expr);                    \
		if(rc_ != VP_OK) return rc_;         \
	} while(0)

static inline int vp_check_ctx(VpContext* ctx, int stream) {
	if(!ctx) {
		vp_set_error("null context");
		return VP_ERR_INVALID;
	}
	if(stream < 0 || stream >= ctx->n_streams) {
		vp_set_error("stream %d out of range [0,%d)", stream, ctx->n_streams);
		return VP_ERR_INVALID;
	}
	return VP_OK;
}

// Scratch of at least `bytes` on `stream` in `slot`; grows (stream-synchronising, zero-filled) only when too small.
// *grew (optional) tells the caller that persistent contents were lost.
int vp_workspace(VpContext* ctx, int stream, int slot, size_t bytes, void** out, bool* grew = nullptr);

void vp_frame_graphs_clear(VpContext* ctx); // vp_frame.hip
// vp_alloc.hip
int vp_probe_stream_impl(VpContext* ctx, int stream, void* ptr, size_t bytes, int write, int reps, float* us);

#ifndef VP_STAGE_EVENT_FLAGS
#define VP_STAGE_EVENT_FLAGS hipEventDisableSystemFence
#endif
// RAII bracket for per-stage hipEvent timing (the cl::Event list of opencl.h:84)
struct VpStageTimer {
	VpContext* ctx;
	hipStream_t s;
	const char* stage;
	hipEvent_t start = nullptr, stop = nullptr;
	VpStageTimer(VpContext* c, int stream, const char* name): ctx(c), s(c->streams[stream]), stage(name) {
		// (timing-only events: no system-scope fence when they fire -- the default writes the caches back and invalidates them at every stage boundary, which
		// the NEXT kernel pays for)
		if(ctx->events_on && hipEventCreateWithFlags(&start, VP_STAGE_EVENT_FLAGS) == hipSuccess && hipEventCreateWithFlags(&stop, VP_STAGE_EVENT_FLAGS) == hipSuccess) (void)hipEventRecord(start, s);
	}
	~VpStageTimer() {
		if(start && stop) {
			(void)hipEventRecord(stop, s);
			std::lock_guard<std::mutex> g(ctx->mu);
			ctx->events.push_back({stage, start, stop});
		}
	}
};

static inline int vp_launch_ok(const char* what) {
	hipError_t e = hipGetLastError();
	if(e != hipSuccess) {
		vp_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
		return VP_ERR_HIP;
	}
	return VP_OK;
}

static inline unsigned vp_div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

// Halo strips (include/vp_hip.h): per frame the first and the last `offset` RGBA8 pixels of every 64-pixel tile row, one record
// [FIRST: flat[y][64 j + k] | LAST: flat[y][64 j + 64 - offset + k]], k < offset, per tile column j and row y.  The staged resampling
// kernel stores the four records of a workgroup (4 rows) as ONE contiguous piece; the other producers write per pixel (emit_strip).  A SAT
// tile takes its left halo from LAST of its left neighbour and its right halo from FIRST of its right neighbour: half of every record it
// touches (the earlier layout -- two dense arrays, every fetched line used whole -- cost the producer two 8-byte pieces per wave and
// frame: 0.085 ms of resampling per 128 frames against 0.04 ms now, for 0.045 GB more SAT reads).
struct VpStrips {
	uint32_t* base;       // nullptr: not in use
	int tiles_x, h, hp, o; // hp = h rounded up to a multiple of 4 (rows of a resampling workgroup)
	size_t frame_words;   // uint32 per frame = tiles_x * hp * 2 * o
	// One record of 2 o pixels per tile j and row y: [FIRST o pixels | LAST o pixels] of that 64-pixel tile row.  The four rows a resampling
	// workgroup produces are contiguous (8 o dwords: one aligned 64-byte store for o = 2).
	__host__ __device__ size_t rec(int j, int y, int e) const { return ((size_t)j * hp + y) * 2 * o + e; } // e < o: FIRST, else LAST
	__host__ __device__ size_t left(int b, int y, int k) const { return rec(b - 1, y, o + k); } // flat[y][64 b - o + k], b >= 1
	__host__ __device__ size_t right(int b, int y, int k) const { return rec(b, y, k); }        // flat[y][64 b + k]
};
static inline bool vp_strips_supported(int w, int offset) { return w > 0 && (w % 64) == 0 && offset >= 1 && offset <= 4; }
static inline VpStrips vp_make_strips(uint32_t* base, int w, int h, int offset) {
	VpStrips s;
	s.base = base;
	s.tiles_x = w / 64;
	s.h = h;
	s.hp = (h + 3) & ~3;
	s.o = offset;
	s.frame_words = (size_t)2 * s.tiles_x * s.hp * offset;
	return s;
}

// ---- device helpers ----------------------------------------------------------------------------------------------
__device__ __forceinline__ int vp_clampi(int v, int lo, int hi) { return min(max(v, lo), hi); } // one v_med3_i32

// XCD-aware block remap (cdna_hip_programming.md T1, bijective form): blocks are dealt round-robin over the 8 XCDs, each with
// a private L2; this hands every XCD a CONTIGUOUS range of logical block ids so that neighbouring tiles share an L2 (halo
// re-reads become L2 hits).  Speed only: any placement gives the same results.
__device__ __forceinline__ unsigned vp_xcd_remap(unsigned b, unsigned n) {
	const unsigned q = n >> 3, r = n & 7u, xcd = b & 7u;
	return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// inclusive wave-64 prefix sum (wrapping int32)
__device__ __forceinline__ int vp_wave_scan_incl(int v) {
	const int lane = threadIdx.x & 63;
#pragma unroll
	for(int d = 1; d < 64; d <<= 1) {
		const int t = __shfl_up(v, d, 64);
		if(lane >= d) v += t;
	}
	return v;
}

typedef short short2v __attribute__((ext_vector_type(2)));

// inclusive wave-64 prefix sum in 6 DPP adds (row_shr 1,2,4,8 inside each 16-lane row, then row_bcast 15 / 31)
__device__ __forceinline__ int scan_incl_dpp(int v) {
	v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
	return v;
}

// sum over channels of (xp - xn) * (yp - yn) on RGBA8 taps (gradientDot.cl:25-29).  Alpha is 255 on every tap: its difference
// is 0 and is left out.  The byte differences are formed by SDWA subtractions that read one byte of each operand and write
// one 16-bit half of the result (v_sub_u16_sdwa ... src_sel:BYTE_n dst_sel:WORD_m): (R, B) packed in one register, G in
// another, then two 2-way 16-bit dot products.  8 VALU instructions per pixel; the mask / shift / v_pk_sub_i16 form took 14
// and this phase is what bounds the kernel (measured: ~60 % of the pass is VALU issue time).
__device__ __forceinline__ int grad_dot_pk(uint32_t xp, uint32_t xn, uint32_t yp, uint32_t yn) {
	uint32_t gx_rb, gx_g, gy_rb, gy_g;
	asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(gx_rb) : "v"(xp), "v"(xn));
	asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_2" : "+v"(gx_rb) : "v"(xp), "v"(xn));
	asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(gx_g) : "v"(xp), "v"(xn));
	asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(gy_rb) : "v"(yp), "v"(yn));
	asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_2" : "+v"(gy_rb) : "v"(yp), "v"(yn));
	asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(gy_g) : "v"(yp), "v"(yn));
	int acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, gx_rb), __builtin_bit_cast(short2v, gy_rb), 0, false);
	acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, gx_g), __builtin_bit_cast(short2v, gy_g), acc, false);
	return acc;
}

// The same value from four unsigned byte dot products (v_dot4_u32_u8): sum_c (xp_c - xn_c)(yp_c - yn_c) = xp.yp + xn.yn - xp.yn - xn.yp over the three
// colour bytes.  The ALPHA BYTE OF ALL FOUR ARGUMENTS MUST BE ZERO (the caller masks each pixel once when it loads it).  5 VALU instructions
// instead of 9; every partial sum is < 2^19, the result is the exact integer.
__device__ __forceinline__ int grad_dot_u8(uint32_t xp, uint32_t xn, uint32_t yp, uint32_t yn) {
	uint32_t pos = __builtin_amdgcn_udot4(xp, yp, 0u, false);
	pos = __builtin_amdgcn_udot4(xn, yn, pos, false);
	uint32_t neg = __builtin_amdgcn_udot4(xp, yn, 0u, false);
	neg = __builtin_amdgcn_udot4(xn, yp, neg, false);
	return (int)(pos - neg);
}

__device__ __forceinline__ int vp_wave_sum(int v) {
#pragma unroll
	for(int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}
__device__ __forceinline__ unsigned vp_wave_sum_u(unsigned v) {
#pragma unroll
	for(int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}
