// Tail of the chain: ringness score from the SAT, thresholded local maxima, blob records.
//   circle      (satBlobCenter.cl:34-42, Resources.cpp:163)
//   matches     (blobList.cl:36-103, launched at main.cpp:289; counter protocol main.cpp:283-288,299-316)
//   matches     (blobScore.cl:23-67, unlaunched in the reference)      -> vp_blobscore
//   circularize (blobCenter.cl:29-63, never compiled by the reference)  -> vp_circularize
//
// The reference's `matches` does detection and the O(R^2) disc statistics in one divergent work-item.  Here the two are
// split: a full-frame detect pass appends candidate pixel indices (wave-aggregated atomics), then an emit pass gives
// each candidate a whole wavefront (64 lanes cover the disc, integer wave reductions => order-independent, exact).
// Floating-point expressions keep the reference's evaluation order; sqrt and division are the correctly rounded
// binary32 forms (HIP default), built with -ffp-contract=off.
#include "vp_common.h"

#include <cmath>

// ---------------------------------------------------------------- circle -------------------------------------------
__device__ __forceinline__ uint32_t sat_tap(const int32_t* __restrict__ s, int w, int h, int x, int y) {
	return (uint32_t)s[(size_t)vp_clampi(y, 0, h - 1) * w + vp_clampi(x, 0, w - 1)];
}

// satBlobCenter.cl:37-41 with wrapping integer box sums (exact), converted once to binary32
__device__ __forceinline__ float circle_at(const int32_t* __restrict__ s, int w, int h, int x, int y, int r, float div) {
	const float pp = (float)(int32_t)(sat_tap(s, w, h, x + r, y + r) - sat_tap(s, w, h, x + r, y + 1) - sat_tap(s, w, h, x + 1, y + r) + sat_tap(s, w, h, x + 1, y + 1));
	const float pn = (float)(int32_t)(sat_tap(s, w, h, x + r, y - r) - sat_tap(s, w, h, x + r, y - 1) - sat_tap(s, w, h, x + 1, y - r) + sat_tap(s, w, h, x + 1, y - 1));
	const float np = (float)(int32_t)(sat_tap(s, w, h, x - r, y + r) - sat_tap(s, w, h, x - r, y + 1) - sat_tap(s, w, h, x - 1, y + r) + sat_tap(s, w, h, x - 1, y + 1));
	const float nn = (float)(int32_t)(sat_tap(s, w, h, x - r, y - r) - sat_tap(s, w, h, x - r, y - 1) - sat_tap(s, w, h, x - 1, y - r) + sat_tap(s, w, h, x - 1, y - 1));
	const float a = nn < pp ? nn : pp; // OpenCL min(x, y) = y < x ? y : x
	const float b = np < pn ? np : pn;
	return (b < a ? b : a) / div;
}

__global__ __launch_bounds__(256) void k_circle(const int32_t* __restrict__ sat, float* __restrict__ out, int w, int h, int r) {
	const int x = blockIdx.x * 64 + threadIdx.x;
	const int y = blockIdx.y * 4 + threadIdx.y;
	if(x >= w || y >= h) return;
	const size_t frame = (size_t)blockIdx.z * w * h;
	out[frame + (size_t)y * w + x] = circle_at(sat + frame, w, h, x, y, r, (float)(r * r));
}

static int launch_circle(VpContext* ctx, int stream, const int32_t* sat, int w, int h, int radius, int batch, float* out) {
	k_circle<<<dim3(vp_div_up(w, 64), vp_div_up(h, 4), batch), dim3(64, 4), 0, ctx->streams[stream]>>>(sat, out, w, h, radius);
	return vp_launch_ok("circle");
}

extern "C" int vp_blobcenter(VpContext* ctx, int stream, const int32_t* sat, int w, int h, int radius, int batch, float* blobcenter) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(sat && blobcenter && w > 0 && h > 0 && batch > 0 && batch <= 65535 && radius >= 0, "vp_blobcenter: bad arguments");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "satBlobCenter");
	return launch_circle(ctx, stream, sat, w, h, radius, batch, blobcenter);
}

// ---------------------------------------------------------------- blobList -----------------------------------------
// Workspace of one list call.  The two counter arrays live in a workspace slot of their own (slot 4) and are SELF-CLEANING: the
// last workgroup of a frame puts cand_count / done back to zero, so no launch has to clear them (a workspace is zero-filled when
// it is allocated or grows, and nothing else is ever placed in that slot).  The lists are per-call scratch (slot 0).
struct ListScratch {
	int32_t* cand_count;  // batch   candidates found by the detect pass (may exceed cap: the emit pass then rescans the frame)
	int32_t* done;        // batch x ARRIVE_WORDS   arrival counters of the last-workgroup-done pattern (frame_arrive_and_sort)
	uint32_t* cand;       // batch x cap   candidate pixel indices
	uint32_t* idx;        // batch x max  (append-order idx: RASTER order, or when the caller passes none)
	uint8_t* rec_tmp;     // batch x max x 22   append-order records (RASTER order only)
	int cap;              // candidate capacity per frame
};

static size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// Arrival counters of one frame: up to ARRIVE_SHARDS shard counters + one top counter, each on a 64-byte line of its own.  A single
// counter per frame serialises the arrivals (one returning atomic per workgroup on ONE address: ~88 per microsecond, measured as
// +11 us on a single frame with 960 workgroups); with shards a workgroup adds to its shard, and only the last arriver of a shard
// adds to the top counter.
constexpr int ARRIVE_SHARDS = 64, ARRIVE_LINE = 16, ARRIVE_WORDS = (ARRIVE_SHARDS + 1) * ARRIVE_LINE;

// The candidate list is bounded: cap = max(4 * max_matches, 4096) entries per frame (vp_ctx_tune "candidate_capacity" overrides).
// A frame with more thresholded local maxima than that is still processed exactly: the emit pass sees cand_count > cap and
// rescans the frame itself instead of reading the list (slow, pathological inputs only).
static int list_scratch(VpContext* ctx, int stream, int batch, int max_matches, ListScratch* s) {
	const int cap = ctx->tune.candidate_capacity > 0 ? ctx->tune.candidate_capacity : (max_matches > 1024 ? 4 * max_matches : 4096);
	const size_t b_cnt = align256(sizeof(int32_t) * (1 + ARRIVE_WORDS) * (size_t)batch), b_cand = align256(sizeof(uint32_t) * (size_t)cap * batch);
	const size_t b_idx = align256(sizeof(uint32_t) * (size_t)max_matches * batch), b_rec = align256((size_t)22 * max_matches * batch);
	void *base = nullptr, *cnt = nullptr;
	VP_TRY(vp_workspace(ctx, stream, 4, b_cnt, &cnt));
	VP_TRY(vp_workspace(ctx, stream, 0, b_cand + b_idx + b_rec, &base));
	s->done = static_cast<int32_t*>(cnt); // [batch][ARRIVE_WORDS], then cand_count[batch]: the offset of a frame's words does not depend on the batch
	s->cand_count = s->done + (size_t)ARRIVE_WORDS * batch;
	uint8_t* p = static_cast<uint8_t*>(base);
	s->cand = reinterpret_cast<uint32_t*>(p); p += b_cand;
	s->idx = reinterpret_cast<uint32_t*>(p); p += b_idx;
	s->rec_tmp = p;
	s->cap = cap;
	return VP_OK;
}

__device__ __forceinline__ float bc_tap(const float* __restrict__ c, int w, int h, int x, int y) {
	return c[(size_t)vp_clampi(y, 0, h - 1) * w + vp_clampi(x, 0, w - 1)];
}

// Wave-aggregated stream compaction (the blob list's atomic_inc of blobList.cl:87 happens later, once per surviving candidate): the
// lanes holding a candidate are counted with a 64-bit ballot, lane 0 reserves that many slots with ONE atomic, and every lane takes
// base + (set bits below it) (v_mbcnt_lo / _hi).  Must be called with all 64 lanes active.  Slots beyond `cap` are counted but not
// written: the emit pass sees count > cap and rescans the frame instead.
__device__ __forceinline__ void wave_append(bool is_cand, uint32_t pix, uint32_t* __restrict__ cand_frame, int32_t* __restrict__ count, int cap, int lane) {
	const unsigned long long m = __ballot(is_cand);
	if(m == 0) return; // wave-uniform
	int base = 0;
	if(lane == 0) base = atomicAdd(count, __popcll(m));
	base = __builtin_amdgcn_readfirstlane(base);
	const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
	if(is_cand && slot < cap) cand_frame[slot] = pix;
}

// blobList.cl:37-55: threshold, 4-neighbour strict-'>' peak test, on a materialised blobCenter image
__global__ __launch_bounds__(256) void k_detect(const float* __restrict__ circ, uint32_t* __restrict__ cand, int32_t* __restrict__ cand_count,
		int32_t* __restrict__ counter, int w, int h, float thr, int cap) {
	const int x = blockIdx.x * 64 + threadIdx.x;
	const int y = blockIdx.y * 4 + threadIdx.y;
	const int z = blockIdx.z;
	const size_t n = (size_t)w * h;
	const float* c = circ + z * n;
	const bool in = x < w && y < h;
	const float v = in ? c[(size_t)y * w + x] : 0.f;
	const bool above = in && !(v < thr);
	bool peak = false;
	if(above) peak = !(bc_tap(c, w, h, x - 1, y) > v || bc_tap(c, w, h, x + 1, y) > v || bc_tap(c, w, h, x, y - 1) > v || bc_tap(c, w, h, x, y + 1) > v);
	const unsigned long long mr = __ballot(above && !peak);
	if(mr != 0 && threadIdx.x == 0) atomicAdd(&counter[3 * z + 2], __popcll(mr));
	wave_append(above && peak, (uint32_t)(y * w + x), cand + (size_t)z * cap, cand_count + z, cap, threadIdx.x);
}

// 22-byte packed record (blobList.cl:26-32): only 2-byte alignment is guaranteed.  The pieces are WRITE-THROUGH stores (agent-scope
// relaxed, global_store_short sc1): the workgroup that sorts a frame's records reads what other CUs wrote in the same launch, and
// write-through stores need no release fence (cdna_hip_programming.md Guideline 16, R1) -- a __threadfence() per emitting workgroup
// (buffer_wbl2: an L2-wide write-back each) made this kernel 20 x slower than the work it does.
typedef __attribute__((address_space(1))) uint16_t gu16_t;
typedef __attribute__((address_space(1))) uint32_t gu32_t;
__device__ __forceinline__ void st_wt16(uint16_t* p, uint32_t v) { __hip_atomic_store((gu16_t*)p, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the matching loads (global_load_ushort / dword sc1): they bypass this CU's L1 and are coherent with the write-through stores of
// other CUs on any XCD -- a plain load could be served a stale copy of the line (the workspace is reused by every launch)
__device__ __forceinline__ uint16_t ld_wt16(const uint16_t* p) { return __hip_atomic_load((gu16_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_wt32(const uint32_t* p) { return __hip_atomic_load((gu32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_match(uint8_t* dst, float x, float y, const uint32_t color[3], uint32_t center, float circ, float score) {
	uint16_t* d = reinterpret_cast<uint16_t*>(dst);
	const uint32_t xb = __float_as_uint(x), yb = __float_as_uint(y), cb = __float_as_uint(circ), sb = __float_as_uint(score);
	st_wt16(d + 0, xb); st_wt16(d + 1, xb >> 16);
	st_wt16(d + 2, yb); st_wt16(d + 3, yb >> 16);
	st_wt16(d + 4, color[0] | (color[1] << 8));
	st_wt16(d + 5, color[2] | ((center & 0xFF) << 8));
	st_wt16(d + 6, ((center >> 8) & 0xFF) | (((center >> 16) & 0xFF) << 8));
	st_wt16(d + 7, cb); st_wt16(d + 8, cb >> 16);
	st_wt16(d + 9, sb); st_wt16(d + 10, sb >> 16);
}

// disc statistics of blobList.cl:57-78 by one wavefront; all lanes return the same values
__device__ __forceinline__ float disc_score_wave(const uint32_t* __restrict__ f, int w, int h, int px, int py, int radius, float circ, uint32_t color[3]) {
	const int lane = threadIdx.x & 63;
	const int D = 2 * radius + 1, sq = radius * radius;
	uint32_t s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0}, n = 0;
	for(int k = lane; k < D * D; k += 64) {
		const int dy = k / D - radius, dx = k % D - radius;
		if(dx * dx + dy * dy <= sq) {
			const uint32_t v = f[(size_t)vp_clampi(py + dy, 0, h - 1) * w + vp_clampi(px + dx, 0, w - 1)];
#pragma unroll
			for(int c = 0; c < 3; c++) {
				const uint32_t b = (v >> (8 * c)) & 0xFF;
				s1[c] += b;
				s2[c] += b * b;
			}
			n++;
		}
	}
	n = vp_wave_sum_u(n);
	const float nf = (float)(int)n;
	float sd[3];
#pragma unroll
	for(int c = 0; c < 3; c++) {
		s1[c] = vp_wave_sum_u(s1[c]);
		s2[c] = vp_wave_sum_u(s2[c]);
		const float a = (float)s1[c];
		const float q = a * a / nf;
		sd[c] = sqrtf(((float)s2[c] - q) / nf); // :76 (native_sqrt -> IEEE sqrt)
		color[c] = s1[c] / n;                    // :96 uint division
	}
	return circ / (sd[0] + sd[1] + sd[2]);       // :78
}

// Last-workgroup-done (cdna_hip_programming.md Guideline 16, R1): the records of a frame were stored write-through by whichever
// workgroups found them; every storing wave drains its stores, the workgroup meets at a barrier, then ONE lane counts the arrival
// with an agent-scope atomic.  The workgroup whose add came last (told by the value the add returned)
// reads the records with sc1 loads, sorts them by pixel index into the caller's arrays when the order is RASTER, and puts the
// self-cleaning counters back to zero: no separate sort launch, no clearing launch.  Called by all 256 threads of a workgroup.
__device__ __forceinline__ void frame_arrive_and_sort(int32_t* done /* this frame's ARRIVE_WORDS */, int wg /* this workgroup's number within the frame */, int n_workgroups,
		int32_t* cand_count, int32_t* counter, const uint8_t* rec_work, const uint32_t* idx_work,
		int max_matches, int order, uint8_t* rec_sorted, uint32_t* idx_sorted, uint32_t* tile /* LDS [256] */, int* s_last /* LDS */) {
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if(threadIdx.x == 0) {
		const int per = (n_workgroups + ARRIVE_SHARDS - 1) / ARRIVE_SHARDS, shard = wg / per, n_shards = (n_workgroups + per - 1) / per;
		const int in_shard = min(per, n_workgroups - shard * per);
		int last = 0;
		if(atomicAdd(done + shard * ARRIVE_LINE, 1) == in_shard - 1) {
			__hip_atomic_store(done + shard * ARRIVE_LINE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // self-cleaning
			last = atomicAdd(done + ARRIVE_SHARDS * ARRIVE_LINE, 1) == n_shards - 1;
		}
		*s_last = last;
	}
	__syncthreads();
	if(!*s_last) return;
	// No acquire fence: every byte read below (records, indices, the counter) was stored write-through / by an L2 atomic and drained before its
	// workgroup's arrival was counted, and is read with sc1 loads (Guideline 16, form R1: that replaces the release AND the acquire).
	if(threadIdx.x == 0) { // self-cleaning: every workgroup of this frame has arrived
		if(cand_count) *cand_count = 0;
		done[ARRIVE_SHARDS * ARRIVE_LINE] = 0;
	}
	if(order != VP_ORDER_RASTER) return;
	// RASTER order: rank of each record = number of records with a smaller pixel index (indices are unique).  This is a serial tail of the
	// launch (a single frame spent 6 us here: fence, counter, indices, records -- four dependent round trips), so slot t of the append-order
	// arrays is requested together with the counter, before it is known whether the slot is in use (it is allocated either way).
	const int t = (int)threadIdx.x;
	const bool slot = t < max_matches;
	const uint16_t* sp0 = reinterpret_cast<const uint16_t*>(rec_work + (size_t)(slot ? t : 0) * 22);
	const uint32_t mine0 = ld_wt32(idx_work + (slot ? t : 0));
	uint16_t r0[11];
#pragma unroll
	for(int k = 0; k < 11; k++) r0[k] = ld_wt16(sp0 + k);
	const int nrec = min(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), max_matches);
	if(nrec <= 256) { // one pass (the usual case: ~130 blobs per frame)
		tile[t] = t < nrec ? mine0 : 0xFFFFFFFFu;
		__syncthreads();
		if(t < nrec) {
			int rank = 0;
			for(int j2 = 0; j2 < nrec; j2++) rank += tile[j2] < mine0;
			uint16_t* dp = reinterpret_cast<uint16_t*>(rec_sorted + (size_t)rank * 22);
#pragma unroll
			for(int k = 0; k < 11; k++) dp[k] = r0[k];
			if(idx_sorted) idx_sorted[rank] = mine0;
		}
		return;
	}
	for(int t0 = 0; t0 < nrec; t0 += 256) {
		const int tt = t0 + t;
		const uint32_t mine = tt < nrec ? ld_wt32(idx_work + tt) : 0;
		int rank = 0;
		for(int j0 = 0; j0 < nrec; j0 += 256) {
			const int j = j0 + t;
			tile[t] = j < nrec ? ld_wt32(idx_work + j) : 0xFFFFFFFFu;
			__syncthreads();
			const int m = min(256, nrec - j0);
			for(int j2 = 0; j2 < m; j2++) rank += tile[j2] < mine;
			__syncthreads();
		}
		if(tt < nrec) {
			const uint16_t* sp = reinterpret_cast<const uint16_t*>(rec_work + (size_t)tt * 22);
			uint16_t* dp = reinterpret_cast<uint16_t*>(rec_sorted + (size_t)rank * 22);
#pragma unroll
			for(int k = 0; k < 11; k++) dp[k] = ld_wt16(sp + k);
			if(idx_sorted) idx_sorted[rank] = mine;
		}
	}
}

// One surviving candidate -> one record (blobList.cl:57-102), by a whole wavefront: disc statistics, score, the atomic append.
// v = blobCenter at the peak, nx / qx / ny / qy its left / right / upper / lower neighbours (clamp-to-edge values).
__device__ __forceinline__ void emit_record(const uint32_t* __restrict__ f, int w, int h, int px, int py, float v, float nx, float qx, float ny, float qy,
		float min_score, int radius, int max_matches, int32_t* counter /* frame's [3] */, uint8_t* rec_work, uint32_t* idx_work, int lane) {
	uint32_t color[3];
	const uint32_t pix = (uint32_t)(py * w + px);
	const uint32_t center = f[pix]; // requested together with the disc's pixels, not behind the atomic below
	const float score = disc_score_wave(f, w, h, px, py, radius, v, color);
	if(lane == 0) {
		if(score < min_score) { // :79-82 (NaN is kept)
			atomicAdd(&counter[1], 1);
		} else {
			const int i = atomicAdd(&counter[0], 1); // :87
			if(i < max_matches) {                   // :88-89
				const float mx = (float)px + 0.5f * (nx - qx) / (nx - 2 * v + qx); // :93
				const float my = (float)py + 0.5f * (ny - qy) / (ny - 2 * v + qy); // :94
				store_match(rec_work + (size_t)i * 22, mx, my, color, center, v, score);
				__hip_atomic_store((gu32_t*)(idx_work + i), pix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
}

// blobList.cl:57-102, one wavefront per candidate.  FROM_SAT: blobCenter was never materialised; the five values the record
// needs (centre + 4 neighbours, clamp-to-edge) are recomputed from the SAT by lanes 0-4.
// The LAST workgroup of a frame to finish sorts the frame's records (frame_arrive_and_sort).
template<bool FROM_SAT>
__global__ __launch_bounds__(256) void k_emit(const uint32_t* __restrict__ flat, const void* __restrict__ src, const uint32_t* __restrict__ cand,
		int32_t* __restrict__ cand_count, int32_t* __restrict__ done, int32_t* __restrict__ counter, uint8_t* __restrict__ rec_work, uint32_t* __restrict__ idx_work,
		int w, int h, float thr, float min_score, int radius, int max_matches, int circle_radius, int cap, int order,
		uint8_t* __restrict__ rec_sorted, uint32_t* __restrict__ idx_sorted) {
	__shared__ uint32_t tile[256];
	__shared__ int s_last;
	const int z = blockIdx.y;
	const size_t n = (size_t)w * h;
	const uint32_t* f = flat + z * n;
	const float* c = reinterpret_cast<const float*>(src) + z * n;
	const int32_t* sat = reinterpret_cast<const int32_t*>(src) + z * n;
	const int lane = threadIdx.x & 63;
	const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
	const int total = cand_count[z];
	const float cdiv = (float)(circle_radius * circle_radius);

	auto emit_one = [&](uint32_t pix) { // all 64 lanes
		const int px = (int)(pix % (uint32_t)w), py = (int)(pix / (uint32_t)w);
		float v, nx, qx, ny, qy;
		if(FROM_SAT) {
			const int dx = lane == 1 ? -1 : (lane == 2 ? 1 : 0), dy = lane == 3 ? -1 : (lane == 4 ? 1 : 0);
			const float mine = lane < 5 ? circle_at(sat, w, h, vp_clampi(px + dx, 0, w - 1), vp_clampi(py + dy, 0, h - 1), circle_radius, cdiv) : 0.f;
			v = __shfl(mine, 0, 64); nx = __shfl(mine, 1, 64); qx = __shfl(mine, 2, 64); ny = __shfl(mine, 3, 64); qy = __shfl(mine, 4, 64);
		} else {
			v = c[pix];
			nx = bc_tap(c, w, h, px - 1, py); qx = bc_tap(c, w, h, px + 1, py);
			ny = bc_tap(c, w, h, px, py - 1); qy = bc_tap(c, w, h, px, py + 1);
		}
		emit_record(f, w, h, px, py, v, nx, qx, ny, qy, min_score, radius, max_matches, counter + 3 * z, rec_work + (size_t)z * max_matches * 22, idx_work + (size_t)z * max_matches, lane);
	};

	if(total <= cap) {
		for(int k = gw; k < total; k += nw) emit_one(cand[(size_t)z * cap + k]);
	} else {
		// More candidates than list slots (pathological frame): the list is incomplete, so find the candidates again -- 64 pixels per
		// wavefront step, the same threshold + peak test on the same values, then one emit per set ballot bit.  Exact, just slow.
		for(size_t p0 = (size_t)gw * 64; p0 < n; p0 += (size_t)nw * 64) {
			const size_t pix = p0 + lane;
			const bool in = pix < n;
			const int px = in ? (int)(pix % (uint32_t)w) : 0, py = in ? (int)(pix / (uint32_t)w) : 0;
			bool is = false;
			if(in) {
				if(FROM_SAT) {
					const float v = circle_at(sat, w, h, px, py, circle_radius, cdiv);
					if(!(v < thr))
						is = !(circle_at(sat, w, h, max(px - 1, 0), py, circle_radius, cdiv) > v || circle_at(sat, w, h, min(px + 1, w - 1), py, circle_radius, cdiv) > v ||
						       circle_at(sat, w, h, px, max(py - 1, 0), circle_radius, cdiv) > v || circle_at(sat, w, h, px, min(py + 1, h - 1), circle_radius, cdiv) > v);
				} else {
					const float v = c[pix];
					if(!(v < thr)) is = !(bc_tap(c, w, h, px - 1, py) > v || bc_tap(c, w, h, px + 1, py) > v || bc_tap(c, w, h, px, py - 1) > v || bc_tap(c, w, h, px, py + 1) > v);
				}
			}
			unsigned long long m = __ballot(is);
			while(m != 0) { // wave-uniform
				const int bpos = __ffsll((long long)m) - 1;
				m &= m - 1;
				emit_one((uint32_t)(p0 + (size_t)bpos));
			}
		}
	}

	frame_arrive_and_sort(done + (size_t)z * ARRIVE_WORDS, (int)blockIdx.x, (int)gridDim.x, cand_count + z, counter + 3 * z, rec_work + (size_t)z * max_matches * 22, idx_work + (size_t)z * max_matches, max_matches, order,
			rec_sorted + (size_t)z * max_matches * 22, idx_sorted ? idx_sorted + (size_t)z * max_matches : nullptr, tile, &s_last);
}

__global__ __launch_bounds__(256) void k_zero(int32_t* __restrict__ a, int na) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if(i < na) a[i] = 0;
}

// blobCenter = min(quadrant sums) / r^2 (satBlobCenter.cl:41) from the integer minimum.  For r = 3..10: m / (r*r), correctly rounded, without
// the division sequence: q0 = m*rcp, e = fma(-d, q0, m), q = fma(e, rcp, q0) -- exhaustively verified equal to m / d for every float m =
// (float)n, |n| <= 2^27, d = 9 .. 100 (tools/divcheck.hip); box sums are < 2^25.
template<int R>
__device__ __forceinline__ float walk_quot(int m) {
	const float div = (float)(R * R), f = (float)m;
	if(R >= 3 && R <= 10) {
		const float rcp = 1.0f / div, q0 = f * rcp;
		return __builtin_fmaf(__builtin_fmaf(-div, q0, f), rcp, q0);
	}
	return f / div;
}

// Threshold + peak test + record emission of one image row for the walkers.  The walkers keep the INTEGER quadrant minimum of the last three
// rows; int -> float conversion and the division by r^2 are both monotonic, so `blobCenter >= threshold` (blobList.cl:39) is the exact integer
// test m >= m_thr with m_thr = the smallest integer whose quotient is not below the threshold (found by the host by bisection with the same
// binary32 division).  Only when some lane of the wavefront passes it (a fraction of a per cent of the rows) are the five blobCenter values
// formed, the neighbours fetched (left / right by DPP wave shifts; outside the image the pixel's own value: `own > own` is false) and the strict
// '>' peak test of blobList.cl:47-52 made.  All 64 lanes must be active.
//
// Survivors do NOT become records inside the walk: they wait in a small per-wavefront queue in LDS (WalkQueue: x, y and the five blobCenter values a
// record needs) and the wavefront turns them into records (blobList.cl:57-102) AFTER its row loop, when the row windows are dead.  Emitting on the spot
// kept the disc statistics' ~27 VGPRs and ~26 SGPRs alive beside the windows across the whole loop: k_circle_walk<5> 75 VGPRs / 106 SGPRs = 6 wavefronts
// per SIMD (both limits; MI355X_MICROARCH.md, Residency) against 8 now -- and the walk is a latency chain whose time is memory round trip / rows in
// flight.  A queue that is about to run out of room (more than WQ_STOP candidates: clustered robots, or a pathological frame in which every pixel is
// a plateau peak) ends the walk after the current row; the wavefront drains the queue and walks again from the next row (WalkStop).
// 256-thread workgroups are admitted per CU up to floor(800 / (ceil(sgpr / 16) * 16 + 16)) (MI355X_MICROARCH.md, Residency): 8 needs <= 80 SGPRs.  The
// walkers' 21 kernel arguments tempt the compiler to 98-106; capped, it re-derives them where they are used (18-23 scalar spills, none in the row loop).
#define VP_WALK_SGPRS __attribute__((amdgpu_num_sgpr(80)))
constexpr int WQ_CAP = 96, WQ_STOP = WQ_CAP - 64, WQ_WORDS = 8; // 3 KB per wavefront; a row appends at most 58 candidates (62 - 2r lanes report)
struct WalkQueue {
	uint32_t* q; // LDS, this wavefront's [WQ_CAP][WQ_WORDS]: {x, y, v, left | right, up, down, -}
	int count;   // wave-uniform
};
struct WalkStop { // wave-uniform loop state the queue may cut short
	unsigned det_n; // rows still to be tested (the range test of the row loop)
	int nsteps;     // row-loop bound
	int y_resume;   // first row NOT yet tested when the walk ends
};
struct WalkEmit {
	const uint32_t* flat; // this frame
	int32_t* counter;     // this frame's [3]
	uint8_t* rec;         // this frame's append-order records
	uint32_t* idx;
	float min_score;
	int list_radius, max_matches;
};
template<int R>
__device__ __forceinline__ void walk_detect(int m_up, int m_mid, int m_dn, bool col_ok, uint64_t col_mask /* ballot of col_ok */, bool x_first, bool x_last, int x, int y, int w, int h, int m_thr,
		int& rej, WalkQueue& wq, WalkStop& ws, int lane) {
	// blobList.cl:39 for the whole wavefront: one compare, one scalar AND with the (loop-invariant) mask of the reporting lanes, one branch
	if((__builtin_amdgcn_ballot_w64(m_mid >= m_thr) & col_mask) == 0) return; // wave-uniform
	const bool above = col_ok && m_mid >= m_thr;
	const float v = walk_quot<R>(m_mid);
	float lf = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false)); // wave_shr:1: lane i <- lane i-1
	float rt = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false)); // wave_shl:1: lane i <- lane i+1
	if(x_first) lf = v;
	if(x_last) rt = v;
	const float uv = y == 0 ? v : walk_quot<R>(m_up), dv = y == h - 1 ? v : walk_quot<R>(m_dn);
	const bool peak = !(lf > v || rt > v || uv > v || dv > v); // :47-52, strict '>'
	rej += __popcll(__ballot(above && !peak));                 // :53 counter[2]++
	const bool hit = above && peak;
	const unsigned long long cm = __ballot(hit);
	if(cm == 0) return; // wave-uniform
	if(hit) { // every survivor queues itself: slot = queue length + survivors in lower lanes
		const int slot = wq.count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
		uint4* e = reinterpret_cast<uint4*>(wq.q + slot * WQ_WORDS);
		e[0] = make_uint4((uint32_t)x, (uint32_t)y, __float_as_uint(v), __float_as_uint(lf));
		e[1] = make_uint4(__float_as_uint(rt), __float_as_uint(uv), __float_as_uint(dv), 0u);
	}
	wq.count += __popcll(cm);
	if(wq.count > WQ_STOP) { // the next row might not fit: this row was the walk's last (its survivors are all queued, its rejects counted)
		ws.det_n = 0;
		ws.nsteps = 0;
		ws.y_resume = y + 1;
	}
}

// The queued survivors of a walk, after the row loop.  Two ways out:
//  * gq != nullptr (large launches): they are handed to the FRAME's queue in global memory (one atomic per wavefront reserves the slots, the entries
//    are copied 8 words each) and the wavefront is done; k_emit_queue, the next launch, turns every frame's queue into records with one wavefront
//    per survivor.  Emitting inside the walker costs a record's whole latency chain (disc pixels from HBM -> wave reductions -> returning atomic ->
//    stores) while the workgroup's other three wavefronts wait at the arrival barrier with their slots occupied: 51 of 337 us at 128 frames.
//  * otherwise, and for entries the frame's queue has no room for: records on the spot (blobList.cl:57-102), one at a time by the whole wavefront.
__device__ __forceinline__ void walk_drain(WalkQueue& wq, const WalkEmit& em, int w, int h, int lane, uint32_t* gq /* this frame's queue or nullptr */, int32_t* gq_count, int gq_cap) {
	int first = 0;
	if(gq && wq.count) { // wave-uniform
		int base = 0;
		if(lane == 0) base = atomicAdd(gq_count, wq.count);
		base = __builtin_amdgcn_readfirstlane(base);
		first = max(0, min(wq.count, gq_cap - base)); // entries that fit (the counter may run past the capacity: readers clamp it)
		for(int i = lane; i < first * WQ_WORDS; i += 64) gq[(size_t)base * WQ_WORDS + i] = wq.q[i]; // read by the NEXT launch: plain stores
	}
	for(int i = first; i < wq.count; i++) { // wave-uniform
		const uint4 a = *reinterpret_cast<const uint4*>(wq.q + i * WQ_WORDS), b = *reinterpret_cast<const uint4*>(wq.q + i * WQ_WORDS + 4); // same address in every lane: broadcast
		emit_record(em.flat, w, h, (int)a.x, (int)a.y, __uint_as_float(a.z), __uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), em.min_score, em.list_radius,
				em.max_matches, em.counter, em.rec, em.idx, lane);
	}
	wq.count = 0;
}

// smallest integer m with !(quotient(m) < thr), by bisection (host; the same correctly rounded binary32 division the kernels' quotient equals)
static int walk_threshold(float thr, int r) {
	if(thr != thr) return INT32_MIN; // `v < NaN` is false for every v: everything passes (blobList.cl:39)
	const float div = (float)(r * r);
	auto passes = [&](long long m) { return !((float)(int)m / div < thr); };
	if(!passes(INT32_MAX)) return INT32_MAX; // (+inf, or beyond every table value: nothing passes short of INT32_MAX itself)
	long long lo = INT32_MIN, hi = INT32_MAX; // passes(hi) holds
	if(passes(lo)) return INT32_MIN;
	while(hi - lo > 1) {
		const long long mid = lo + (hi - lo) / 2;
		if(passes(mid)) hi = mid;
		else lo = mid;
	}
	return (int)hi;
}

// The walkers' second launch (large launch groups): every frame's queue of survivors -> records (blobList.cl:57-102) -> the frame's sorted list, by
// ONE 1024-thread workgroup per frame.  Everything that made record emission slow inside the walkers is a serial chain of memory round trips per
// record -- disc pixels, the returning atomic_inc, write-through stores, the arrival count, the sorting workgroup reading it all back (16.6 k records
// of 128 frames: 51 us inside the walkers, 64 us as one wavefront per record in a launch of its own).  Here 8 lanes make one record, 128 records at
// a time per workgroup: all disc pixels of those records are requested at once (ONE round trip), the sums are reduced over the 8 lanes by DPP, the
// append slots come from a workgroup prefix sum (blobList.cl:87's atomic_inc once per frame, not per record), the records wait in LDS and are
// ranked there by pixel index (RASTER) and stored straight into the caller's arrays.  Only a frame whose queue overflowed (the walkers then made
// the remaining records themselves, walk_drain) goes through the append-order arrays and the generic sort.
constexpr int EQ_NT = 1024, EQ_GROUP = 8, EQ_PER_PASS = EQ_NT / EQ_GROUP, WALK_GQ_MAX = 1536; // WALK_GQ_MAX: queue entries per frame (LDS staging of the records: 28 B each)

__device__ __forceinline__ uint32_t group8_sum(uint32_t v) { // sum over the 8 lanes of a group (lanes 8g .. 8g+7 of a 16-lane DPP row); every lane gets it
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false); // row_half_mirror: lane i <- lane 7 - i of its half row (the other quad)
	return v;
}

// generic RASTER sort of the append-order arrays by NT threads (all of the workgroup): rank = number of records with a smaller pixel index
template<int NT>
__device__ __forceinline__ void sort_append_order(int nrec, const uint8_t* rec_work, const uint32_t* idx_work, uint8_t* rec_sorted, uint32_t* idx_sorted, uint32_t* tile /* LDS [NT] */) {
	const int t = (int)threadIdx.x;
	for(int t0 = 0; t0 < nrec; t0 += NT) {
		const int tt = t0 + t;
		const uint32_t mine = tt < nrec ? ld_wt32(idx_work + tt) : 0;
		int rank = 0;
		for(int j0 = 0; j0 < nrec; j0 += NT) {
			const int j = j0 + t;
			tile[t] = j < nrec ? ld_wt32(idx_work + j) : 0xFFFFFFFFu;
			__syncthreads();
			const int m = min(NT, nrec - j0);
			for(int j2 = 0; j2 < m; j2++) rank += tile[j2] < mine;
			__syncthreads();
		}
		if(tt < nrec) {
			const uint16_t* sp = reinterpret_cast<const uint16_t*>(rec_work + (size_t)tt * 22);
			uint16_t* dp = reinterpret_cast<uint16_t*>(rec_sorted + (size_t)rank * 22);
#pragma unroll
			for(int k = 0; k < 11; k++) dp[k] = ld_wt16(sp + k);
			if(idx_sorted) idx_sorted[rank] = mine;
		}
	}
}

__global__ __launch_bounds__(EQ_NT) void k_emit_queue(const uint32_t* __restrict__ flat, const uint32_t* __restrict__ gq, int32_t* __restrict__ gq_count, int gq_cap,
		int32_t* __restrict__ counter, uint8_t* __restrict__ rec_work, uint32_t* __restrict__ idx_work, int w, int h, float min_score, int radius, int max_matches, int order,
		uint8_t* __restrict__ rec_sorted, uint32_t* __restrict__ idx_sorted) {
	__shared__ uint32_t s_idx[WALK_GQ_MAX];      // pixel index of the accepted records, append order
	__shared__ uint16_t s_rec[WALK_GQ_MAX * 12]; // their 22-byte records (24-byte pitch)
	__shared__ uint32_t tile[EQ_NT];
	__shared__ int s_acc[EQ_NT / 64], s_rej[EQ_NT / 64], s_base;
	const int z = blockIdx.x, t = (int)threadIdx.x, lane = t & 63, wv = t >> 6, gl = t & (EQ_GROUP - 1), grp = t / EQ_GROUP;
	const size_t n = (size_t)w * h;
	const uint32_t* f = flat + z * n;
	counter += 3 * z;
	rec_work += (size_t)z * max_matches * 22;
	idx_work += (size_t)z * max_matches;
	rec_sorted += (size_t)z * max_matches * 22;
	if(idx_sorted) idx_sorted += (size_t)z * max_matches;
	const int queued = gq_count[z];
	const int total = min(queued, gq_cap);
	const bool overflowed = queued > gq_cap; // the walkers made the remaining records themselves: they sit in the append-order arrays already
	const uint32_t* q = gq + (size_t)z * gq_cap * WQ_WORDS;
	const int D = 2 * radius + 1, DD = D * D, sq = radius * radius;
	const float invD = 1.0f / (float)D;
	int nacc = 0, nrej = 0; // workgroup-uniform running totals
	__syncthreads();        // every thread has read the queue length ...
	if(t == 0) gq_count[z] = 0; // ... before it is put back to zero for the next launch (self-cleaning)
	for(int k0 = 0; k0 < total; k0 += EQ_PER_PASS) { // workgroup-uniform
		const int k = k0 + grp;
		const bool active = k < total;
		uint4 ea = make_uint4(0, 0, 0, 0), eb = ea;
		if(active) {
			ea = *reinterpret_cast<const uint4*>(q + (size_t)k * WQ_WORDS);
			eb = *reinterpret_cast<const uint4*>(q + (size_t)k * WQ_WORDS + 4);
		}
		const int px = (int)ea.x, py = (int)ea.y;
		// disc statistics (blobList.cl:57-73): element e of the (2R+1)^2 square goes to lane e % 8 of the group; four requests in flight per round
		uint32_t s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0}, cnt = 0;
		for(int e0 = gl; e0 < DD; e0 += 4 * EQ_GROUP) {
			uint32_t pv[4];
			bool in[4];
#pragma unroll
			for(int u = 0; u < 4; u++) {
				const int e = e0 + u * EQ_GROUP;
				const int row = (int)(((float)e + 0.5f) * invD); // e / D, exact for e < 2^12 (the fraction is at least 0.5 / D away from an integer)
				const int dy = row - radius, dx = e - row * D - radius;
				in[u] = active && e < DD && dx * dx + dy * dy <= sq;
				pv[u] = in[u] ? f[(size_t)vp_clampi(py + dy, 0, h - 1) * w + vp_clampi(px + dx, 0, w - 1)] : 0u;
			}
#pragma unroll
			for(int u = 0; u < 4; u++) {
				if(in[u]) {
#pragma unroll
					for(int c = 0; c < 3; c++) {
						const uint32_t b = (pv[u] >> (8 * c)) & 0xFF;
						s1[c] += b;
						s2[c] += b * b;
					}
					cnt++;
				}
			}
		}
		const uint32_t center = active ? f[(size_t)py * w + px] : 0u;
		cnt = group8_sum(cnt);
		const float nf = (float)(int)cnt;
		float sd[3];
		uint32_t color[3];
#pragma unroll
		for(int c = 0; c < 3; c++) {
			const uint32_t a1 = group8_sum(s1[c]), a2 = group8_sum(s2[c]);
			const float a = (float)a1;
			const float qq = a * a / nf;
			sd[c] = sqrtf(((float)a2 - qq) / nf); // :76 (native_sqrt -> IEEE sqrt)
			color[c] = cnt ? a1 / cnt : 0u;        // :96 uint division
		}
		const float v = __uint_as_float(ea.z), nx = __uint_as_float(ea.w), qx = __uint_as_float(eb.x), ny = __uint_as_float(eb.y), qy = __uint_as_float(eb.z);
		const float score = v / (sd[0] + sd[1] + sd[2]); // :78
		const bool leader = active && gl == 0;
		const bool rejected = leader && score < min_score; // :79-82 (NaN is kept)
		const bool accepted = leader && !rejected;
		// append slots in queue order: survivors of lower groups first (blobList.cl:87, one workgroup-wide prefix sum instead of an atomic per record)
		const unsigned long long ma = __ballot(accepted), mr = __ballot(rejected);
		if(lane == 0) {
			s_acc[wv] = __popcll(ma);
			s_rej[wv] = __popcll(mr);
		}
		__syncthreads();
		int before = 0, pass_acc = 0, pass_rej = 0;
#pragma unroll
		for(int i = 0; i < EQ_NT / 64; i++) {
			before += i < wv ? s_acc[i] : 0;
			pass_acc += s_acc[i];
			pass_rej += s_rej[i];
		}
		const int slot = nacc + before + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u));
		if(accepted && slot < WALK_GQ_MAX) {
			const float mx = (float)px + 0.5f * (nx - qx) / (nx - 2 * v + qx); // :93
			const float my = (float)py + 0.5f * (ny - qy) / (ny - 2 * v + qy); // :94
			const uint32_t xb = __float_as_uint(mx), yb = __float_as_uint(my), cb = __float_as_uint(v), sb = __float_as_uint(score);
			uint16_t* d = s_rec + slot * 12;
			d[0] = (uint16_t)xb; d[1] = (uint16_t)(xb >> 16);
			d[2] = (uint16_t)yb; d[3] = (uint16_t)(yb >> 16);
			d[4] = (uint16_t)(color[0] | (color[1] << 8));
			d[5] = (uint16_t)(color[2] | ((center & 0xFF) << 8));
			d[6] = (uint16_t)(((center >> 8) & 0xFF) | (((center >> 16) & 0xFF) << 8));
			d[7] = (uint16_t)cb; d[8] = (uint16_t)(cb >> 16);
			d[9] = (uint16_t)sb; d[10] = (uint16_t)(sb >> 16);
			s_idx[slot] = (uint32_t)(py * w + px);
		}
		nacc += pass_acc;
		nrej += pass_rej;
		__syncthreads(); // s_acc / s_rej are rewritten by the next pass
	}
	// counters (blobList.cl:80,87): once per frame
	if(t == 0) {
		if(nrej) atomicAdd(&counter[1], nrej);
		s_base = nacc ? atomicAdd(&counter[0], nacc) : (overflowed ? __hip_atomic_load(&counter[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
	}
	__syncthreads();
	const int base = s_base;
	const bool raster = order == VP_ORDER_RASTER;
	if(!overflowed) { // (base == 0: nobody else appended to this frame) the records are all here: rank in LDS, store once
		const int nrec = min(nacc, max_matches); // :88-89: entries past the capacity are never written
		for(int i = t; i < nrec; i += EQ_NT) {
			const uint32_t mine = s_idx[i];
			int rank = i;
			if(raster) {
				rank = 0;
				for(int j = 0; j < nrec; j++) rank += s_idx[j] < mine;
			}
			const uint16_t* sp = s_rec + i * 12;
			uint16_t* dp = reinterpret_cast<uint16_t*>((raster ? rec_sorted : rec_work) + (size_t)rank * 22);
#pragma unroll
			for(int k2 = 0; k2 < 11; k2++) dp[k2] = sp[k2];
			uint32_t* ip = raster ? idx_sorted : idx_work;
			if(ip) ip[rank] = mine;
		}
		return;
	}
	// overflowed queue: our records join the walkers' in the append-order arrays (write-through: the sort below reads them back with sc1 loads)
	for(int i = t; i < nacc; i += EQ_NT) {
		const int dst = base + i;
		if(dst < max_matches) {
			const uint16_t* sp = s_rec + i * 12;
			uint16_t* dp = reinterpret_cast<uint16_t*>(rec_work + (size_t)dst * 22);
#pragma unroll
			for(int k2 = 0; k2 < 11; k2++) st_wt16(dp + k2, sp[k2]);
			__hip_atomic_store((gu32_t*)(idx_work + dst), s_idx[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	if(!raster) return;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	const int nrec = min(__hip_atomic_load(&counter[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), max_matches);
	sort_append_order<EQ_NT>(nrec, rec_work, idx_work, rec_sorted, idx_sorted, tile);
}

// ---------------------------------------------------------------- circle + detect, fused --------------------------------------
// satBlobCenter.cl:34-42 + the detection half of blobList.cl:37-55 straight from the SAT; blobCenter never reaches memory.
//
// All four quadrant sums of satBlobCenter.cl:37-40 are ONE box-sum image sampled at four offsets:
//   Bx(u, v) = S(u+r-1, v+r-1) - S(u+r-1, v) - S(u, v+r-1) + S(u, v)        (every tap clamped on its own, as in the reference)
//   pp(x, y) = Bx(x+1, y+1),  nn = Bx(x-r, y-r),  pn = -Bx(x+1, y-r),  np = -Bx(x-r, y+1)
// and Bx separates into row differences: with  D1(row) = S(x+r, row) - S(x+1, row)  and  D2(row) = S(x-1, row) - S(x-r, row)
//   pp = D1(y+r) - D1(y+1),   -Bx(x+1, y-r) = D1(y-r) - D1(y-1),   -Bx(x-r, y+1) = D2(y+1) - D2(y+r),   nn = D2(y-1) - D2(y-r).
// So a lane that owns image column x and WALKS DOWN the rows needs four SAT values of row y+r -- columns x+r, x+1, x-1, x-r -- and
// keeps the last 2r+1 values of D1 and D2 in registers: ~25 VALU instructions per pixel, no LDS memory, no barrier, no index
// arithmetic in the loop.  The four values are ONE coalesced dword load per lane (its own column, clamp-to-edge folded into the
// per-lane column offset and the scalar row index) shifted across the wavefront: +-1 by DPP wave shifts, +-r by ds_bpermute (the LDS
// crossbar, no LDS memory).  [Four loads per lane and row kept the texture addresser busy 84 % of the kernel; one load does not.]
// Wrapping int32 differences are exact; min(min(pp,nn),min(pn,np)) is taken on the integers and converted once
// (int -> float is monotonic, so the converted minimum is the minimum of the converted values).
// The peak test needs blobCenter of the four neighbours: up / down are the lane's own previous / next row, left / right come from
// the neighbouring lanes by DPP wave shifts again.  A wavefront therefore yields 62 - 2r image columns (lanes r+1 .. 62-r; the
// lanes around them only carry the neighbours' columns) and `seg_h` rows plus the 2r+2 rows it needs to get going.  Neighbours
// outside the image take the pixel's own value: that is what the reference's clamped reads return, and `own > own` is false.

template<int RT, bool BCOUT>
__global__ __launch_bounds__(256) VP_WALK_SGPRS void k_circle_walk(const int32_t* __restrict__ sat_all, float* __restrict__ bc_out, const uint32_t* __restrict__ flat, int32_t* __restrict__ done,
		int32_t* __restrict__ counter, uint8_t* __restrict__ rec_work, uint32_t* __restrict__ idx_work, int w, int h, int seg_h, int strips_x, int items_per_frame, int wgs_per_frame,
		float thr, int m_thr, float min_score, int list_radius, int max_matches, int order, uint8_t* __restrict__ rec_sorted, uint32_t* __restrict__ idx_sorted,
		uint32_t* __restrict__ gq /* frames' survivor queues, or nullptr: records are made here */, int32_t* __restrict__ gq_count, int gq_cap) {
	__shared__ uint32_t tile[256];
	__shared__ int s_last;
	__shared__ __attribute__((aligned(16))) uint32_t wqueue[4 * WQ_CAP * WQ_WORDS];
	// PF rows of loads in flight per lane (a walk is one dependent chain per lane).  With 4 the kernel held 7 MB in flight on the whole chip, half of
	// what HBM latency x bandwidth asks for: 128 frames 0.379 ms; 8 and more: 0.354 (12 and 16 no better).  WL (window length, >= 2r+1) must be a
	// multiple of PF, so PF is chosen per radius to keep the windows tight: 2r+1 itself up to 13, about half of it above.
	constexpr int R = RT, NW = 2 * R + 1, PF = NW <= 13 ? (NW < 8 ? 8 : NW) : (NW + 1) / 2, WL = ((NW + PF - 1) / PF) * PF, USE = 62 - 2 * R; // USE: image columns a wavefront yields
	const int lane = threadIdx.x & 63;
	// A workgroup = 4 independent wavefronts of ONE frame (the frame's work items are padded to a multiple of 4, so that the workgroup can
	// take part in that frame's arrival count).  XCD-aware: each XCD walks a contiguous run of workgroups (strips of one segment row are
	// neighbours), so the columns adjacent strips share are L2 hits.  Speed only.
	const unsigned lb = vp_xcd_remap(blockIdx.x, gridDim.x);
	const int z = (int)(lb / (unsigned)wgs_per_frame);
	const int item = __builtin_amdgcn_readfirstlane((int)(lb - (unsigned)z * (unsigned)wgs_per_frame) * 4 + (int)(threadIdx.x >> 6));
	const size_t n = (size_t)w * h;
	if(item < items_per_frame) { // wave-uniform: the padding wavefronts only take part in the arrival below
	const int strip = item % strips_x, seg = item / strips_x;
	const int ya = seg * seg_h, yb = min(h, ya + seg_h);
	const int x = strip * USE - (R + 1) + lane;
	const uint32_t oe = 4u * (uint32_t)vp_clampi(x, 0, w - 1);
	int perm_hi = 4 * ((lane + R) & 63), perm_lo = 4 * ((lane - R) & 63); // ds_bpermute byte addresses: lane l reads lane l + r / l - r (mod 64)
	asm volatile("" : "+v"(perm_hi), "+v"(perm_lo)); // opaque: keeps the compiler from splitting them into base + the instruction's offset field
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(sat_all + z * n), 0, (int)(n * 4), 0x00020000); // host: n * 4 < 2^31
	const uint32_t row_bytes = (uint32_t)w * 4u;
	const bool col_ok = lane >= R + 1 && lane <= 62 - R && x < w; // (x >= 0 for these lanes)
	const uint64_t col_mask = __builtin_amdgcn_ballot_w64(col_ok);
	const bool x_first = x == 0, x_last = x == w - 1;
	int rej = 0;
	WalkQueue wq = {wqueue + (threadIdx.x >> 6) * (WQ_CAP * WQ_WORDS), 0};
	for(int ys = ya; ys < yb;) { // one walk from row ys to the segment's end, or to the row after which the survivor queue had to be drained
	// Windows of the last WL >= 2r+1 rows of -D1 / D2, and the raw SAT values of the next PF rows in flight: a row's load is
	// issued PF steps before it is used (a walk is one dependent chain per lane; without the run-ahead every step would wait a full
	// memory round trip).  WL is a multiple of PF and the loop is unrolled WL-fold, so every array index below is a constant.
	int D1[WL], D2[WL], Pe[PF];
#pragma unroll
	for(int j = 0; j < WL; j++) D1[j] = D2[j] = 0;
	int m_up = 0, m_mid = 0; // integer quadrant minimum (blobCenter * r^2) of rows t-2, t-1 while row t is being formed
	const int s0 = ys - 1 - 2 * R;                 // step k feeds SAT row t + r and completes blobCenter row t = s0 + k: row ys - 1 at k = 2r, when the windows are full
	WalkStop ws;
	ws.nsteps = 2 * R + (yb - ys) + 2;
	ws.y_resume = yb;
	const int row_hi = min(h - 1, s0 + ws.nsteps - 1 + R); // the run-ahead past the last step re-reads the segment's last row (a cache hit) instead of rows nobody needs
	// The scalar side of a row step is kept short on purpose (it was 19 scalar instructions against 12 vector ones and filled 61 % of the scalar
	// issue slots): the byte offset of the next row to request runs along as ONE scalar (add, max, min: clamp-to-edge of satBlobCenter.cl:20 and
	// the segment's last row), and "windows full and row inside the segment" is ONE unsigned range test.
	int so_next = (s0 + R) * (int)row_bytes;            // row s0 + K + R of the K-th request (negative above the image: clamped when used)
	const int so_hi = row_hi * (int)row_bytes;
	ws.det_n = (unsigned)max(0, yb - s0 + 1 - (2 * R + 2)); // steps 2r+2 .. 2r+2+det_n-1 complete a row y = s0 + k - 1 < yb with full windows
#define VP_WK_ISSUE(SLOT, K)                                                                                                           \
	do {                                                                                                                               \
		Pe[SLOT] = __builtin_amdgcn_raw_buffer_load_b32(rs, oe, (uint32_t)min(max(so_next, 0), so_hi), 0); /* request number K, in order */ \
		so_next += (int)row_bytes;                                                                                                     \
	} while(0)
#pragma unroll
	for(int p = 0; p < PF; p++) VP_WK_ISSUE(p, p);
	for(int kb = 0; kb < ws.nsteps; kb += WL) {
#pragma unroll
		for(int j = 0; j < WL; j++) {
			const int k = kb + j;
			{
				const int e = Pe[j % PF]; // S(clamp(x), row); the taps at clamp(x +- 1), clamp(x +- r) sit in the neighbouring lanes
				const int sa = __builtin_amdgcn_ds_bpermute(perm_hi, e), sd = __builtin_amdgcn_ds_bpermute(perm_lo, e);
				const int sb = __builtin_amdgcn_update_dpp(0, e, 0x130, 0xf, 0xf, false), sc = __builtin_amdgcn_update_dpp(0, e, 0x138, 0xf, 0xf, false); // lane + 1, lane - 1
				// NOTE the operand order: a DPP-shifted value is only ever the MINUEND.  hipcc folds the shift into the subtraction, and for a
				// shifted SUBTRAHEND it emits v_subrev_u32_dpp, which on gfx950 (ROCm 7.2) shifts the OTHER operand: a - shl(e) comes out as
				// shl(a) - e (tools/dbg/dpp_test.hip shows it in isolation; tests/test_capi_symbols.py checks that no such instruction is
				// in the library).  So the first window holds N1 = -D1 = S(x+1) - S(x+r).
				D1[j] = (int)((uint32_t)sb - (uint32_t)sa);
				D2[j] = (int)((uint32_t)sc - (uint32_t)sd);
			}
			VP_WK_ISSUE(j % PF, k + PF);
			const int jp1 = (j - R + 1 + 2 * WL) % WL, jm1 = (j - R - 1 + 2 * WL) % WL, jmr = (j - 2 * R + 2 * WL) % WL; // rows t+1, t-1, t-r (row t+r is j)
			const int pp = (int)((uint32_t)D1[jp1] - (uint32_t)D1[j]), pn = (int)((uint32_t)D1[jm1] - (uint32_t)D1[jmr]); // D1[] holds -D1 (see above)
			const int np = (int)((uint32_t)D2[jp1] - (uint32_t)D2[j]), nn = (int)((uint32_t)D2[jm1] - (uint32_t)D2[jmr]);
			const int m_dn = min(min(pp, nn), min(pn, np)); // blobCenter(x, t) * r^2, t = s0 + k (satBlobCenter.cl:41)
			const int t = s0 + k;
			if(BCOUT && k >= 2 * R && t >= ys && t < yb && col_ok) bc_out[z * n + (size_t)t * w + x] = walk_quot<R>(m_dn);
			const int y = t - 1; // the row whose four neighbours are now known
			if((unsigned)(k - (2 * R + 2)) < ws.det_n) walk_detect<R>(m_up, m_mid, m_dn, col_ok, col_mask, x_first, x_last, x, y, w, h, m_thr, rej, wq, ws, lane); // scalar condition
			m_up = m_mid;
			m_mid = m_dn;
		}
	}
#undef VP_WK_ISSUE
	// the walk's survivors -> records, now that the windows are dead (see WalkQueue)
	const WalkEmit em = {flat + z * n, counter + 3 * z, rec_work + (size_t)z * max_matches * 22, idx_work + (size_t)z * max_matches, min_score, list_radius, max_matches};
	walk_drain(wq, em, w, h, lane, gq ? gq + (size_t)z * gq_cap * WQ_WORDS : nullptr, gq_count + z, gq_cap);
	ys = ws.y_resume;
	}
	if(rej != 0 && lane == 0) atomicAdd(&counter[3 * z + 2], rej);
	} // item < items_per_frame
	if(gq) return; // k_emit_queue makes the records and sorts them
	frame_arrive_and_sort(done + (size_t)z * ARRIVE_WORDS, (int)(lb - (unsigned)z * (unsigned)wgs_per_frame), wgs_per_frame, nullptr, counter + 3 * z, rec_work + (size_t)z * max_matches * 22, idx_work + (size_t)z * max_matches, max_matches, order,
			rec_sorted + (size_t)z * max_matches * 22, idx_sorted ? idx_sorted + (size_t)z * max_matches : nullptr, tile, &s_last);
}


// ---------------------------------------------------------------- VP_SAT_LOCAL: gradientDot + circle + blobList from flat -------------
// The reference keeps its SAT private to rgba2blobCenter (Resources.cpp:155-163): nothing outside reads it, and what `circle` takes from it are
// four box sums of gradientDot.  With S the table and every tap clamped to the image (satBlobCenter.cl:20,37-40) the four quadrant terms are
//     pp =  B(x+2, y+r)     np = -B(x-r+1, y+r)     nn =  B(x-r+1, y-1)     pn = -B(x+2, y-1)
// where B(cx, ry) is the sum of gradientDot over the (r-1) x (r-1) box of columns cx .. cx+r-2 and rows ry-r+2 .. ry, taken on gradientDot
// with its FIRST ROW AND FIRST COLUMN SET TO ZERO and zero outside the image.  (Why: S(x+r) - S(x+1) is the sum of the table's columns
// x+2 .. x+r; a clamped upper tap ends the box at the image edge, which is zero extension; a clamped lower tap -- column x-r < 0 read as
// column 0 -- subtracts column 0 as well, and column 0 lies in the box exactly in those cases; rows likewise.)  So no table is needed at all:
// a lane owns a column and walks down the rows; per row it loads one flat pixel, forms gradientDot from the flat values of the lanes +- o
// (ds_bpermute) and of its own column o rows up / down (register window), keeps the sum of the last r-1 rows of its column (add the new
// value, subtract the one from r-1 rows ago) and sums that over r-1 neighbouring lanes by doubling (one DPP add, then ds_bpermute + add per
// level): one row of B.  The lane keeps B of its own column start (for pp / pn) and, shifted by r+1 lanes, of the box left of it (nn / np)
// for the last r+2 rows, takes the integer minimum, and from there on it is the walker above: integer threshold test, peak test, record
// emission, sort by the frame's last workgroup.  ~18 VALU instructions and 4 cross-lane operations per pixel.
// HBM traffic per frame: flat once (+ strip overlap) instead of flat + the SAT written and read back: 53.5 MB -> ~31 MB at 2048x1536, and
// two launches become one.  Bit-identical to the table paths (exact wrapping int32 sums of the same values).
// Lane l owns detection column x0 + l and the DATA column two to the right (the box of pp starts at x+2): no shift is needed for pp / pn.
// A wavefront yields 63 - 2(r + o) columns.
template<int N>
__device__ __forceinline__ int lane_window_sum(int v, int lane) { // sum of v over lanes l .. l+N-1 (garbage in the last N-1 lanes)
	int t2 = 0, t4 = 0, t8 = 0;
	if(N >= 2) t2 = v + __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false);                      // + lane l+1 (wave_shl:1)
	if(N >= 4) t4 = t2 + __builtin_amdgcn_ds_bpermute(4 * ((lane + 2) & 63), t2);
	if(N >= 8) t8 = t4 + __builtin_amdgcn_ds_bpermute(4 * ((lane + 4) & 63), t4);
	int acc = 0, off = 0;
	if(N & 8) { acc = t8; off = 8; }
	if(N & 4) { acc = off ? acc + __builtin_amdgcn_ds_bpermute(4 * ((lane + off) & 63), t4) : t4; off += 4; }
	if(N & 2) { acc = off ? acc + __builtin_amdgcn_ds_bpermute(4 * ((lane + off) & 63), t2) : t2; off += 2; }
	if(N & 1) { acc = off ? acc + __builtin_amdgcn_ds_bpermute(4 * ((lane + off) & 63), v) : v; off += 1; }
	return acc;
}

template<int RT, int OT>
__global__ __launch_bounds__(256) VP_WALK_SGPRS void k_flat_walk(const uint32_t* __restrict__ flat, int32_t* __restrict__ done, int32_t* __restrict__ counter, uint8_t* __restrict__ rec_work,
		uint32_t* __restrict__ idx_work, int w, int h, int seg_h, int strips_x, int items_per_frame, int wgs_per_frame, float thr, int m_thr, float min_score, int list_radius, int max_matches,
		int order, uint8_t* __restrict__ rec_sorted, uint32_t* __restrict__ idx_sorted, uint32_t* __restrict__ gq, int32_t* __restrict__ gq_count, int gq_cap) {
	__shared__ uint32_t tile[256];
	__shared__ int s_last;
	__shared__ __attribute__((aligned(16))) uint32_t wqueue[4 * WQ_CAP * WQ_WORDS]; // survivors wait here until the row loop is over (WalkQueue)
	constexpr int R = RT, O = OT, PF = 8, NB = R - 1;                        // NB: box side; PF rows of loads in flight per lane (see k_circle_walk: 4 were too few)
	constexpr int WFL = ((2 * O + 1 + PF - 1) / PF) * PF;                     // own-column flat window (rows t'-o .. t'+o), a multiple of PF
	constexpr int RING = ((R + 2 + WFL - 1) / WFL) * WFL;                     // ring length of the row windows: >= r+2, a multiple of WFL; the loop is unrolled RING-fold
	constexpr int LO = R + O + 2, HI = 64 - R - O, USE = HI - LO + 1;         // lanes that report blobs: their boxes, gradientDot taps and peak-test neighbours are in the wavefront
	static_assert(USE >= 16 && RING % WFL == 0 && WFL % PF == 0 && NB >= 1 && NB <= 11, "window geometry");
	const int lane = threadIdx.x & 63;
	const unsigned lb = vp_xcd_remap(blockIdx.x, gridDim.x);
	const int z = (int)(lb / (unsigned)wgs_per_frame);
	const int item = __builtin_amdgcn_readfirstlane((int)(lb - (unsigned)z * (unsigned)wgs_per_frame) * 4 + (int)(threadIdx.x >> 6));
	const size_t n = (size_t)w * h;
	if(item < items_per_frame) { // wave-uniform: the padding wavefronts only take part in the arrival below
	const int strip = item % strips_x, seg = item / strips_x;
	const int ya = seg * seg_h, yb = min(h, ya + seg_h);
	const int x = strip * USE - LO + lane, xd = x + 2; // detection column, data column
	const uint32_t oe = 4u * (uint32_t)vp_clampi(xd, 0, w - 1); // clamp-to-edge on the column (gradientDot.cl:20): the lanes +- o then hold flat(clamp(xd +- o))
	int a_fp = 4 * ((lane + O) & 63), a_fn = 4 * ((lane - O) & 63), a_bn = 4 * ((lane - (R + 1)) & 63);
	asm volatile("" : "+v"(a_fp), "+v"(a_fn), "+v"(a_bn));
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(flat + z * n), 0, (int)(n * 4), 0x00020000); // host: n * 4 < 2^31
	const uint32_t row_bytes = (uint32_t)w * 4u;
	const bool g_ok = lane >= O && lane <= 63 - O && xd >= 1 && xd < w;   // gradientDot of the data column counts: inside the image, not column 0, taps in the wavefront
	const bool col_ok = lane >= LO && lane <= HI && x < w;                 // this lane reports blobs (x >= 0 for these lanes)
	const uint64_t col_mask = __builtin_amdgcn_ballot_w64(col_ok);
	const bool x_first = x == 0, x_last = x == w - 1;
	int rej = 0;
	WalkQueue wq = {wqueue + (threadIdx.x >> 6) * (WQ_CAP * WQ_WORDS), 0};
	for(int ys = ya; ys < yb;) { // one walk from row ys to the segment's end, or to the row after which the survivor queue had to be drained
	uint32_t FW[WFL], Pe[PF]; // own-column flat values of the last WFL rows; flat values of the next PF rows in flight
	int GW[RING], BP[RING], BN[RING]; // gradientDot of the data column, B(x+2, .), B(x-r+1, .) of the last RING rows
#pragma unroll
	for(int j = 0; j < WFL; j++) FW[j] = 0;
#pragma unroll
	for(int j = 0; j < RING; j++) GW[j] = BP[j] = BN[j] = 0;
	int V = 0, m_up = 0, m_mid = 0; // V: sum of the last r-1 gradientDot rows of the data column; m_*: integer quadrant minimum of rows y-2, y-1
	// step k loads flat row f = f0 + k, forms gradientDot row t' = f - o (from k = 2o on: row ys - r, the first row of the first box needed) and
	// the box row B(., t'), the quadrant minimum of row y = t' - r (boxes ending at rows y + r and y - 1) and tests row y - 1
	const int f0 = ys - R - O;
	WalkStop ws;
	ws.nsteps = 2 * R + 2 * O + 1 + (yb - ys);
	ws.y_resume = yb;
	const int row_hi = min(h - 1, f0 + ws.nsteps - 1); // the run-ahead past the last step re-reads the segment's last row (a cache hit)
	int so_next = f0 * (int)row_bytes;                  // scalar side kept short, see k_circle_walk: running byte offset of the next request ...
	const int so_hi = row_hi * (int)row_bytes;
	ws.det_n = (unsigned)max(0, yb + R + O + 1 - f0 - (2 * R + 2 * O + 1)); // ... and one range test: y = f0 + k - O - R - 1 < yb
#define VP_FW_ISSUE(SLOT, K)                                                                                                           \
	do {                                                                                                                               \
		Pe[SLOT] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, oe, (uint32_t)min(max(so_next, 0), so_hi), 0);                   \
		so_next += (int)row_bytes;                                                                                                     \
	} while(0)
#pragma unroll
	for(int p = 0; p < PF; p++) VP_FW_ISSUE(p, p);
	for(int kb = 0; kb < ws.nsteps; kb += RING) {
#pragma unroll
		for(int j = 0; j < RING; j++) {
			const int k = kb + j;
			const uint32_t e = Pe[j % PF] & 0x00FFFFFFu; // colour bytes only (grad_dot_u8)
			VP_FW_ISSUE(j % PF, k + PF);
			FW[j % WFL] = e;
			const int tq = f0 + k - O; // gradientDot / box row t'
			{
				const uint32_t c = FW[(j - O + 2 * WFL) % WFL], yn = FW[(j - 2 * O + 2 * WFL) % WFL]; // flat(xd, t'), flat(xd, t' - o); e = flat(xd, t' + o)
				const uint32_t xp = (uint32_t)__builtin_amdgcn_ds_bpermute(a_fp, (int)c), xn = (uint32_t)__builtin_amdgcn_ds_bpermute(a_fn, (int)c);
				int g = grad_dot_u8(xp, xn, e, yn); // gradientDot.cl:25-29
				if(!(g_ok && k >= 2 * O && tq >= 1 && tq < h)) g = 0; // outside the image, row 0 and column 0 contribute nothing (see above)
				V += g - GW[(j - NB + 2 * RING) % RING]; // the value from r-1 rows ago leaves the window
				GW[j] = g;
			}
			const int bp = lane_window_sum<NB>(V, lane);      // B(x+2, t')
			BP[j] = bp;
			BN[j] = __builtin_amdgcn_ds_bpermute(a_bn, bp);   // B(x-r+1, t'): the box r+1 columns to the left
			const int jo = (j - (R + 1) + 2 * RING) % RING;   // row t' - r - 1 = y - 1
			const int m_dn = min(min(BP[j], BN[jo]), -max(BP[jo], BN[j])); // min(pp, nn, pn, np) of row y = t' - r (satBlobCenter.cl:37-41, times r^2)
			const int y = tq - R - 1; // the row whose four neighbours are now known
			if((unsigned)(k - (2 * R + 2 * O + 1)) < ws.det_n) walk_detect<R>(m_up, m_mid, m_dn, col_ok, col_mask, x_first, x_last, x, y, w, h, m_thr, rej, wq, ws, lane); // scalar condition
			m_up = m_mid;
			m_mid = m_dn;
		}
	}
#undef VP_FW_ISSUE
	const WalkEmit em = {flat + z * n, counter + 3 * z, rec_work + (size_t)z * max_matches * 22, idx_work + (size_t)z * max_matches, min_score, list_radius, max_matches};
	walk_drain(wq, em, w, h, lane, gq ? gq + (size_t)z * gq_cap * WQ_WORDS : nullptr, gq_count + z, gq_cap);
	ys = ws.y_resume;
	}
	if(rej != 0 && lane == 0) atomicAdd(&counter[3 * z + 2], rej);
	} // item < items_per_frame
	if(gq) return; // k_emit_queue makes the records and sorts them
	frame_arrive_and_sort(done + (size_t)z * ARRIVE_WORDS, (int)(lb - (unsigned)z * (unsigned)wgs_per_frame), wgs_per_frame, nullptr, counter + 3 * z,
			rec_work + (size_t)z * max_matches * 22, idx_work + (size_t)z * max_matches, max_matches, order,
			rec_sorted + (size_t)z * max_matches * 22, idx_sorted ? idx_sorted + (size_t)z * max_matches : nullptr, tile, &s_last);
}

// flat -> blob list with no global table (VP_SAT_LOCAL).  (r, o) pairs: every pair ceil(20 / fs), ceil(25 / fs) / 3 can form for r = 2 .. 12.
int vp_sat_common(VpContext* ctx, int stream, const void* in, bool fused, int w, int h, int offset, int batch, int variant, int32_t* sat, float* gd, const char* name, const uint32_t* strips,
		int32_t* zero3, bool* zeroed); // vp_sat.hip
int vp_list_from_blobcenter(VpContext* ctx, int stream, const uint8_t* flat, const float* blobcenter, const int32_t* sat, int circle_radius, float* bc_out,
		int w, int h, int batch, float thr, float min_score, int radius, int max_matches, int order, VpMatch* matches, uint32_t* idx, int32_t* counter, bool counters_zeroed);

// Walkers, two launches (WalkQueue, walk_drain): the survivors go through the frames' queues in global memory -- the candidate list's memory, 8 words
// per entry -- to k_emit_queue.  Measured against records made inside the walker (cfg2, circle + blobList alone): 128 frames 344 -> 295 us, 32: 108 -> 90,
// 8: 44.3 -> 38.8, 4: 34.8 -> 34.3, 1: 24.4 -> 24.0 (profiles/r04_tail_emit_queue.json): never slower, so it is the default at every launch size.
// vp_ctx_tune "tail_emit_kernel" 0 keeps the records inside the walker (also what happens to survivors a full queue has no room for).
static int walk_queue_cap(const VpContext* ctx, int batch, const ListScratch& ws) {
	(void)batch;
	const bool on = ctx->tune.tail_emit_kernel != 0;
	return on && (ws.cap % WQ_WORDS) == 0 ? min(ws.cap / WQ_WORDS, WALK_GQ_MAX) : 0; // 0: records are made inside the walker
}
static int walk_emit_launch(VpContext* ctx, hipStream_t s, const ListScratch& ws, int gq_cap, const uint32_t* flat, int32_t* counter, uint8_t* rec_work, uint32_t* idx_work, int w, int h, int batch,
		float min_score, int radius, int max_matches, int order, uint8_t* rec_sorted, uint32_t* idx_sorted) {
	if(gq_cap <= 0) return VP_OK;
	k_emit_queue<<<batch, EQ_NT, 0, s>>>(flat, ws.cand, ws.cand_count, gq_cap, counter, rec_work, idx_work, w, h, min_score, radius, max_matches, order, rec_sorted, idx_sorted);
	return vp_launch_ok("blobList records");
}

// the (circle radius, gradient offset) pairs k_flat_walk is instantiated for (the VP_FW list below)
bool vp_flat_walk_supported(int r, int o) {
	static const int pairs[][2] = {{2, 1}, {3, 1}, {4, 1}, {5, 2}, {6, 2}, {7, 2}, {7, 3}, {8, 3}, {9, 3}, {9, 4}, {10, 4}, {11, 4}, {12, 4}};
	for(const auto& p : pairs)
		if(p[0] == r && p[1] == o) return true;
	return false;
}

int vp_flat_bloblist_impl(VpContext* ctx, int stream, const uint8_t* flat, int w, int h, int batch, int grad_offset, int circle_radius, float thr, float min_score, int radius,
		int max_matches, int order, VpMatch* matches, uint32_t* idx, int32_t* counter) {
	hipStream_t s = ctx->streams[stream];
	// every argument is checked before anything is enqueued: a refused call leaves the caller's counters and records untouched
	VP_REQUIRE((size_t)w * h * 4 < 0x7fffffffu, "VP_SAT_LOCAL: frame too large");
	if(!vp_flat_walk_supported(circle_radius, grad_offset)) {
		// A (radius, offset) pair without a walker (the pairs real geometries form with r = 2..12 are built; r > 12 needs fieldScale < 1.67 mm per
		// pixel): the same list through a PRIVATE table -- the single-pass SAT into a workspace, then circle + blobList from it.  VP_SAT_LOCAL
		// never refuses a geometry; it is merely no faster than VP_SAT_TILED for these.
		VP_REQUIRE(circle_radius >= 0 && grad_offset >= 0 && grad_offset <= 32, "VP_SAT_LOCAL: circle radius %d / gradient offset %d out of range", circle_radius, grad_offset);
		void* table = nullptr;
		VP_TRY(vp_workspace(ctx, stream, 5, sizeof(int32_t) * (size_t)w * h * batch, &table));
		bool zeroed = false;
		VP_TRY(vp_sat_common(ctx, stream, flat, true, w, h, grad_offset, batch, VP_SAT_TILED, static_cast<int32_t*>(table), nullptr, "gradientDot+sat (private table)", nullptr, counter, &zeroed));
		return vp_list_from_blobcenter(ctx, stream, flat, nullptr, static_cast<const int32_t*>(table), circle_radius, nullptr, w, h, batch, thr, min_score, radius, max_matches, order, matches, idx,
				counter, zeroed);
	}
	const int use = 63 - 2 * (circle_radius + grad_offset);
	ListScratch ws;
	VP_TRY(list_scratch(ctx, stream, batch, max_matches, &ws));
	k_zero<<<vp_div_up(3 * batch, 256), 256, 0, s>>>(counter, 3 * batch); // main.cpp:283-288
	VP_TRY(vp_launch_ok("blobList counters"));
	const bool raster = order == VP_ORDER_RASTER;
	uint32_t* idx_work = (raster || !idx) ? ws.idx : idx;
	uint8_t* rec_work = raster ? ws.rec_tmp : reinterpret_cast<uint8_t*>(matches);
	const int strips_x = (int)vp_div_up(w, use);
	int seg_h = ctx->tune.tail_segment_rows;
	if(seg_h <= 0) {
		seg_h = 32;
		for(int cand_h : {256, 128, 64})
			if((long long)strips_x * vp_div_up(h, cand_h) * batch >= 8192) {
				seg_h = cand_h;
				break;
			}
	}
	const int segs_y = (int)vp_div_up(h, seg_h);
	const long long ipf = (long long)strips_x * segs_y, wpf = (ipf + 3) / 4;
	VP_REQUIRE(wpf * batch < (1ll << 31) && ipf < (1ll << 30), "VP_SAT_LOCAL: too many work items");
	const unsigned grid = (unsigned)(wpf * batch);
	const uint32_t* f32 = reinterpret_cast<const uint32_t*>(flat);
	uint8_t* msorted = reinterpret_cast<uint8_t*>(matches);
	const int m_thr = walk_threshold(thr, circle_radius);
	const int gq_cap = walk_queue_cap(ctx, batch, ws);
	uint32_t* gq = gq_cap ? ws.cand : nullptr;
	bool launched = false;
#define VP_FW(R, OO) \
	if(!launched && circle_radius == R && grad_offset == OO) { \
		k_flat_walk<R, OO><<<grid, 256, 0, s>>>(f32, ws.done, counter, rec_work, idx_work, w, h, seg_h, strips_x, (int)ipf, (int)wpf, thr, m_thr, min_score, radius, max_matches, order, msorted, idx, gq, \
				ws.cand_count, gq_cap); \
		launched = true; \
	}
	VP_FW(2, 1) VP_FW(3, 1) VP_FW(4, 1) VP_FW(5, 2) VP_FW(6, 2) VP_FW(7, 2) VP_FW(7, 3) VP_FW(8, 3) VP_FW(9, 3) VP_FW(9, 4) VP_FW(10, 4) VP_FW(11, 4) VP_FW(12, 4)
#undef VP_FW
	VP_REQUIRE(launched, "VP_SAT_LOCAL: vp_flat_walk_supported() and the kernel list disagree on (%d, %d)", circle_radius, grad_offset);
	VP_TRY(vp_launch_ok("gradientDot+circle+blobList (local table)"));
	return walk_emit_launch(ctx, s, ws, gq_cap, f32, counter, rec_work, idx_work, w, h, batch, min_score, radius, max_matches, order, msorted, idx);
}

// Run-time radius fallback (radii outside the specialised range): the SAT tile (+ r+1 halo) is staged in LDS with clamp-to-edge
// addressing (satBlobCenter.cl:20, H7), the box-sum image Bx and then blobCenter for the 64x64 tile plus a 1-pixel ring are built in
// LDS, then thresholded / peak-tested.  Out-of-image ring positions take the value of the clamped position.
constexpr int FT_W = 64, FT_H = 64, FT_NT = 512;

__global__ __launch_bounds__(FT_NT) void k_circle_detect(const int32_t* __restrict__ sat_all, float* __restrict__ bc_out, uint32_t* __restrict__ cand,
		int32_t* __restrict__ cand_count, int32_t* __restrict__ counter, int w, int h, int r, float thr, int cap) {
	extern __shared__ __attribute__((aligned(16))) char smem_tail[];
	const int halo = r + 1;
	const int RW = FT_W + 2 * halo, RH = FT_H + 2 * halo;
	uint32_t* region = reinterpret_cast<uint32_t*>(smem_tail); // region[ry][rx] = SAT[clamp(y0 - halo + ry)][clamp(x0 - halo + rx)]
	const unsigned tiles_x = (unsigned)(w + FT_W - 1) / FT_W, tiles_y = (unsigned)(h + FT_H - 1) / FT_H;
	const unsigned lb = vp_xcd_remap(blockIdx.x, gridDim.x);
	const int z = (int)(lb / (tiles_x * tiles_y));
	const unsigned trem = lb - (unsigned)z * tiles_x * tiles_y;
	const size_t n = (size_t)w * h;
	const int32_t* sat = sat_all + z * n;
	const int x0 = (int)(trem % tiles_x) * FT_W, y0 = (int)(trem / tiles_x) * FT_H;
	const int tid = threadIdx.x;
	for(int ry = tid >> 6; ry < RH; ry += FT_NT / 64) {
		const int gy = vp_clampi(y0 - halo + ry, 0, h - 1);
		for(int rx = tid & 63; rx < RW; rx += 64) region[ry * RW + rx] = (uint32_t)sat[(size_t)gy * w + vp_clampi(x0 - halo + rx, 0, w - 1)];
	}
	__syncthreads();
	// Bx covers u in [x0-1-r, x0+FT_W+1], v likewise (its origin coincides with the region's); wrapping int32 arithmetic, exact
	const int BXW = FT_W + r + 3, BXH = FT_H + r + 3;
	int32_t* bx = reinterpret_cast<int32_t*>(region + RW * RH);
	for(int e = tid; e < BXW * BXH; e += FT_NT) {
		const int by = e / BXW, bxx = e - by * BXW;
		const uint32_t* c = region + by * RW + bxx;
		bx[e] = (int32_t)(c[(r - 1) * RW + (r - 1)] - c[r - 1] - c[(r - 1) * RW] + c[0]);
	}
	__syncthreads();
	const float div = (float)(r * r);
	constexpr int BW = FT_W + 2, BH = FT_H + 2;
	float* bc = reinterpret_cast<float*>(region); // the blobCenter ring tile aliases the (now dead) SAT region
	for(int e = tid; e < BW * BH; e += FT_NT) {
		const int by = e / BW, bxx = e - by * BW;
		// clamped centre, in Bx coordinates (origin x0-1-r, y0-1-r)
		const int cx = vp_clampi(x0 - 1 + bxx, 0, w - 1) - (x0 - 1 - r), cy = vp_clampi(y0 - 1 + by, 0, h - 1) - (y0 - 1 - r);
		const int32_t* c = bx + cy * BXW + cx;
		bc[e] = (float)min(min(c[BXW + 1], c[-r * BXW - r]), min(-c[-r * BXW + 1], -c[BXW - r])) / div;
	}
	__syncthreads();
	int rej = 0;
	for(int e = tid; e < FT_W * FT_H; e += FT_NT) { // one tile row per wavefront step: every lane of a wavefront takes part
		const int ty = e / FT_W, tx = e - ty * FT_W;
		const int x = x0 + tx, y = y0 + ty;
		const bool in = x < w && y < h;
		const float* p = bc + (ty + 1) * BW + tx + 1;
		const float v = p[0];
		if(bc_out && in) bc_out[z * n + (size_t)y * w + x] = v;
		const bool above = in && !(v < thr);
		const bool peak = !(p[-1] > v || p[1] > v || p[-BW] > v || p[BW] > v);
		rej += __popcll(__ballot(above && !peak));
		wave_append(above && peak, (uint32_t)(y * w + x), cand + (size_t)z * cap, cand_count + z, cap, tid & 63);
	}
	if(rej != 0 && (tid & 63) == 0) atomicAdd(&counter[3 * z + 2], rej);
}

// blobcenter != nullptr: list from a materialised blobCenter image (vp_bloblist: detect -> candidate list -> emit).  Otherwise fused
// from the SAT (vp_blobcenter_list): one launch of the register walker for radii 2..12 (blobCenter stays in registers, `bc_out`
// (optional) receives the debug view); other radii take the LDS fallback + the emit kernel.
// counters_zeroed: the caller's counter[3 x batch] was already cleared on this stream (vp_frame lets the SAT kernel do it).
int vp_list_from_blobcenter(VpContext* ctx, int stream, const uint8_t* flat, const float* blobcenter, const int32_t* sat, int circle_radius, float* bc_out,
		int w, int h, int batch, float thr, float min_score, int radius, int max_matches, int order, VpMatch* matches, uint32_t* idx, int32_t* counter, bool counters_zeroed) {
	hipStream_t s = ctx->streams[stream];
	ListScratch ws;
	VP_TRY(list_scratch(ctx, stream, batch, max_matches, &ws));
	if(!counters_zeroed) { // main.cpp:283-288: the host zeroes counter[3] before every launch
		k_zero<<<vp_div_up(3 * batch, 256), 256, 0, s>>>(counter, 3 * batch);
		VP_TRY(vp_launch_ok("blobList counters"));
	}
	const bool raster = order == VP_ORDER_RASTER;
	uint32_t* idx_work = (raster || !idx) ? ws.idx : idx;
	// RASTER order: the emit kernel appends into workspace records and its last workgroup per frame sorts them into the caller's arrays
	uint8_t* rec_work = raster ? ws.rec_tmp : reinterpret_cast<uint8_t*>(matches);
	if(blobcenter) {
		k_detect<<<dim3(vp_div_up(w, 64), vp_div_up(h, 4), batch), dim3(64, 4), 0, s>>>(blobcenter, ws.cand, ws.cand_count, counter, w, h, thr, ws.cap);
		VP_TRY(vp_launch_ok("blobList detect"));
		k_emit<false><<<dim3(64, batch), 256, 0, s>>>(reinterpret_cast<const uint32_t*>(flat), blobcenter, ws.cand, ws.cand_count, ws.done, counter, rec_work, idx_work,
				w, h, thr, min_score, radius, max_matches, 0, ws.cap, order, reinterpret_cast<uint8_t*>(matches), idx);
		return vp_launch_ok("blobList emit");
	}
	if(circle_radius >= 2 && circle_radius <= 12 && (size_t)w * h * 4 < 0x7fffffffu) {
		// the register walker: circle + detect + record emission + sort in ONE launch, no candidate list
		const int strips_x = (int)vp_div_up(w, 62 - 2 * circle_radius);
		// rows per wavefront: long walks amortise the 2r+2 start-up rows, short ones give a small launch enough wavefronts
		int seg_h = ctx->tune.tail_segment_rows;
		if(seg_h <= 0) {
			seg_h = 16;
			for(int cand_h : {128, 64, 32})
				if((long long)strips_x * vp_div_up(h, cand_h) * batch >= 8192) {
					seg_h = cand_h;
					break;
				}
		}
		const int segs_y = (int)vp_div_up(h, seg_h);
		const long long ipf = (long long)strips_x * segs_y, wpf = (ipf + 3) / 4; // work items and workgroups per frame
		VP_REQUIRE(wpf * batch < (1ll << 31) && ipf < (1ll << 30), "vp_blobcenter_list: too many work items");
		const unsigned grid = (unsigned)(wpf * batch);
		const uint32_t* f32 = reinterpret_cast<const uint32_t*>(flat);
		uint8_t* msorted = reinterpret_cast<uint8_t*>(matches);
		const int m_thr = walk_threshold(thr, circle_radius);
		const int gq_cap = walk_queue_cap(ctx, batch, ws);
		uint32_t* gq = gq_cap ? ws.cand : nullptr;
#define VP_WK(R) \
		case R: \
			if(bc_out) k_circle_walk<R, true><<<grid, 256, 0, s>>>(sat, bc_out, f32, ws.done, counter, rec_work, idx_work, w, h, seg_h, strips_x, (int)ipf, (int)wpf, thr, m_thr, min_score, radius, max_matches, order, msorted, idx, gq, ws.cand_count, gq_cap); \
			else k_circle_walk<R, false><<<grid, 256, 0, s>>>(sat, bc_out, f32, ws.done, counter, rec_work, idx_work, w, h, seg_h, strips_x, (int)ipf, (int)wpf, thr, m_thr, min_score, radius, max_matches, order, msorted, idx, gq, ws.cand_count, gq_cap); \
			break
		switch(circle_radius) { // the radii real geometries produce (ceil(20 mm / fieldScale))
			VP_WK(2); VP_WK(3); VP_WK(4); VP_WK(5); VP_WK(6); VP_WK(7); VP_WK(8); VP_WK(9); VP_WK(10); VP_WK(11); VP_WK(12);
		}
#undef VP_WK
		VP_TRY(vp_launch_ok("circle+blobList"));
		return walk_emit_launch(ctx, s, ws, gq_cap, f32, counter, rec_work, idx_work, w, h, batch, min_score, radius, max_matches, order, msorted, idx);
	} else {
		const int halo = circle_radius + 1;
		const size_t lds = sizeof(uint32_t) * ((size_t)(FT_W + 2 * halo) * (FT_H + 2 * halo) + (size_t)(FT_W + circle_radius + 3) * (FT_H + circle_radius + 3));
		VP_REQUIRE(lds <= 160 * 1024, "vp_blobcenter_list: circle radius %d too large for the LDS tile", circle_radius);
		VP_REQUIRE((long long)vp_div_up(w, FT_W) * vp_div_up(h, FT_H) * batch < (1ll << 31), "vp_blobcenter_list: too many tiles");
		if(lds > 48 * 1024) VP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_circle_detect), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		k_circle_detect<<<dim3((unsigned)(vp_div_up(w, FT_W) * vp_div_up(h, FT_H) * batch)), FT_NT, lds, s>>>(sat, bc_out, ws.cand, ws.cand_count, counter, w, h, circle_radius, thr, ws.cap);
	}
	VP_TRY(vp_launch_ok("circle+detect"));
	k_emit<true><<<dim3(64, batch), 256, 0, s>>>(reinterpret_cast<const uint32_t*>(flat), sat, ws.cand, ws.cand_count, ws.done, counter, rec_work, idx_work,
			w, h, thr, min_score, radius, max_matches, circle_radius, ws.cap, order, reinterpret_cast<uint8_t*>(matches), idx);
	return vp_launch_ok("blobList emit");
}

static int list_args_ok(const char* name, const void* flat, const void* src, int w, int h, int batch, int radius, int max_matches, int order, const void* matches, const void* counter) {
	VP_REQUIRE(flat && src && matches && counter, "%s: null pointer", name);
	VP_REQUIRE(w > 0 && h > 0 && batch > 0 && batch <= 65535 && radius >= 0 && max_matches > 0, "%s: bad shape", name);
	VP_REQUIRE((size_t)w * h < (size_t(1) << 31), "%s: frame too large", name);
	VP_REQUIRE(order == VP_ORDER_ATOMIC || order == VP_ORDER_RASTER, "%s: bad order", name);
	VP_REQUIRE((reinterpret_cast<uintptr_t>(matches) & 1) == 0, "%s: matches must be 2-byte aligned", name);
	return VP_OK;
}

extern "C" int vp_bloblist(VpContext* ctx, int stream, const uint8_t* flat, const float* blobcenter, int w, int h, int batch,
		float circ_threshold, float min_score, int radius, int max_matches, int order, VpMatch* matches, uint32_t* idx, int32_t* counter) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_TRY(list_args_ok("vp_bloblist", flat, blobcenter, w, h, batch, radius, max_matches, order, matches, counter));
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "blobList");
	return vp_list_from_blobcenter(ctx, stream, flat, blobcenter, nullptr, 0, nullptr, w, h, batch, circ_threshold, min_score, radius, max_matches, order, matches, idx, counter, false);
}

extern "C" int vp_blobcenter_list(VpContext* ctx, int stream, const uint8_t* flat, const int32_t* sat, int w, int h, int batch,
		int circle_radius, float circ_threshold, float min_score, int list_radius, int max_matches, int order,
		VpMatch* matches, uint32_t* idx, int32_t* counter, float* blobcenter_out) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_TRY(list_args_ok("vp_blobcenter_list", flat, sat, w, h, batch, list_radius, max_matches, order, matches, counter));
	VP_REQUIRE(circle_radius >= 0, "vp_blobcenter_list: bad circle radius");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "satBlobCenter+blobList");
	return vp_list_from_blobcenter(ctx, stream, flat, nullptr, sat, circle_radius, blobcenter_out, w, h, batch, circ_threshold, min_score, list_radius, max_matches, order, matches, idx, counter, false);
}

extern "C" int vp_flat_bloblist(VpContext* ctx, int stream, const uint8_t* flat, int w, int h, int batch, int grad_offset, int circle_radius, float circ_threshold, float min_score,
		int list_radius, int max_matches, int order, VpMatch* matches, uint32_t* idx, int32_t* counter) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_TRY(list_args_ok("vp_flat_bloblist", flat, flat, w, h, batch, list_radius, max_matches, order, matches, counter));
	VP_REQUIRE(circle_radius >= 0 && grad_offset >= 0, "vp_flat_bloblist: bad radius / offset");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "gradientDot+satBlobCenter+blobList (local)");
	return vp_flat_bloblist_impl(ctx, stream, flat, w, h, batch, grad_offset, circle_radius, circ_threshold, min_score, list_radius, max_matches, order, matches, idx, counter);
}

// ---------------------------------------------------------------- validation kernels -------------------------------
// blobScore.cl:23-67 -- per-pixel score image, -INFINITY where rejected.  One work-item per pixel like the reference.
__global__ __launch_bounds__(256) void k_blobscore(const uint32_t* __restrict__ flat, const float* __restrict__ circ, float* __restrict__ score, int w, int h, float thr, int radius) {
	const int x = blockIdx.x * 64 + threadIdx.x;
	const int y = blockIdx.y * 4 + threadIdx.y;
	if(x >= w || y >= h) return;
	const size_t frame = (size_t)blockIdx.z * w * h;
	const float* c = circ + frame;
	const uint32_t* f = flat + frame;
	const float v = c[(size_t)y * w + x];
	float out = -INFINITY;
	if(!(v < thr) && !(bc_tap(c, w, h, x - 1, y) > v || bc_tap(c, w, h, x + 1, y) > v || bc_tap(c, w, h, x, y - 1) > v || bc_tap(c, w, h, x, y + 1) > v)) {
		uint32_t s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
		int n = 0;
		const int sq = radius * radius;
		for(int dy = -radius; dy <= radius; dy++)
			for(int dx = -radius; dx <= radius; dx++)
				if(dx * dx + dy * dy <= sq) {
					const uint32_t p = f[(size_t)vp_clampi(y + dy, 0, h - 1) * w + vp_clampi(x + dx, 0, w - 1)];
#pragma unroll
					for(int k = 0; k < 3; k++) {
						const uint32_t b = (p >> (8 * k)) & 0xFF;
						s1[k] += b;
						s2[k] += b * b;
					}
					n++;
				}
		const float nf = (float)n;
		float sd[3];
#pragma unroll
		for(int k = 0; k < 3; k++) {
			const float a = (float)s1[k];
			const float q = a * a / nf;
			sd[k] = sqrtf(((float)s2[k] - q) / nf);
		}
		out = v / (sd[0] + sd[1] + sd[2]);
	}
	score[frame + (size_t)y * w + x] = out;
}

extern "C" int vp_blobscore(VpContext* ctx, int stream, const uint8_t* flat, const float* blobcenter, int w, int h, int batch, float circ_threshold, int radius, float* score) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(flat && blobcenter && score && w > 0 && h > 0 && batch > 0 && batch <= 65535 && radius >= 0, "vp_blobscore: bad arguments");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "blobScore");
	k_blobscore<<<dim3(vp_div_up(w, 64), vp_div_up(h, 4), batch), dim3(64, 4), 0, ctx->streams[stream]>>>(reinterpret_cast<const uint32_t*>(flat), blobcenter, score, w, h, circ_threshold, radius);
	return vp_launch_ok("blobScore");
}

// blobCenter.cl:29-63 -- quadrant means over quarter discs, sequential binary32 accumulation in the reference's loop order
__global__ __launch_bounds__(256) void k_circularize(const float* __restrict__ g, float* __restrict__ out, int w, int h, int maxr) {
	const int px = blockIdx.x * 64 + threadIdx.x;
	const int py = blockIdx.y * 4 + threadIdx.y;
	if(px >= w || py >= h) return;
	const size_t frame = (size_t)blockIdx.z * w * h;
	const float* c = g + frame;
	const float sq = ((float)maxr + 0.5f) * ((float)maxr + 0.5f);
	int n = 0;
	float pp = 0.f, pn = 0.f, np = 0.f, nn = 0.f;
	for(int y = 1; y <= maxr; y++)
		for(int x = 1; x <= maxr; x++)
			if((float)(x * x + y * y) <= sq) {
				np += bc_tap(c, w, h, px - x, py + y);
				pp += bc_tap(c, w, h, px + x, py + y);
				nn += bc_tap(c, w, h, px - x, py - y);
				pn += bc_tap(c, w, h, px + x, py - y);
				n++;
			}
	const float nf = (float)n;
	pp /= nf; nn /= nf; pn /= nf; np /= nf;
	const float a = nn < pp ? nn : pp;
	const float b = (-np) < (-pn) ? (-np) : (-pn);
	out[frame + (size_t)py * w + px] = b < a ? b : a;
}

extern "C" int vp_circularize(VpContext* ctx, int stream, const float* graddot, int w, int h, int batch, int min_radius, int max_radius, float* out) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(graddot && out && w > 0 && h > 0 && batch > 0 && batch <= 65535 && max_radius >= 0, "vp_circularize: bad arguments");
	(void)min_radius; // unused by the kernel body as well (blobCenter.cl:37-45 is commented out)
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "blobCenter(legacy)");
	k_circularize<<<dim3(vp_div_up(w, 64), vp_div_up(h, 4), batch), dim3(64, 4), 0, ctx->streams[stream]>>>(graddot, out, w, h, max_radius);
	return vp_launch_ok("circularize");
}
