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
struct ListScratch {
	uint32_t* cand;       // batch x n   candidate pixel indices
	int32_t* cand_count;  // batch
	uint32_t* idx;        // batch x max  (internal idx when the caller passes none)
	uint32_t* idx_tmp;    // batch x max
	uint8_t* rec_tmp;     // batch x max x 22
	float* blobcenter;    // batch x n, only when the fused entry point has to materialise it itself
};

static size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

static int list_scratch(VpContext* ctx, int stream, size_t n, int batch, int max_matches, bool need_bc, ListScratch* s) {
	const size_t b_cand = align256(sizeof(uint32_t) * n * batch), b_cnt = align256(sizeof(int32_t) * batch);
	const size_t b_idx = align256(sizeof(uint32_t) * (size_t)max_matches * batch), b_rec = align256((size_t)22 * max_matches * batch);
	const size_t b_bc = need_bc ? align256(sizeof(float) * n * batch) : 0;
	void* base = nullptr;
	VP_TRY(vp_workspace(ctx, stream, 0, b_cand + b_cnt + 2 * b_idx + b_rec + b_bc, &base));
	uint8_t* p = static_cast<uint8_t*>(base);
	s->cand = reinterpret_cast<uint32_t*>(p); p += b_cand;
	s->cand_count = reinterpret_cast<int32_t*>(p); p += b_cnt;
	s->idx = reinterpret_cast<uint32_t*>(p); p += b_idx;
	s->idx_tmp = reinterpret_cast<uint32_t*>(p); p += b_idx;
	s->rec_tmp = p; p += b_rec;
	s->blobcenter = need_bc ? reinterpret_cast<float*>(p) : nullptr;
	return VP_OK;
}

__device__ __forceinline__ float bc_tap(const float* __restrict__ c, int w, int h, int x, int y) {
	return c[(size_t)vp_clampi(y, 0, h - 1) * w + vp_clampi(x, 0, w - 1)];
}

// blobList.cl:37-55: threshold, 4-neighbour strict-'>' peak test
__global__ __launch_bounds__(256) void k_detect(const float* __restrict__ circ, uint32_t* __restrict__ cand, int32_t* __restrict__ cand_count,
		int32_t* __restrict__ counter, int w, int h, float thr) {
	const int x = blockIdx.x * 64 + threadIdx.x;
	const int y = blockIdx.y * 4 + threadIdx.y;
	if(x >= w || y >= h) return;
	const int z = blockIdx.z;
	const size_t n = (size_t)w * h;
	const float* c = circ + z * n;
	const float v = c[(size_t)y * w + x];
	if(v < thr) return;
	if(bc_tap(c, w, h, x - 1, y) > v || bc_tap(c, w, h, x + 1, y) > v || bc_tap(c, w, h, x, y - 1) > v || bc_tap(c, w, h, x, y + 1) > v) {
		atomicAdd(&counter[3 * z + 2], 1);
		return;
	}
	const int slot = atomicAdd(&cand_count[z], 1);
	cand[z * n + slot] = (uint32_t)(y * w + x);
}

__device__ __forceinline__ void store_match(uint8_t* dst, float x, float y, const uint32_t color[3], uint32_t center, float circ, float score) {
	// 22-byte packed record (blobList.cl:26-32): only 2-byte alignment is guaranteed
	uint16_t* d = reinterpret_cast<uint16_t*>(dst);
	const uint32_t xb = __float_as_uint(x), yb = __float_as_uint(y), cb = __float_as_uint(circ), sb = __float_as_uint(score);
	d[0] = (uint16_t)xb; d[1] = (uint16_t)(xb >> 16);
	d[2] = (uint16_t)yb; d[3] = (uint16_t)(yb >> 16);
	d[4] = (uint16_t)(color[0] | (color[1] << 8));
	d[5] = (uint16_t)(color[2] | ((center & 0xFF) << 8));
	d[6] = (uint16_t)(((center >> 8) & 0xFF) | (((center >> 16) & 0xFF) << 8));
	d[7] = (uint16_t)cb; d[8] = (uint16_t)(cb >> 16);
	d[9] = (uint16_t)sb; d[10] = (uint16_t)(sb >> 16);
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

// blobList.cl:57-102, one wavefront per candidate.  FROM_SAT: blobCenter was never materialised; the five values the record
// needs (centre + 4 neighbours, clamp-to-edge) are recomputed from the SAT by lanes 0-4.
template<bool FROM_SAT>
__global__ __launch_bounds__(256) void k_emit(const uint32_t* __restrict__ flat, const void* __restrict__ src, const uint32_t* __restrict__ cand,
		const int32_t* __restrict__ cand_count, int32_t* __restrict__ counter, uint8_t* __restrict__ matches, uint32_t* __restrict__ idx_out,
		int w, int h, float min_score, int radius, int max_matches, int circle_radius) {
	const int z = blockIdx.y;
	const size_t n = (size_t)w * h;
	const uint32_t* f = flat + z * n;
	const float* c = reinterpret_cast<const float*>(src) + z * n;
	const int32_t* sat = reinterpret_cast<const int32_t*>(src) + z * n;
	const int lane = threadIdx.x & 63;
	const int total = cand_count[z];
	for(int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < total; k += gridDim.x * 4) {
		const uint32_t pix = cand[z * n + k];
		const int px = (int)(pix % (uint32_t)w), py = (int)(pix / (uint32_t)w);
		float v, nx, qx, ny, qy;
		if(FROM_SAT) {
			const int dx = lane == 1 ? -1 : (lane == 2 ? 1 : 0), dy = lane == 3 ? -1 : (lane == 4 ? 1 : 0);
			const float mine = lane < 5 ? circle_at(sat, w, h, vp_clampi(px + dx, 0, w - 1), vp_clampi(py + dy, 0, h - 1), circle_radius, (float)(circle_radius * circle_radius)) : 0.f;
			v = __shfl(mine, 0, 64); nx = __shfl(mine, 1, 64); qx = __shfl(mine, 2, 64); ny = __shfl(mine, 3, 64); qy = __shfl(mine, 4, 64);
		} else {
			v = c[pix];
			nx = bc_tap(c, w, h, px - 1, py); qx = bc_tap(c, w, h, px + 1, py);
			ny = bc_tap(c, w, h, px, py - 1); qy = bc_tap(c, w, h, px, py + 1);
		}
		uint32_t color[3];
		const float score = disc_score_wave(f, w, h, px, py, radius, v, color);
		if(lane != 0) continue;
		if(score < min_score) { // :79-82 (NaN is kept)
			atomicAdd(&counter[3 * z + 1], 1);
			continue;
		}
		const int i = atomicAdd(&counter[3 * z], 1); // :87
		if(i >= max_matches) continue;               // :88-89
		const float mx = (float)px + 0.5f * (nx - qx) / (nx - 2 * v + qx); // :93
		const float my = (float)py + 0.5f * (ny - qy) / (ny - 2 * v + qy); // :94
		store_match(matches + ((size_t)z * max_matches + i) * 22, mx, my, color, f[pix], v, score);
		idx_out[(size_t)z * max_matches + i] = pix;
	}
}

// RASTER order: rank of each record = number of records with a smaller pixel index (indices are unique)
__global__ __launch_bounds__(256) void k_rank_scatter(const uint8_t* __restrict__ rec, const uint32_t* __restrict__ idx, const int32_t* __restrict__ counter,
		uint8_t* __restrict__ rec_tmp, uint32_t* __restrict__ idx_tmp, int max_matches) {
	__shared__ uint32_t tile[256];
	const int z = blockIdx.y;
	const int n = min(counter[3 * z], max_matches);
	const int t = blockIdx.x * 256 + threadIdx.x;
	const uint32_t* id = idx + (size_t)z * max_matches;
	const uint32_t mine = t < n ? id[t] : 0;
	int rank = 0;
	for(int j0 = 0; j0 < n; j0 += 256) {
		const int j = j0 + threadIdx.x;
		tile[threadIdx.x] = j < n ? id[j] : 0xFFFFFFFFu;
		__syncthreads();
		const int m = min(256, n - j0);
		for(int j2 = 0; j2 < m; j2++) rank += tile[j2] < mine;
		__syncthreads();
	}
	if(t < n) {
		const uint16_t* src = reinterpret_cast<const uint16_t*>(rec + ((size_t)z * max_matches + t) * 22);
		uint16_t* dst = reinterpret_cast<uint16_t*>(rec_tmp + ((size_t)z * max_matches + rank) * 22);
#pragma unroll
		for(int k = 0; k < 11; k++) dst[k] = src[k];
		if(idx_tmp) idx_tmp[(size_t)z * max_matches + rank] = mine;
	}
}

__global__ __launch_bounds__(256) void k_zero2(int32_t* __restrict__ a, int na, int32_t* __restrict__ b, int nb) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if(i < na) a[i] = 0;
	if(i < nb) b[i] = 0;
}

// circle + detect fused: the SAT tile (+ r+1 halo) is staged in LDS with clamp-to-edge addressing (satBlobCenter.cl:20, H7),
// blobCenter is computed for the 64x64 tile plus a 1-pixel ring into LDS, then thresholded / peak-tested (blobList.cl:37-55).
// Out-of-image ring positions take the value of the clamped position, exactly what the clamped reads of the reference return.
constexpr int FT_W = 64, FT_H = 64, FT_NT = 512; // tile and threads per workgroup (64x64: region / tile = 1.48, 64x32 was 1.72)

// NE: region elements per thread held in registers so that ALL global loads are in flight before the first LDS store
// (a plain load->store loop with run-time bounds is not unrolled and pays one memory round trip per iteration); NE = 0
// selects that plain loop for very large radii.
// RT: the circle radius as a compile-time constant (0 = run-time radius).  The kernel is VALU-bound on index arithmetic
// (measured: SQ_ACTIVE_INST_VALU == kernel time); with RT fixed every tap offset, the region pitch and the divisions by it
// fold into immediates.
template<int NE, int RT>
__global__ __launch_bounds__(FT_NT) void k_circle_detect(const int32_t* __restrict__ sat_all, float* __restrict__ bc_out, uint32_t* __restrict__ cand,
		int32_t* __restrict__ cand_count, int32_t* __restrict__ counter, int w, int h, int r_arg, float thr) {
	extern __shared__ __attribute__((aligned(16))) char smem_tail[];
	const int r = RT > 0 ? RT : r_arg;
	const int halo = r + 1;
	// horizontal halo rounded up to whole 16-byte groups when the radius is a compile-time constant: interior tiles are then
	// fetched with aligned dwordx4 loads and no clamping (the two paths fill the same LDS layout)
	const int HA = RT > 0 ? ((RT + 1 + 3) & ~3) : halo, XO = HA - halo;
	const int RW = FT_W + 2 * HA, RH = FT_H + 2 * halo;
	uint32_t* region = reinterpret_cast<uint32_t*>(smem_tail); // region[ry][rx] = SAT[clamp(y0 - halo + ry)][clamp(x0 - HA + rx)]
	// 1-D grid, XCD-aware: each XCD walks a contiguous run of tiles (row-major inside a frame), so the halo lines neighbouring
	// tiles share are L2 hits instead of one HBM fetch per XCD
	const unsigned tiles_x = (unsigned)(w + FT_W - 1) / FT_W, tiles_y = (unsigned)(h + FT_H - 1) / FT_H;
	const unsigned lb = vp_xcd_remap(blockIdx.x, gridDim.x);
	const int z = (int)(lb / (tiles_x * tiles_y));
	const unsigned trem = lb - (unsigned)z * tiles_x * tiles_y;
	const size_t n = (size_t)w * h;
	const int32_t* sat = sat_all + z * n;
	const int x0 = (int)(trem % tiles_x) * FT_W, y0 = (int)(trem / tiles_x) * FT_H;
	const int tid = threadIdx.x;
	const bool vec_ok = RT > 0 && (w & 3) == 0 && w >= 4 && (reinterpret_cast<uintptr_t>(sat_all) & 15) == 0;
	const bool interior = vec_ok && x0 - HA >= 0 && x0 + FT_W + HA <= w && y0 - halo >= 0 && y0 + FT_H + halo <= h;
	if(vec_ok) { // workgroup-uniform.  Border tiles too: rows clamp by index; a 4-pixel group is either wholly inside the image or wholly
	             // outside it (w, x0 and HA are multiples of 4), and an outside group is the edge column replicated (clamp-to-edge)
		constexpr int RW4 = RT > 0 ? (FT_W + 2 * ((RT + 1 + 3) & ~3)) / 4 : 1, RHc = RT > 0 ? FT_H + 2 * (RT + 1) : 1, NV = (RW4 * RHc + FT_NT - 1) / FT_NT;
		int4 v4[NV];
#pragma unroll
		for(int k = 0; k < NV; k++) {
			const int i = min(tid + FT_NT * k, RW4 * RHc - 1), row = i / RW4, c4 = i - row * RW4;
			const int gy = interior ? y0 - halo + row : vp_clampi(y0 - halo + row, 0, h - 1), gx = x0 - HA + 4 * c4;
			typedef int v4i_ __attribute__((ext_vector_type(4)));
			const v4i_ vv = __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(sat + (size_t)gy * w + (interior ? gx : vp_clampi(gx, 0, w - 4))));
			const int4 v = make_int4(vv.x, vv.y, vv.z, vv.w);
			v4[k] = interior || (gx >= 0 && gx < w) ? v : (gx < 0 ? make_int4(v.x, v.x, v.x, v.x) : make_int4(v.w, v.w, v.w, v.w));
		}
#pragma unroll
		for(int k = 0; k < NV; k++)
			if(tid + FT_NT * k < RW4 * RHc) reinterpret_cast<int4*>(region)[tid + FT_NT * k] = v4[k];
	} else if(NE > 0) {
		uint32_t v[NE > 0 ? NE : 1];
		const int total = RW * RH, dq = FT_NT / RW, dr = FT_NT - dq * RW;
		int ry = tid / RW, rx = tid - ry * RW;
#pragma unroll
		for(int k = 0; k < NE; k++) {
			v[k] = (k * FT_NT + tid < total) ? (uint32_t)sat[(size_t)vp_clampi(y0 - halo + ry, 0, h - 1) * w + vp_clampi(x0 - HA + rx, 0, w - 1)] : 0u;
			ry += dq;
			rx += dr;
			if(rx >= RW) {
				rx -= RW;
				ry++;
			}
		}
#pragma unroll
		for(int k = 0; k < NE; k++)
			if(k * FT_NT + tid < total) region[k * FT_NT + tid] = v[k];
	} else {
		for(int ry = tid >> 6; ry < RH; ry += FT_NT / 64) {
			const int gy = vp_clampi(y0 - halo + ry, 0, h - 1);
			for(int rx = tid & 63; rx < RW; rx += 64) region[ry * RW + rx] = (uint32_t)sat[(size_t)gy * w + vp_clampi(x0 - HA + rx, 0, w - 1)];
		}
	}
	__syncthreads();
	// All four quadrant sums of satBlobCenter.cl:37-40 are ONE box-sum image sampled at four offsets:
	//   Bx(u, v) = S(u+r-1, v+r-1) - S(u+r-1, v) - S(u, v+r-1) + S(u, v)        (every tap clamped on its own, as in the reference)
	//   pp(x, y) = Bx(x+1, y+1),  nn = Bx(x-r, y-r),  pn = -Bx(x+1, y-r),  np = -Bx(x-r, y+1)
	// so Bx is built once (4 LDS taps per entry) and blobCenter needs 4 reads instead of 16.  Bx covers u in
	// [x0-1-r, x0+FT_W+1], v likewise (its origin coincides with the region's); wrapping int32 arithmetic, exact.
	const int BXW = FT_W + r + 3, BXH = FT_H + r + 3;
	int32_t* bx = reinterpret_cast<int32_t*>(region + RW * RH);
	// A wavefront that straddled two rows of the 72-wide image would touch a span of more than 64 dwords: 2-way bank conflicts on
	// every tap (measured: SQ_LDS_BANK_CONFLICT = 54 % of the LDS cycles).  So: columns 0..63 with one row per wavefront, then the
	// few remaining columns.
	auto bx_entry = [&](int by, int bxx) {
		const uint32_t* c = region + by * RW + bxx + XO;
		bx[by * BXW + bxx] = (int32_t)(c[(r - 1) * RW + (r - 1)] - c[r - 1] - c[(r - 1) * RW] + c[0]);
	};
	// Odd compile-time radius, interior tile: two horizontally adjacent entries per thread through 8-byte LDS accesses (ds_read_b64 /
	// ds_write_b64 move 256 B per clock, the 4-byte forms 128 B).  With r odd every tap of an even-column pair is 8-byte aligned:
	// RW, BXW, BW and XO are even and the blobCenter taps sit at odd offsets (+1, -r) from an odd column (bxx + r).
	constexpr bool PAIRS = RT > 0 && (RT & 1) == 1;
	const bool pairs = PAIRS && interior; // workgroup-uniform; blobCenter clamps its centre on border tiles, the box sums never clamp
	if(PAIRS && vec_ok) {
		auto bx_pair = [&](int by, int bxx) {
			const uint32_t* c = region + by * RW + bxx + XO;
			const uint2 t11 = *reinterpret_cast<const uint2*>(c + (r - 1) * RW + (r - 1)), t10 = *reinterpret_cast<const uint2*>(c + (r - 1));
			const uint2 t01 = *reinterpret_cast<const uint2*>(c + (r - 1) * RW), t00 = *reinterpret_cast<const uint2*>(c);
			*reinterpret_cast<int2*>(bx + by * BXW + bxx) = make_int2((int32_t)(t11.x - t10.x - t01.x + t00.x), (int32_t)(t11.y - t10.y - t01.y + t00.y));
		};
		for(int e = tid; e < 32 * BXH; e += FT_NT) bx_pair(e >> 5, 2 * (e & 31)); // one row per 32-lane half (the bank-conflict group of b64)
		const int xp = (BXW - 64) / 2;
		for(int e = tid; e < xp * BXH; e += FT_NT) {
			const int by = e / xp;
			bx_pair(by, 64 + 2 * (e - by * xp));
		}
	} else {
		for(int e = tid; e < 64 * BXH; e += FT_NT) bx_entry(e >> 6, e & 63);
		const int extra = BXW - 64; // r + 3 columns
		for(int e = tid; e < extra * BXH; e += FT_NT) {
			const int by = e / extra;
			bx_entry(by, 64 + (e - by * extra));
		}
	}
	__syncthreads();
	const float div = (float)(r * r), rcp = 1.0f / div;
	constexpr int BW = FT_W + 2, BH = FT_H + 2;
	float* bc = reinterpret_cast<float*>(region); // the blobCenter ring tile aliases the (now dead) SAT region
	auto bc_entry = [&](int by, int bxx) {
		const int e = by * BW + bxx;
		// clamped centre, in Bx coordinates (origin x0-1-r, y0-1-r)
		const int cx = vp_clampi(x0 - 1 + bxx, 0, w - 1) - (x0 - 1 - r), cy = vp_clampi(y0 - 1 + by, 0, h - 1) - (y0 - 1 - r);
		const int32_t* c = bx + cy * BXW + cx;
		// min(min(pp, nn), min(pn, np)) of satBlobCenter.cl:41 taken on the exact int32 box sums and converted once: int -> float
		// conversion is monotonic, so the converted minimum equals the minimum of the converted values
		const float m = (float)min(min(c[BXW + 1], c[-r * BXW - r]), min(-c[-r * BXW + 1], -c[BXW - r]));
		if(RT >= 3 && RT <= 10) {
			// m / (r*r), correctly rounded, without the division sequence: q0 = m*rcp, e = fma(-d, q0, m), q = fma(e, rcp, q0).
			// Exhaustively verified equal to m / d for every float m = (float)n, |n| <= 2^27, d = 9 .. 100 (tools/divcheck.hip);
			// the box sums here are bounded by 100 * 3 * 255^2 < 2^25.
			const float q0 = m * rcp;
			bc[e] = __builtin_fmaf(__builtin_fmaf(-div, q0, m), rcp, q0);
		} else bc[e] = m / div;
	};
	if(pairs) { // interior: no clamping, cx = bxx + r, cy = by + r
		auto quot = [&](float m) {
			const float q0 = m * rcp;
			return __builtin_fmaf(__builtin_fmaf(-div, q0, m), rcp, q0); // PAIRS implies 3 <= RT <= 9: see bc_entry
		};
		auto bc_pair = [&](int by, int bxx) {
			const int32_t* c = bx + (by + r) * BXW + bxx + r;
			const int2 ppi = *reinterpret_cast<const int2*>(c + BXW + 1), nni = *reinterpret_cast<const int2*>(c - r * BXW - r);
			const int2 pni = *reinterpret_cast<const int2*>(c - r * BXW + 1), npi = *reinterpret_cast<const int2*>(c + BXW - r);
			float2 o;
			o.x = quot((float)min(min(ppi.x, nni.x), min(-pni.x, -npi.x))); // integer minimum, one conversion (see bc_entry)
			o.y = quot((float)min(min(ppi.y, nni.y), min(-pni.y, -npi.y)));
			*reinterpret_cast<float2*>(bc + by * BW + bxx) = o;
		};
		static_assert(!PAIRS || (RT >= 3 && RT <= 10), "the fma quotient is verified for r = 3 .. 10 only");
		for(int e = tid; e < 32 * BH; e += FT_NT) bc_pair(e >> 5, 2 * (e & 31));
		for(int e = tid; e < BH; e += FT_NT) bc_pair(e, 64); // the ring columns 64, 65
	} else {
		for(int e = tid; e < 64 * BH; e += FT_NT) bc_entry(e >> 6, e & 63); // one row per wavefront (see above), then the two ring columns
		for(int e = tid; e < 2 * BH; e += FT_NT) bc_entry(e >> 1, 64 + (e & 1));
	}
	__syncthreads();
	for(int e = tid; e < FT_W * FT_H; e += FT_NT) {
		const int ty = e / FT_W, tx = e - ty * FT_W;
		const int x = x0 + tx, y = y0 + ty;
		if(x >= w || y >= h) continue;
		const float* p = bc + (ty + 1) * BW + tx + 1;
		const float v = p[0];
		if(bc_out) bc_out[z * n + (size_t)y * w + x] = v;
		if(v < thr) continue;
		if(p[-1] > v || p[1] > v || p[-BW] > v || p[BW] > v) {
			atomicAdd(&counter[3 * z + 2], 1);
			continue;
		}
		const int slot = atomicAdd(&cand_count[z], 1);
		cand[z * n + slot] = (uint32_t)(y * w + x);
	}
}

// blobcenter != nullptr: list from a materialised blobCenter image (vp_bloblist).  Otherwise fused from the SAT
// (vp_blobcenter_list): blobCenter lives in LDS only, `bc_out` (optional) receives the debug view.
static int list_from_blobcenter(VpContext* ctx, int stream, const uint8_t* flat, const float* blobcenter, const int32_t* sat, int circle_radius, float* bc_out,
		int w, int h, int batch, float thr, float min_score, int radius, int max_matches, int order, VpMatch* matches, uint32_t* idx, int32_t* counter, const ListScratch& ws) {
	hipStream_t s = ctx->streams[stream];
	// main.cpp:283-288: the host zeroes counter[3] before every launch; here one small kernel clears it and the candidate counters
	// (two hipMemsetAsync calls cost ~10 us each)
	k_zero2<<<vp_div_up(3 * batch, 256), 256, 0, s>>>(counter, 3 * batch, ws.cand_count, batch);
	VP_TRY(vp_launch_ok("blobList counters"));
	uint32_t* idx_work = (order == VP_ORDER_RASTER || !idx) ? ws.idx : idx;
	// RASTER order: the emit kernels append into workspace records, k_rank_scatter sorts them straight into the caller's arrays
	uint8_t* rec_work = order == VP_ORDER_RASTER ? ws.rec_tmp : reinterpret_cast<uint8_t*>(matches);
	if(blobcenter) {
		k_detect<<<dim3(vp_div_up(w, 64), vp_div_up(h, 4), batch), dim3(64, 4), 0, s>>>(blobcenter, ws.cand, ws.cand_count, counter, w, h, thr);
		VP_TRY(vp_launch_ok("blobList detect"));
		k_emit<false><<<dim3(64, batch), 256, 0, s>>>(reinterpret_cast<const uint32_t*>(flat), blobcenter, ws.cand, ws.cand_count, counter, rec_work, idx_work,
				w, h, min_score, radius, max_matches, 0);
	} else {
		const int halo = circle_radius + 1;
		const int ha = (circle_radius >= 3 && circle_radius <= 10) ? ((halo + 3) & ~3) : halo; // k_circle_detect: aligned horizontal halo of the specialised radii
		const size_t lds = sizeof(uint32_t) * ((size_t)(FT_W + 2 * ha) * (FT_H + 2 * halo) + (size_t)(FT_W + circle_radius + 3) * (FT_H + circle_radius + 3));
		VP_REQUIRE(lds <= 160 * 1024, "vp_blobcenter_list: circle radius %d too large for the LDS tile", circle_radius);
		const int per_thread = ((FT_W + 2 * halo) * (FT_H + 2 * halo) + FT_NT - 1) / FT_NT;
		VP_REQUIRE((long long)vp_div_up(w, FT_W) * vp_div_up(h, FT_H) * batch < (1ll << 31), "vp_blobcenter_list: too many tiles");
		const dim3 grid((unsigned)(vp_div_up(w, FT_W) * vp_div_up(h, FT_H) * batch));
#define VP_CD(N, R) \
		do { \
			if(lds > 48 * 1024) VP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_circle_detect<N, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
			k_circle_detect<N, R><<<grid, FT_NT, lds, s>>>(sat, bc_out, ws.cand, ws.cand_count, counter, w, h, circle_radius, thr); \
		} while(0)
#define VP_CD_R(R) case R: VP_CD(((FT_W + 2 * ((R + 1 + 3) & ~3)) * (FT_H + 2 * (R + 1)) + FT_NT - 1) / FT_NT, R); break
		switch(circle_radius) { // the radii real geometries produce (ceil(20 mm / fieldScale)): compile-time specialisations
			VP_CD_R(3); VP_CD_R(4); VP_CD_R(5); VP_CD_R(6); VP_CD_R(7); VP_CD_R(8); VP_CD_R(9); VP_CD_R(10);
		default:
			if(per_thread <= 14) VP_CD(14, 0);
			else if(per_thread <= 20) VP_CD(20, 0);
			else if(per_thread <= 32) VP_CD(32, 0);
			else VP_CD(0, 0);
		}
#undef VP_CD_R
#undef VP_CD
		VP_TRY(vp_launch_ok("circle+detect"));
		k_emit<true><<<dim3(64, batch), 256, 0, s>>>(reinterpret_cast<const uint32_t*>(flat), sat, ws.cand, ws.cand_count, counter, rec_work, idx_work,
				w, h, min_score, radius, max_matches, circle_radius);
	}
	VP_TRY(vp_launch_ok("blobList emit"));
	if(order == VP_ORDER_RASTER) {
		const dim3 grid(vp_div_up(max_matches, 256), batch);
		k_rank_scatter<<<grid, 256, 0, s>>>(ws.rec_tmp, ws.idx, counter, reinterpret_cast<uint8_t*>(matches), idx, max_matches);
		VP_TRY(vp_launch_ok("blobList rank"));
	}
	return VP_OK;
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
	ListScratch ws;
	VP_TRY(list_scratch(ctx, stream, (size_t)w * h, batch, max_matches, false, &ws));
	VpStageTimer t(ctx, stream, "blobList");
	return list_from_blobcenter(ctx, stream, flat, blobcenter, nullptr, 0, nullptr, w, h, batch, circ_threshold, min_score, radius, max_matches, order, matches, idx, counter, ws);
}

extern "C" int vp_blobcenter_list(VpContext* ctx, int stream, const uint8_t* flat, const int32_t* sat, int w, int h, int batch,
		int circle_radius, float circ_threshold, float min_score, int list_radius, int max_matches, int order,
		VpMatch* matches, uint32_t* idx, int32_t* counter, float* blobcenter_out) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_TRY(list_args_ok("vp_blobcenter_list", flat, sat, w, h, batch, list_radius, max_matches, order, matches, counter));
	VP_REQUIRE(circle_radius >= 0, "vp_blobcenter_list: bad circle radius");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	ListScratch ws;
	VP_TRY(list_scratch(ctx, stream, (size_t)w * h, batch, max_matches, false, &ws));
	VpStageTimer t(ctx, stream, "satBlobCenter+blobList");
	return list_from_blobcenter(ctx, stream, flat, nullptr, sat, circle_radius, blobcenter_out, w, h, batch, circ_threshold, min_score, list_radius, max_matches, order, matches, idx, counter, ws);
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
