// Single-pass LDS-tiled summed-area table with decoupled look-back of tile carries (VP_SAT_TILED).
//
// Replaces gradientDot.cl + satHorizontal.cl + satVertical.cl (Resources.cpp:160-162) with ONE launch that reads the
// flat image once and writes the int32 SAT once: 8 B/px of HBM traffic (plus the stencil halo, mostly L2 hits, and
// ~1 KB of carry granules per 64x64 tile).
//
// Per 64x64 tile, one 256-thread workgroup (4 wavefronts):
//   A  stage the RGBA tile + halo `o` in LDS with clamp-to-edge addressing (gradientDot.cl:20)
//   B  each wavefront owns 16 rows: packed-16-bit gradient dot product from LDS taps, DPP inclusive row scan across the
//      64 lanes, running column sum kept in registers => local SAT of the 16-row segment, written to LDS
//   C  local aggregates: bottom row PC(x) and right column PR(y) of the tile-local SAT
//   D  publish PC / PR as "aggregate" carry granules; look back along the tile column (wave 0), the tile row (wave 1)
//      and the corner chain (wave 2) until an "inclusive" granule is met; publish our inclusive granules
//   E  out(x,y) = local(x,y) + A(x) + B(y) + K, 16-byte coalesced stores
// with  A(x) = sum of PC over tiles above,  B(y) = sum of PR over tiles to the left,  K = SAT at the tile's upper-left
// corner = sum over tiles above of their row-carry totals.
//
// Carry granules follow the gfx950 hand-off rules of cdna_hip_programming.md, Guideline 16 (form R2): one naturally
// aligned 8-byte {tag, value} word per element, written by ONE agent-scope relaxed store (global_store_dwordx2 sc1) and
// polled by agent-scope relaxed loads (global_load_dwordx2 sc1): the data is the flag, no fence, no L1 staleness.
// tag = (epoch << 2) | state (1 aggregate, 2 inclusive); `epoch` is bumped per launch so stale granules never match.
// Forward progress does not depend on dispatch order: tiles are claimed in raster order from a ticket counter, so a
// workgroup only ever waits on tiles whose workgroups have already started.  Every spin is bounded; on expiry the kernel
// raises a device abort word and a pinned host flag, finishes with garbage and the host reports VP_ERR_TIMEOUT.
#include "vp_common.h"

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;
typedef short short2v __attribute__((ext_vector_type(2)));

namespace {

constexpr int TW = 64, TH = 64, NWAVES = 4, SEG = TH / NWAVES;
constexpr unsigned SPIN_LIMIT = 1u << 22; // polls (each with s_sleep): seconds, never reached in a healthy run
constexpr uint32_t ST_AGG = 1, ST_INC = 2;

struct TiledArgs {
	int w, h, o;
	int tilesX, tilesY, nTiles;
	int pitch, off; // LDS raw-tile row pitch (dwords) and column of the tile interior
	uint32_t ticket_base, epoch;
	u64 *V, *H, *K; // carry granules: [tile][TW], [tile][TH], [tile]
	uint32_t* ctrl; // [0] ticket counter (monotonic across launches), [1] abort word
	int* host_flag; // pinned, device-mapped
};

__device__ __forceinline__ u64 ld_granule(const u64* p) {
	return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(u64* p, uint32_t epoch, uint32_t state, int value) {
	__hip_atomic_store((gu64*)p, ((u64)((epoch << 2) | state) << 32) | (uint32_t)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

// sum over channels of (xp - xn) * (yp - yn) on RGBA8 taps: bytes widened to 16-bit pairs (R,B) / (G,A), packed
// subtraction, 2-way integer dot products.  Alpha is 255 on every tap, its difference is 0.  (gradientDot.cl:25-29)
__device__ __forceinline__ int grad_dot_pk(uint32_t xp, uint32_t xn, uint32_t yp, uint32_t yn) {
	const uint32_t m = 0x00FF00FFu;
	const uint32_t gxl = __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, xp & m) - __builtin_bit_cast(short2v, xn & m));
	const uint32_t gxh = __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, (xp >> 8) & m) - __builtin_bit_cast(short2v, (xn >> 8) & m));
	const uint32_t gyl = __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, yp & m) - __builtin_bit_cast(short2v, yn & m));
	const uint32_t gyh = __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, (yp >> 8) & m) - __builtin_bit_cast(short2v, (yn >> 8) & m));
	int acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, gxl), __builtin_bit_cast(short2v, gyl), 0, false);
	acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, gxh), __builtin_bit_cast(short2v, gyh), acc, false);
	return acc;
}

__device__ __forceinline__ void raise_timeout(const TiledArgs& a) {
	__hip_atomic_store((gu32*)(a.ctrl + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(a.host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool aborted(const TiledArgs& a) {
	return __hip_atomic_load((gu32*)(a.ctrl + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// Exclusive carry of this lane's element along one chain: walk the predecessors (nearest first), adding aggregates
// until an inclusive granule is met.  `g` points at THIS tile's granule for this lane, predecessors are `stride`
// granules apart.  Wave-uniform control flow; returns false on timeout / abort.
__device__ __forceinline__ bool lookback(const u64* g, long stride, int npred, const TiledArgs& a, int& carry) {
	int acc = 0;
	bool done = false;
	for(int k = 1; k <= npred; k++) {
		const u64* p = g - (long)k * stride;
		u64 v;
		uint32_t tag;
		for(unsigned spins = 0;; spins++) {
			v = ld_granule(p);
			tag = (uint32_t)(v >> 32);
			const bool ready = done || ((tag >> 2) == a.epoch && (tag & 3u) != 0);
			if(__all(ready)) break;
			if(spins > SPIN_LIMIT || ((spins & 63u) == 63u && aborted(a))) {
				if((threadIdx.x & 63) == 0) raise_timeout(a);
				carry = 0;
				return false;
			}
			__builtin_amdgcn_s_sleep(1);
		}
		if(!done) {
			acc += (int)(uint32_t)v;
			done = (tag & 3u) == ST_INC;
		}
		if(__all(done)) break;
	}
	carry = acc;
	return true;
}

template<bool FUSED>
__global__ __launch_bounds__(256) void k_sat_tiled(const void* __restrict__ in, int32_t* __restrict__ sat, float* __restrict__ gd_out, const TiledArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int rawRows = FUSED ? TH + 2 * a.o : TH;
	uint32_t* raw = reinterpret_cast<uint32_t*>(smem);
	int* S = reinterpret_cast<int*>(raw + rawRows * a.pitch);
	int* T = S + TH * TW;
	int* Av = T + NWAVES * TW;
	int* Bv = Av + TW;
	int* misc = Bv + TH; // [0] ticket, [1] K, [2] Q

	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if(tid == 0) misc[0] = (int)(atomicAdd(a.ctrl, 1u) - a.ticket_base);
	__syncthreads();
	const int ticket = misc[0];
	if(ticket < 0 || ticket >= a.nTiles) return; // cannot happen unless the host lost count; never index out of range
	const int perFrame = a.tilesX * a.tilesY;
	const int f = ticket / perFrame, rem = ticket - f * perFrame;
	const int ti = rem / a.tilesX, tj = rem - ti * a.tilesX;
	const int x0 = tj * TW, y0 = ti * TH;
	const int w = a.w, h = a.h, o = a.o;
	const size_t frame = (size_t)f * w * h;

	// ---- A: stage the tile (+ halo) in LDS ------------------------------------------------------------------------
	if(FUSED) {
		const uint32_t* flat = reinterpret_cast<const uint32_t*>(in) + frame;
		const bool vec = (x0 + TW <= w) && ((w & 3) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
		if(vec) {
			const int c4 = tid & 15;
			for(int r = tid >> 4; r < rawRows; r += 16) {
				const int gy = vp_clampi(y0 - o + r, 0, h - 1);
				const uint4 v = *reinterpret_cast<const uint4*>(flat + (size_t)gy * w + x0 + 4 * c4);
				*reinterpret_cast<uint4*>(raw + r * a.pitch + a.off + 4 * c4) = v;
			}
			if(tid < rawRows) { // left and right halo columns, one row per thread
				const int gy = vp_clampi(y0 - o + tid, 0, h - 1);
				for(int k = 0; k < o; k++) {
					raw[tid * a.pitch + a.off - o + k] = flat[(size_t)gy * w + vp_clampi(x0 - o + k, 0, w - 1)];
					raw[tid * a.pitch + a.off + TW + k] = flat[(size_t)gy * w + vp_clampi(x0 + TW + k, 0, w - 1)];
				}
			}
		} else {
			const int rawCols = TW + 2 * o;
			for(int r = wv; r < rawRows; r += NWAVES) {
				const int gy = vp_clampi(y0 - o + r, 0, h - 1);
				for(int c = lane; c < rawCols; c += 64) raw[r * a.pitch + a.off - o + c] = flat[(size_t)gy * w + vp_clampi(x0 - o + c, 0, w - 1)];
			}
		}
	} else {
		const float* gin = reinterpret_cast<const float*>(in) + frame;
		for(int r = wv; r < TH; r += NWAVES) {
			const int gy = y0 + r, gx = x0 + lane;
			raw[r * a.pitch + lane] = (gy < h && gx < w) ? (uint32_t)(int)gin[(size_t)gy * w + gx] : 0u;
		}
	}
	__syncthreads();

	// ---- B: gradient dot product, row scan (lanes), running column sum (registers) ----------------------------------
	{
		int run = 0;
		const bool xin = x0 + lane < w;
#pragma unroll 4
		for(int rr = 0; rr < SEG; rr++) {
			const int r = wv * SEG + rr;
			int g;
			if(FUSED) {
				const uint32_t* c = raw + (r + o) * a.pitch + a.off + lane;
				g = grad_dot_pk(c[o], c[-o], c[o * a.pitch], c[-o * a.pitch]);
				if(!(xin && y0 + r < h)) g = 0;
				if(gd_out && xin && y0 + r < h) gd_out[frame + (size_t)(y0 + r) * w + x0 + lane] = (float)g;
			} else {
				g = (int)raw[r * a.pitch + lane];
			}
			run += scan_incl_dpp(g);
			S[r * TW + lane] = run;
		}
		T[wv * TW + lane] = run;
	}
	__syncthreads();

	// ---- C/D: aggregates, publication, look-back -----------------------------------------------------------------
	const size_t tile = (size_t)ticket;
	bool ok = true;
	if(wv == 0) { // column chain: A(x)
		const int pc = T[lane] + T[TW + lane] + T[2 * TW + lane] + T[3 * TW + lane]; // bottom row of the local SAT
		u64* g = a.V + tile * TW + lane;
		int carry = 0;
		if(ti > 0) {
			st_granule(g, a.epoch, ST_AGG, pc);
			ok = lookback(g, (long)a.tilesX * TW, ti, a, carry);
		}
		st_granule(g, a.epoch, ST_INC, carry + pc);
		Av[lane] = carry;
	} else if(wv == 1) { // row chain: B(y), then the corner chain's aggregate Q = B(TH-1)
		int pr = S[lane * TW + TW - 1]; // right column of the local SAT
		const int seg = lane / SEG;
		for(int s = 0; s < seg; s++) pr += T[s * TW + TW - 1];
		u64* g = a.H + tile * TH + lane;
		int carry = 0;
		if(tj > 0) {
			st_granule(g, a.epoch, ST_AGG, pr);
			ok = lookback(g, TH, tj, a, carry);
		}
		st_granule(g, a.epoch, ST_INC, carry + pr);
		Bv[lane] = carry;
		if(lane == TH - 1) {
			misc[2] = carry; // Q: everything left of this tile within its tile row
			st_granule(a.K + tile, a.epoch, ti > 0 ? ST_AGG : ST_INC, carry);
		}
	} else if(wv == 2) { // corner chain: K = sum of Q over the tiles above (all lanes poll the same word)
		int carry = 0;
		if(ti > 0) ok = lookback(a.K + tile, a.tilesX, ti, a, carry);
		if(lane == 0) misc[1] = carry;
	}
	__syncthreads();
	const int K = misc[1];
	if(wv == 1 && lane == 0 && ti > 0) st_granule(a.K + tile, a.epoch, ST_INC, K + misc[2]);
	(void)ok;

	// ---- E: out = local + A(x) + B(y) + K -----------------------------------------------------------------------------
	{
		const int cg = tid & 15, rr = tid >> 4;
		const int4 av = *reinterpret_cast<const int4*>(Av + 4 * cg);
		int4 base = make_int4(av.x + K, av.y + K, av.z + K, av.w + K);
		int32_t* out = sat + frame;
		const int gx = x0 + 4 * cg;
		const bool vec = (gx + 3 < w) && ((w & 3) == 0) && ((reinterpret_cast<uintptr_t>(sat) & 15) == 0);
#pragma unroll
		for(int k = 0; k < NWAVES; k++) {
			const int r = k * SEG + rr, gy = y0 + r;
			const int4 s = *reinterpret_cast<const int4*>(S + r * TW + 4 * cg);
			const int b = Bv[r];
			const int4 v = make_int4(s.x + base.x + b, s.y + base.y + b, s.z + base.z + b, s.w + base.w + b);
			if(gy < h) {
				if(vec) *reinterpret_cast<int4*>(out + (size_t)gy * w + gx) = v;
				else {
					if(gx < w) out[(size_t)gy * w + gx] = v.x;
					if(gx + 1 < w) out[(size_t)gy * w + gx + 1] = v.y;
					if(gx + 2 < w) out[(size_t)gy * w + gx + 2] = v.z;
					if(gx + 3 < w) out[(size_t)gy * w + gx + 3] = v.w;
				}
			}
			const int4 t = *reinterpret_cast<const int4*>(T + k * TW + 4 * cg);
			base = make_int4(base.x + t.x, base.y + t.y, base.z + t.z, base.w + t.w);
		}
	}
}

} // namespace

int vp_check_timeout(VpContext* ctx, int stream);

int vp_sat_tiled(VpContext* ctx, int stream, const void* in, bool fused, int w, int h, int offset, int batch, int32_t* sat, float* graddot_out) {
	hipStream_t s = ctx->streams[stream];
	const int o = fused ? offset : 0;
	VP_REQUIRE(o <= 32, "VP_SAT_TILED: gradient offset %d > 32 is not supported (LDS halo)", o);
	TiledArgs a;
	a.w = w;
	a.h = h;
	a.o = o;
	a.tilesX = (int)vp_div_up(w, TW);
	a.tilesY = (int)vp_div_up(h, TH);
	const long long nt = (long long)a.tilesX * a.tilesY * batch;
	VP_REQUIRE(nt < (1ll << 30), "VP_SAT_TILED: too many tiles (%lld)", nt);
	a.nTiles = (int)nt;
	const int halo = (o + 3) & ~3;
	a.off = fused ? halo : 0;
	a.pitch = fused ? halo + TW + halo : TW;
	const int rawRows = fused ? TH + 2 * o : TH;
	const size_t lds = sizeof(int) * ((size_t)rawRows * a.pitch + TH * TW + NWAVES * TW + TW + TH + 8);

	// persistent carry state (slot 1): [control 256 B][V][H][K]
	const size_t bV = sizeof(u64) * (size_t)a.nTiles * TW, bH = sizeof(u64) * (size_t)a.nTiles * TH, bK = sizeof(u64) * (size_t)a.nTiles;
	void* base = nullptr;
	bool grew = false;
	VP_TRY(vp_workspace(ctx, stream, 1, 256 + bV + bH + bK, &base, &grew));
	if(!ctx->host_flag[stream]) {
		VP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->host_flag[stream]), 64, hipHostMallocMapped));
		*ctx->host_flag[stream] = 0;
	}
	const bool poisoned = *reinterpret_cast<volatile int*>(ctx->host_flag[stream]) != 0; // an earlier launch timed out (reported by vp_sync)
	if(grew || poisoned || ctx->sat_epoch[stream] >= (1u << 30) - 2) {
		if(!grew) {
			VP_CHECK_HIP(hipStreamSynchronize(s));
			VP_CHECK_HIP(hipMemsetAsync(base, 0, ctx->ws[1][stream].bytes, s));
		}
		*ctx->host_flag[stream] = 0;
		ctx->sat_epoch[stream] = 0;
		ctx->sat_ticket_base[stream] = 0;
	}
	a.epoch = ++ctx->sat_epoch[stream];
	a.ticket_base = ctx->sat_ticket_base[stream];
	ctx->sat_ticket_base[stream] += (uint32_t)a.nTiles;
	uint8_t* p = static_cast<uint8_t*>(base);
	a.ctrl = reinterpret_cast<uint32_t*>(p);
	a.V = reinterpret_cast<u64*>(p + 256);
	a.H = reinterpret_cast<u64*>(p + 256 + bV);
	a.K = reinterpret_cast<u64*>(p + 256 + bV + bH);
	int* dflag = nullptr;
	VP_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dflag), ctx->host_flag[stream], 0));
	a.host_flag = dflag;

	if(fused) {
		if(lds > 48 * 1024) VP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sat_tiled<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		k_sat_tiled<true><<<a.nTiles, 256, lds, s>>>(in, sat, graddot_out, a);
	} else {
		k_sat_tiled<false><<<a.nTiles, 256, lds, s>>>(in, sat, nullptr, a);
	}
	return vp_launch_ok("tiled sat");
}
