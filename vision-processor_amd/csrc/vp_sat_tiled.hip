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

#include <cstdlib>
#include <vector>

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
	int tilesX, tilesY, nTiles, batch;
	int pitch, off; // LDS raw-tile row pitch (dwords) and column of the tile interior
	uint32_t ticket_base, epoch;
	u64 *V, *H, *K; // carry granules: [tile][TW], [tile][TH], [tile]
	const uint32_t* order; // ticket -> (ti << 16 | tj): anti-diagonal order of the tile grid (a linear extension of the carry dependencies)
	uint32_t* ctrl; // [0] abort word, [16 + 16 f] ticket counter of frame f (monotonic across launches, one 64-B line each)
	int* host_flag; // pinned, device-mapped
	int debug;      // timing experiments only (VP_SAT_DEBUG env): bit0 = skip look-back waits (WRONG results)
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
	__hip_atomic_store((gu32*)a.ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(a.host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool aborted(const TiledArgs& a) {
	return __hip_atomic_load((gu32*)a.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// Exclusive carry of this lane's element along one chain: walk the predecessors (nearest first), adding aggregates
// until an inclusive granule is met.  `g` points at THIS tile's granule for this lane, predecessors are `stride`
// granules apart.  Wave-uniform control flow; returns false on timeout / abort.
__device__ __forceinline__ bool lookback(const u64* g, long stride, int npred, const TiledArgs& a, int& carry, bool active = true) {
	constexpr int WIN = 4; // predecessor granules kept in flight per round
	int acc = 0;
	bool done = !active; // lanes beyond the vector length ride along
	carry = 0;
	if(a.debug & 1) return true;
	unsigned n_win = 0, n_repoll = 0;
	const bool stats = (a.debug & 4) && (threadIdx.x & 63) == 0;
	for(int k0 = 1; k0 <= npred; k0 += WIN) {
		n_win++;
		u64 v[WIN];
#pragma unroll
		for(int q = 0; q < WIN; q++) v[q] = (k0 + q <= npred) ? ld_granule(g - (long)(k0 + q) * stride) : 0;
#pragma unroll
		for(int q = 0; q < WIN; q++) {
			if(k0 + q > npred) break;
			uint32_t tag = (uint32_t)(v[q] >> 32);
			for(unsigned spins = 0;; spins++) {
				const bool ready = done || ((tag >> 2) == a.epoch && (tag & 3u) != 0);
				if(__all(ready)) break;
				if(spins > SPIN_LIMIT || ((spins & 63u) == 63u && aborted(a))) {
					if((threadIdx.x & 63) == 0) raise_timeout(a);
					return false;
				}
				__builtin_amdgcn_s_sleep(1);
				n_repoll++;
				v[q] = ld_granule(g - (long)(k0 + q) * stride);
				tag = (uint32_t)(v[q] >> 32);
			}
			if(!done) {
				acc += (int)(uint32_t)v[q];
				done = (tag & 3u) == ST_INC;
			}
			if(__all(done)) {
				carry = acc;
				if(stats) { atomicAdd(a.ctrl + 8, n_win); atomicAdd(a.ctrl + 9, n_repoll); atomicAdd(a.ctrl + 10, 1u); atomicAdd(a.ctrl + 11, (unsigned)(k0 + q)); }
				return true;
			}
		}
	}
	carry = acc;
	return true;
}

// Workgroup barrier that orders LDS only.  __syncthreads() would also drain the loader's in-flight LDS-DMA (vmcnt(0)).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// MODE 0: input is a float gradDot image (vp_sat).  MODE -1: fused gradDot, synchronous tile loads (any offset, any
// width).  MODE 1: fused gradDot with the loader wavefront (wave 3) streaming the NEXT tile into LDS by LDS-DMA
// (global_load_lds, no VGPRs) while waves 0-2 look back for the current one; needs w % 4 == 0 and a 16-byte aligned flat.
// Tickets are requested by the loader's lane 0 one tile ahead: neither the atomic nor the loads sit on the critical path.
//
// LDS image of a tile: interior I[rows][64] dwords (rows = TH + 2o, one 256-B row per 16 lanes => a wave instruction of
// 16 B per lane lands 4 rows contiguously, which is what LDS-DMA requires) followed by the halo columns Hh[rows][2o].
template<int MODE>
__global__ __launch_bounds__(256) void k_sat_tiled(const void* __restrict__ in, int32_t* __restrict__ sat, float* __restrict__ gd_out, const TiledArgs a) {
	constexpr bool FUSED = MODE != 0;
	constexpr bool PIPE = MODE > 0;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int o = FUSED ? a.o : 0;
	const int rows = TH + 2 * o;
	uint32_t* I = reinterpret_cast<uint32_t*>(smem);
	const int rowsPad = (rows + 3) & ~3; // DMA pieces are 4 rows: the last one may spill into the slack rows
	uint32_t* Hh = I + rowsPad * TW;
	int* S = reinterpret_cast<int*>(Hh + a.pitch); // a.pitch = halo array size in dwords, padded to a multiple of 4
	int* T = S + TH * TW;
	int* Av = T + NWAVES * TW;
	int* Bv = Av + TW;
	int* misc = Bv + TH; // [0] first ticket, [1] K, [2] Q, [3] next ticket

	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int w = a.w, h = a.h;
	// One ticket queue per frame: workgroup b serves frame b % batch; tiles of a frame are claimed in raster order and a
	// workgroup works through its tickets in increasing order.  (A single counter would serialise the grid at ~88 atomics/us.)
	const int perFrame = a.tilesX * a.tilesY;
	const int f = (int)(blockIdx.x % (unsigned)a.batch);
	uint32_t* queue = a.ctrl + 16 + 16 * f;
	const size_t frame = (size_t)f * w * h;
	const uint32_t* flat = reinterpret_cast<const uint32_t*>(in) + frame;
	const float* gin = reinterpret_cast<const float*>(in) + frame;
	int32_t* out = sat + frame;
	const bool loader = tid == 3 * 64; // lane 0 of wave 3 owns the ticket queue

	uint32_t pending = 0; // loader: ticket requested one tile ahead
	if(loader) {
		const uint32_t t0 = atomicAdd(queue, 1u);
		pending = atomicAdd(queue, 1u);
		misc[0] = (int)(t0 - a.ticket_base);
	}
	__syncthreads();
	int rem = misc[0];
	if(rem < 0 || rem >= perFrame) return; // two tickets taken, none valid: the host accounts for exactly two overshoots

	// LDS-DMA of one full-width tile by one wavefront: ceil(rows/4) 1-KiB pieces + the halo columns, 64 dwords apiece
	auto dma_tile = [&](int x0, int y0) {
		const int c4 = lane & 15, r0 = lane >> 4;
		for(int p = 0; p < rows; p += 4) {
			const int r = min(p + r0, rows - 1); // the last piece may run past the tile: re-fetch the last row into the slack
			const uint32_t* g = flat + (size_t)vp_clampi(y0 - o + r, 0, h - 1) * w + x0 + 4 * c4;
			__builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(I + p * TW), 16, 0, 0);
		}
		const int nh = rows * 2 * o;
		for(int e0 = 0; e0 < nh; e0 += 64) {
			const int e = min(e0 + lane, nh - 1);
			const int r = e / (2 * o), k = e - r * (2 * o);
			const int gx = k < o ? x0 - o + k : x0 + TW + (k - o);
			const uint32_t* g = flat + (size_t)vp_clampi(y0 - o + r, 0, h - 1) * w + vp_clampi(gx, 0, w - 1);
			__builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(Hh + e0), 4, 0, 0);
		}
	};
	bool staged = false; // wave 3: the current tile was (or is being) delivered by LDS-DMA
	if(PIPE && wv == 3) {
		const uint32_t t0 = a.order[rem];
		const int tj0 = (int)(t0 & 0xFFFF), ti0 = (int)(t0 >> 16);
		if(tj0 * TW + TW <= w) {
			dma_tile(tj0 * TW, ti0 * TH);
			staged = true;
		}
	}

	for(;;) {
		const uint32_t tcur = a.order[rem];
		const int ti = (int)(tcur >> 16), tj = (int)(tcur & 0xFFFF);
		const int x0 = tj * TW, y0 = ti * TH;
		const bool full = x0 + TW <= w; // full-width tile: DMA path

		// ---- A: the tile (+ halo) in LDS; publish the next ticket, request the one after --------------------------------
		if(PIPE && full) {
			// delivered by the loader's LDS-DMA; the __syncthreads() below drains it (vmcnt(0)) before anyone reads
		} else if(FUSED) {
			for(int r = wv; r < rows; r += NWAVES) {
				const int gy = vp_clampi(y0 - o + r, 0, h - 1);
				I[r * TW + lane] = flat[(size_t)gy * w + vp_clampi(x0 + lane, 0, w - 1)];
				if(lane < 2 * o) {
					const int gx = lane < o ? x0 - o + lane : x0 + TW + (lane - o);
					Hh[r * 2 * o + lane] = flat[(size_t)gy * w + vp_clampi(gx, 0, w - 1)];
				}
			}
		} else {
			for(int r = wv; r < TH; r += NWAVES) {
				const int gy = y0 + r, gx = x0 + lane;
				I[r * TW + lane] = (gy < h && gx < w) ? (uint32_t)(int)gin[(size_t)gy * w + gx] : 0u;
			}
		}
		if(loader) misc[3] = (int)(pending - a.ticket_base); // requested a whole tile ago: long since returned
		__syncthreads();
		if(loader) pending = atomicAdd(queue, 1u); // after the draining barrier, so nobody waits for it
		const int next = misc[3];
		const bool more = next >= 0 && next < perFrame;

		// ---- B: gradient dot product, row scan (lanes), running column sum (registers) --------------------------------
		{
			int run = 0;
			const bool xin = x0 + lane < w;
			// horizontal taps cross into the halo array at the tile's edge columns
			const int xp = lane + o, xn = lane - o;
			const int offp = xp < TW ? xp : rowsPad * TW + o + (xp - TW), strp = xp < TW ? TW : 2 * o;
			const int offn = xn >= 0 ? xn : rowsPad * TW + o + xn, strn = xn >= 0 ? TW : 2 * o;
#pragma unroll 4
			for(int rr = 0; rr < SEG; rr++) {
				const int r = wv * SEG + rr;
				int g;
				if(FUSED) {
					const int rc = r + o;
					g = grad_dot_pk(I[offp + rc * strp], I[offn + rc * strn], I[(rc + o) * TW + lane], I[(rc - o) * TW + lane]);
					if(!(xin && y0 + r < h)) g = 0;
					if(gd_out && xin && y0 + r < h) gd_out[frame + (size_t)(y0 + r) * w + x0 + lane] = (float)g;
				} else {
					g = (int)I[r * TW + lane];
				}
				run += scan_incl_dpp(g);
				S[r * TW + lane] = run;
			}
			T[wv * TW + lane] = run;
		}
		lds_barrier();

		// ---- C/D: aggregates, publication, look-back (waves 0-2); wave 3 streams the next tile into LDS meanwhile ------
		const size_t tile = (size_t)f * perFrame + (size_t)ti * a.tilesX + tj; // granules are indexed by grid position
		if(wv == 0) { // column chain: A(x)
			const int pc = T[lane] + T[TW + lane] + T[2 * TW + lane] + T[3 * TW + lane]; // bottom row of the local SAT
			u64* g = a.V + tile * TW + lane;
			int carry = 0;
			if(ti > 0) {
				st_granule(g, a.epoch, ST_AGG, pc);
				lookback(g, (long)a.tilesX * TW, ti, a, carry);
			}
			st_granule(g, a.epoch, ST_INC, carry + pc);
			Av[lane] = carry;
		} else if(wv == 1) { // row chain: B(y), then the corner chain's aggregate Q = B(TH-1)
			const bool act = lane < TH;
			const int y = act ? lane : TH - 1;
			int pr = S[y * TW + TW - 1]; // right column of the local SAT
			const int seg = y / SEG;
			for(int s = 0; s < seg; s++) pr += T[s * TW + TW - 1];
			u64* g = a.H + tile * TH + y;
			int carry = 0;
			if(tj > 0) {
				if(act) st_granule(g, a.epoch, ST_AGG, pr);
				lookback(g, TH, tj, a, carry, act);
			}
			if(act) {
				st_granule(g, a.epoch, ST_INC, carry + pr);
				Bv[y] = carry;
			}
			if(lane == TH - 1) {
				misc[2] = carry; // Q: everything left of this tile within its tile row
				st_granule(a.K + tile, a.epoch, ti > 0 ? ST_AGG : ST_INC, carry);
			}
		} else if(wv == 2) { // corner chain: K = sum of Q over the tiles above (all lanes poll the same word)
			int carry = 0;
			if(ti > 0) lookback(a.K + tile, a.tilesX, ti, a, carry);
			if(lane == 0) misc[1] = carry;
		} else if(PIPE) { // wave 3: the loader.  I / Hh were last read in phase B, before the barrier above.
			staged = false;
			if(more) {
				const uint32_t tn = a.order[next];
				const int ntj = (int)(tn & 0xFFFF), nti = (int)(tn >> 16);
				if(ntj * TW + TW <= w) {
					dma_tile(ntj * TW, nti * TH);
					staged = true;
				}
			}
		}
		lds_barrier();
		const int K = misc[1];
		if(wv == 1 && lane == 0 && ti > 0) st_granule(a.K + tile, a.epoch, ST_INC, K + misc[2]);

		// ---- E: out = local + A(x) + B(y) + K -------------------------------------------------------------------------
		{
			const int cg = tid & 15, rr = tid >> 4;
			const int4 av = *reinterpret_cast<const int4*>(Av + 4 * cg);
			int4 base = make_int4(av.x + K, av.y + K, av.z + K, av.w + K);
			const int gx = x0 + 4 * cg;
			const bool vec = (gx + 3 < w) && ((w & 3) == 0) && ((reinterpret_cast<uintptr_t>(sat) & 15) == 0);
#pragma unroll
			for(int q = 0; q < TH / 16; q++) {
				const int r = q * 16 + rr, gy = y0 + r;
				const int seg = r / SEG;
				int4 add = base; // + totals of the 16-row... SEG-row segments above this row's segment
#pragma unroll
				for(int sgm = 0; sgm < NWAVES - 1; sgm++) {
					if(sgm < seg) {
						const int4 t = *reinterpret_cast<const int4*>(T + sgm * TW + 4 * cg);
						add = make_int4(add.x + t.x, add.y + t.y, add.z + t.z, add.w + t.w);
					}
				}
				const int4 s = *reinterpret_cast<const int4*>(S + r * TW + 4 * cg);
				const int b = Bv[r];
				const int4 v = make_int4(s.x + add.x + b, s.y + add.y + b, s.z + add.z + b, s.w + add.w + b);
				if(gy < h) {
					if(vec) *reinterpret_cast<int4*>(out + (size_t)gy * w + gx) = v;
					else {
						if(gx < w) out[(size_t)gy * w + gx] = v.x;
						if(gx + 1 < w) out[(size_t)gy * w + gx + 1] = v.y;
						if(gx + 2 < w) out[(size_t)gy * w + gx + 2] = v.z;
						if(gx + 3 < w) out[(size_t)gy * w + gx + 3] = v.w;
					}
				}
			}
		}
		if(!more) break;
		rem = next;
		lds_barrier(); // S/T/Av/Bv/misc of this tile are dead only now
	}
	(void)staged;
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
	a.batch = batch;
	const int rawRows = TH + 2 * o;
	a.off = 0;
	a.pitch = (rawRows * 2 * o + 63 + 3) & ~3; // halo array (dwords): whole 64-dword DMA pieces, 16-byte multiple
	const size_t lds = sizeof(int) * ((size_t)((rawRows + 3) & ~3) * TW + a.pitch + TH * TW + NWAVES * TW + TW + TH + 8);

	// persistent carry state (slot 1): [control: abort word + one 64-B ticket line per frame][V][H][K]
	const size_t bCtl = 64 * ((size_t)batch + 1);
	const size_t bOrd = (sizeof(uint32_t) * (size_t)a.tilesX * a.tilesY + 255) & ~size_t(255);
	const size_t bV = sizeof(u64) * (size_t)a.nTiles * TW, bH = sizeof(u64) * (size_t)a.nTiles * TH, bK = sizeof(u64) * (size_t)a.nTiles;
	void* base = nullptr;
	bool grew = false;
	VP_TRY(vp_workspace(ctx, stream, 1, bCtl + bOrd + bV + bH + bK, &base, &grew));
	if(!ctx->host_flag[stream]) {
		VP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->host_flag[stream]), 64, hipHostMallocMapped));
		*ctx->host_flag[stream] = 0;
	}
	const bool poisoned = *reinterpret_cast<volatile int*>(ctx->host_flag[stream]) != 0; // an earlier launch timed out (reported by vp_sync)
	// the per-frame ticket counters advance by tilesX*tilesY per launch: a different tiling or batch restarts the count
	const uint64_t shape = ((uint64_t)a.tilesX << 40) | ((uint64_t)a.tilesY << 20) | (uint64_t)batch;
	const bool reshaped = ctx->sat_shape[stream] != shape;
	ctx->sat_shape[stream] = shape;
	const bool reset = grew || poisoned || reshaped || ctx->sat_epoch[stream] >= (1u << 30) - 2;
	if(reset) {
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
	uint8_t* p = static_cast<uint8_t*>(base);
	a.ctrl = reinterpret_cast<uint32_t*>(p);
	a.order = reinterpret_cast<const uint32_t*>(p + bCtl);
	a.V = reinterpret_cast<u64*>(p + bCtl + bOrd);
	a.H = reinterpret_cast<u64*>(p + bCtl + bOrd + bV);
	a.K = reinterpret_cast<u64*>(p + bCtl + bOrd + bV + bH);
	if(reset) {
		// (re)build the visiting order of this tile grid: anti-diagonal wavefronts.  It is a linear extension of the carry
		// dependencies, and both the left and the upper neighbour of a tile are claimed about one diagonal length of tickets
		// earlier, so their carries are usually published by the time a tile looks back (raster / Morton orders leave the
		// left neighbour one ticket ahead: measured ~3.7 re-polls per chain).  VP_SAT_ORDER=raster|morton for experiments.
		std::vector<uint32_t> ord;
		ord.reserve((size_t)a.tilesX * a.tilesY);
		const char* oe = getenv("VP_SAT_ORDER");
		const int mode = !oe ? 0 : (oe[0] == 'r' ? 1 : (oe[0] == 'm' ? 2 : 0));
		if(mode == 0) {
			for(int d = 0; d < a.tilesX + a.tilesY - 1; d++)
				for(int y = 0; y < a.tilesY; y++) {
					const int x = d - y;
					if(x >= 0 && x < a.tilesX) ord.push_back(((uint32_t)y << 16) | (uint32_t)x);
				}
		} else if(mode == 1) {
			for(int y = 0; y < a.tilesY; y++)
				for(int x = 0; x < a.tilesX; x++) ord.push_back(((uint32_t)y << 16) | (uint32_t)x);
		} else {
			int bits = 0;
			while((1 << bits) < a.tilesX || (1 << bits) < a.tilesY) bits++;
			for(uint32_t z = 0; z < (1u << (2 * bits)); z++) {
				uint32_t x = 0, y = 0;
				for(int b = 0; b < bits; b++) {
					x |= ((z >> (2 * b)) & 1u) << b;
					y |= ((z >> (2 * b + 1)) & 1u) << b;
				}
				if((int)x < a.tilesX && (int)y < a.tilesY) ord.push_back((y << 16) | x);
			}
		}
		VP_CHECK_HIP(hipMemcpyAsync(p + bCtl, ord.data(), sizeof(uint32_t) * ord.size(), hipMemcpyHostToDevice, s));
		VP_CHECK_HIP(hipStreamSynchronize(s)); // ord is a temporary; this path runs only when the shape changes
	}
	int* dflag = nullptr;
	VP_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dflag), ctx->host_flag[stream], 0));
	a.host_flag = dflag;
	a.debug = getenv("VP_SAT_DEBUG") ? atoi(getenv("VP_SAT_DEBUG")) : 0;

	// persistent grid: P workgroups per frame queue; every workgroup overshoots its queue by exactly two tickets
	const int perFrame = a.tilesX * a.tilesY;
	int P = (4 * ctx->cu_count + batch - 1) / batch;
	if(P < 1) P = 1;
	if(P > perFrame) P = perFrame;
	ctx->sat_ticket_base[stream] += (uint32_t)(perFrame + 2 * P);
	const unsigned grid = (unsigned)P * (unsigned)batch;
	const bool pipe = fused && o >= 1 && (w & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && !(a.debug & 2);
#define VP_TILED_LAUNCH(M) \
	do { \
		if(lds > 48 * 1024) VP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sat_tiled<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
		k_sat_tiled<M><<<grid, 256, lds, s>>>(in, sat, graddot_out, a); \
	} while(0)
	if(!fused) VP_TILED_LAUNCH(0);
	else if(!pipe) VP_TILED_LAUNCH(-1);
	else VP_TILED_LAUNCH(1);
#undef VP_TILED_LAUNCH
	VP_TRY(vp_launch_ok("tiled sat"));
	if(a.debug & 4) { // look-back statistics (debug only): windows, re-polls, chains, predecessors visited
		uint32_t st[4];
		VP_CHECK_HIP(hipStreamSynchronize(s));
		VP_CHECK_HIP(hipMemcpy(st, a.ctrl + 8, sizeof(st), hipMemcpyDeviceToHost));
		fprintf(stderr, "[sat-debug] chains %u windows %u repolls %u predecessors %u\n", st[2], st[0], st[1], st[3]);
		VP_CHECK_HIP(hipMemset(a.ctrl + 8, 0, sizeof(st)));
	}
	return VP_OK;
}
