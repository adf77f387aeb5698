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
// Forward progress does not depend on dispatch order: tiles are claimed from a ticket counter in an order that is a linear
// extension of the carry dependencies (anti-diagonals), so a workgroup only ever waits on tiles whose workgroups have started.
// Every spin is bounded; on expiry the kernel raises a device abort word and a pinned host flag, finishes with garbage and the
// host reports VP_ERR_TIMEOUT.
#include "vp_common.h"

#include <vector>

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;

namespace {

constexpr int TW = 64, TH = 64, NWAVES = 4, SEG = TH / NWAVES;
constexpr unsigned SPIN_LIMIT = 1u << 22; // polls (each with s_sleep) a look-back wait may take by default: seconds, never reached in a healthy run (vp_ctx_tune "sat_spin_limit")
constexpr uint32_t ST_AGG = 1, ST_INC = 2;

struct TiledArgs {
	int w, h, o;
	int tilesX, tilesY, nTiles, batch;
	int pitch;      // generic kernel: size of the LDS halo array in dwords
	uint32_t tickets_per_launch; // tickets every launch takes from each frame's queue: LAUNCH-INVARIANT for a given shape (host: reset when it changes)
	uint32_t ticket_base, epoch; // NOT launch arguments: derived inside the kernel from the first ticket a workgroup draws (see sat_launch_state)
	u64 *V, *H, *K; // carry granules: [tile][TW], [tile][TH], [tile]
	const uint32_t* order; // ticket -> (ti << 16 | tj): anti-diagonal order of the tile grid (a linear extension of the carry dependencies)
	uint32_t* ctrl; // [0] abort word, [16 + 16 f] ticket counter of frame f (monotonic across launches, one 64-B line each)
	int* host_flag; // pinned, device-mapped
	VpStrips strips; // optional halo strips (base == nullptr: halo columns are read from flat)
	int single;     // k_sat_fast2: as many workgroups as tiles per frame -- each takes ONE ticket and none in advance
	int32_t* zero;  // optional: int[3 x batch] cleared by the first workgroup of each frame (the blob list's counters, main.cpp:283-288)
	uint32_t spin_limit; // polls before a look-back wait gives up (raise_timeout)
	int drop_tile;       // fault injection (vp_ctx_tune "sat_debug_drop_tile", tests only; -1 = none): the column chain of this tile (frame * tiles + ti * tilesX + tj) publishes NOTHING,
	                     // so the tile below it waits until the limit, raises the timeout and the launch drains with a wrong table -- the library's one recovery path, exercised by
	                     // tests/test_gpu_round6.py.  One scalar compare per tile.
};

__device__ __forceinline__ u64 ld_granule(const u64* p) {
	return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(u64* p, uint32_t epoch, uint32_t state, int value) {
	__hip_atomic_store((gu64*)p, ((u64)((epoch << 2) | state) << 32) | (uint32_t)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void raise_timeout(const TiledArgs& a) {
	__hip_atomic_store((gu32*)a.ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(a.host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool aborted(const TiledArgs& a) {
	return __hip_atomic_load((gu32*)a.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// Which launch is this?  The per-frame ticket counters are monotonic across launches and every launch of a given shape draws exactly
// tickets_per_launch tickets from each of them (launches of one stream are ordered), so the first ticket t0 a workgroup draws names its launch:
// number L = t0 / tickets_per_launch, ticket base L * tickets_per_launch, granule epoch L + 1.  Nothing that changes from launch to launch is a
// kernel argument: the launch can be captured in a hipGraph and replayed (vp_ctx_tune "frame_graph").
__device__ __forceinline__ void sat_launch_state(uint32_t t0, uint32_t tickets_per_launch, uint32_t& base, uint32_t& epoch) {
	const uint32_t L = t0 / tickets_per_launch;
	base = L * tickets_per_launch;
	epoch = L + 1u;
}

// Exclusive carry of this lane's element along one chain: walk the predecessors (nearest first), adding aggregates
// until an inclusive granule is met.  `g` points at THIS tile's granule for this lane, predecessors are `stride`
// granules apart.  Wave-uniform control flow; returns false on timeout / abort.
__device__ __forceinline__ bool lookback(const u64* g, long stride, int npred, const TiledArgs& a, int& carry, bool active = true) {
	constexpr int WIN = 4; // predecessor granules kept in flight per round
	int acc = 0;
	bool done = !active; // lanes beyond the vector length ride along
	carry = 0;
	for(int k0 = 1; k0 <= npred; k0 += WIN) {
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
				if(spins > a.spin_limit || ((spins & 63u) == 63u && aborted(a))) {
					if((threadIdx.x & 63) == 0) raise_timeout(a);
					return false;
				}
				__builtin_amdgcn_s_sleep(1);
				v[q] = ld_granule(g - (long)(k0 + q) * stride);
				tag = (uint32_t)(v[q] >> 32);
			}
			if(!done) {
				acc += (int)(uint32_t)v[q];
				done = (tag & 3u) == ST_INC;
			}
			if(__all(done)) {
				carry = acc;
				return true;
			}
		}
	}
	carry = acc;
	return true;
}

// Workgroup barrier that orders LDS only (no vmcnt drain).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The two nearest predecessors of a chain were polled EARLY (pre0 = 1 back, pre1 = 2 back, loaded at the top of the tile, before
// its own sums exist): with the anti-diagonal order they are nearly always inclusive by now, so the carry resolves without a
// memory round trip on the critical path.  A true return tells the caller it may skip the aggregate state.
__device__ __forceinline__ bool early_resolve(u64 pre0, u64 pre1, int npred, const TiledArgs& a, int& carry) {
	const uint32_t t0 = (uint32_t)(pre0 >> 32), t1 = (uint32_t)(pre1 >> 32);
	const bool r0 = (t0 >> 2) == a.epoch && (t0 & 3u) != 0;
	const bool inc0 = r0 && (t0 & 3u) == ST_INC;
	const bool inc1 = npred >= 2 && (t1 >> 2) == a.epoch && (t1 & 3u) == ST_INC;
	carry = (int)(uint32_t)pre0 + (inc0 ? 0 : (int)(uint32_t)pre1);
	return __all(inc0 || (r0 && inc1));
}

// Generic kernel: any gradient offset, any width.  FUSED = false: the input is a float gradDot image (vp_sat); FUSED = true: the
// gradient dot product is formed from the flat tile.  Plain synchronous tile loads: the shapes real geometries produce (offset
// 1..4) run k_sat_fast2 below; this is the fallback for offset 0 / > 4 and for the unfused entry point.
// LDS image of a tile: interior I[rows][64] dwords (rows = TH + 2o) followed by the halo columns Hh[rows][2o].
template<bool FUSED>
__global__ __launch_bounds__(256) void k_sat_tiled(const void* __restrict__ in, int32_t* __restrict__ sat, float* __restrict__ gd_out, const TiledArgs a_in) {
	TiledArgs a = a_in; // ticket_base and epoch are filled in below
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int o = FUSED ? a.o : 0;
	const int rows = TH + 2 * o;
	uint32_t* I = reinterpret_cast<uint32_t*>(smem);
	uint32_t* Hh = I + rows * TW;
	int* S = reinterpret_cast<int*>(Hh + a.pitch); // a.pitch = halo array size in dwords, padded to a multiple of 4
	int* T = S + TH * TW;
	int* Av = T + NWAVES * TW;
	int* Bv = Av + TW;
	int* misc = Bv + TH; // [0] first ticket, [1] K, [2] Q, [3] next ticket

	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int w = a.w, h = a.h;
	// One ticket queue per frame: workgroup b serves frame b % batch; a workgroup works through its tickets in increasing order.
	// (A single counter would serialise the grid at ~88 atomics/us.)
	const int perFrame = a.tilesX * a.tilesY;
	const int f = (int)(blockIdx.x % (unsigned)a.batch);
	uint32_t* queue = a.ctrl + 16 + 16 * f;
	const size_t frame = (size_t)f * w * h;
	const uint32_t* flat = reinterpret_cast<const uint32_t*>(in) + frame;
	const float* gin = reinterpret_cast<const float*>(in) + frame;
	int32_t* out = sat + frame;
	const bool loader = tid == 3 * 64; // lane 0 of wave 3 owns the ticket queue

	if(a.zero && blockIdx.x < (unsigned)a.batch && tid < 3) a.zero[3 * blockIdx.x + tid] = 0;
	uint32_t pending = 0; // ticket requested one tile ahead
	if(loader) {
		const uint32_t t0 = atomicAdd(queue, 1u);
		pending = atomicAdd(queue, 1u);
		uint32_t base, epoch;
		sat_launch_state(t0, a.tickets_per_launch, base, epoch);
		misc[0] = (int)(t0 - base);
		misc[4] = (int)base;
		misc[5] = (int)epoch;
	}
	__syncthreads();
	int rem = misc[0];
	a.ticket_base = (uint32_t)__builtin_amdgcn_readfirstlane(misc[4]);
	a.epoch = (uint32_t)__builtin_amdgcn_readfirstlane(misc[5]);
	if(rem < 0 || rem >= perFrame) return; // two tickets taken, none valid: the host accounts for exactly two overshoots

	for(;;) {
		const uint32_t tcur = a.order[rem];
		const int ti = (int)(tcur >> 16), tj = (int)(tcur & 0xFFFF);
		const int x0 = tj * TW, y0 = ti * TH;

		// ---- A: the tile (+ halo) in LDS; publish the next ticket, request the one after --------------------------------
		if(FUSED) {
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
		if(loader) pending = atomicAdd(queue, 1u);
		const int next = misc[3];
		const bool more = next >= 0 && next < perFrame;

		// ---- B: gradient dot product, row scan (lanes), running column sum (registers) --------------------------------
		{
			int run = 0;
			const bool xin = x0 + lane < w;
			// horizontal taps cross into the halo array at the tile's edge columns
			const int xp = lane + o, xn = lane - o;
			const int offp = xp < TW ? xp : rows * TW + o + (xp - TW), strp = xp < TW ? TW : 2 * o;
			const int offn = xn >= 0 ? xn : rows * TW + o + xn, strn = xn >= 0 ? TW : 2 * o;
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

		// ---- C/D: aggregates, publication, look-back (waves 0-2) ----------------------------------------------------------
		const size_t tile = (size_t)f * perFrame + (size_t)ti * a.tilesX + tj; // granules are indexed by grid position
		if(wv == 0) { // column chain: A(x)
			const int pc = T[lane] + T[TW + lane] + T[2 * TW + lane] + T[3 * TW + lane]; // bottom row of the local SAT
			u64* g = a.V + tile * TW + lane;
			int carry = 0;
			const bool drop = (long)tile == (long)a.drop_tile; // fault injection, see TiledArgs
			if(ti > 0) {
				if(!drop) st_granule(g, a.epoch, ST_AGG, pc);
				lookback(g, (long)a.tilesX * TW, ti, a, carry);
			}
			if(!drop) st_granule(g, a.epoch, ST_INC, carry + pc);
			Av[lane] = carry;
		} else if(wv == 1) { // row chain: B(y), then the corner chain's aggregate Q = B(TH-1)
			const bool act = lane < TH;
			const int y = act ? lane : TH - 1;
			int pr = S[y * TW + TW - 1]; // right column of the local SAT
			const int seg = y / SEG;
			for(int sg = 0; sg < seg; sg++) pr += T[sg * TW + TW - 1];
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
				int4 add = base; // + totals of the SEG-row segments above this row's segment
#pragma unroll
				for(int sgm = 0; sgm < NWAVES - 1; sgm++) {
					if(sgm < seg) {
						const int4 t = *reinterpret_cast<const int4*>(T + sgm * TW + 4 * cg);
						add = make_int4(add.x + t.x, add.y + t.y, add.z + t.z, add.w + t.w);
					}
				}
				const int4 sv = *reinterpret_cast<const int4*>(S + r * TW + 4 * cg);
				const int b = Bv[r];
				const int4 v = make_int4(sv.x + add.x + b, sv.y + add.y + b, sv.z + add.z + b, sv.w + add.w + b);
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
}


// ---------------------------------------------------------------------------------------------------------------------
// Fast path (compile-time gradient offset O in 1..4): the kernel the benchmark runs.  Each design point answers a measurement
// (DESIGN.md section 4 has the history):
//   * phase B works on 4 pixels per lane: a 16-lane DPP row holds one 64-pixel tile row, a wavefront sweeps 4 row blocks at
//     once, taps come from ds_read_b128, the row scan is 3 serial adds + 4 DPP steps, column sums stay in registers;
//   * the sub-segment-local SAT plane lives in registers: phase E uses phase B's pixel-to-lane mapping, so the 16 values a lane
//     produced are the 16 values it completes and stores (24.6 KB of LDS per workgroup: 5 or 6 workgroups per CU);
//   * the next tile is prefetched into registers by all 256 threads (5 x 16 B + <= 2 x 2 halo dwords each) behind the look-back
//     and written to LDS at the top of the next iteration; its addresses are 32-bit lane offsets against buffer descriptors with
//     a scalar tile offset (no 64-bit VGPR arithmetic, no divisions in the loop);
//   * wave 3, which has no carry chain, turns the sub-segment totals into their exclusive prefix while waves 0-2 look back.
// LDS image of a tile: raw[rows][72] dwords = [4 - O pad | O halo | 64 interior | O halo | pad], rows = 64 + 2 O.
// RAG: width not a multiple of 64 (the last tile column clamps / masks per pixel).  GD: also write the F32 gradDot view.
// LDC: the tile loads use the default cache policy instead of nt.  A launch whose flat fits the Infinity Cache beside its table (<= 112 MiB: 8 frames of
// cfg2) reads what the resampling kernel has just written: cached loads take 4.4 % off the whole frame at 8 frames and 3.5 % at one, nothing from 16 on
// (interleaved A/B, profiles/r04_sat_cache_policy.json); cached STORES lose at every size (+ 7 % at 128 frames), so the table is always streamed out.
template<int O, int WPC, bool GD, bool RAG, bool LDC = false>
__global__ __launch_bounds__(256, WPC) void k_sat_fast2(const uint32_t* __restrict__ in, int32_t* __restrict__ sat, float* __restrict__ gd_out, const TiledArgs a_in) {
	TiledArgs a = a_in; // ticket_base and epoch are filled in below
	static_assert(O >= 1 && O <= 4 && TW == 64 && TH == 64, "fast path geometry");
	constexpr int ROWS = TH + 2 * O, PITCH = TW + 8;
	constexpr int NI = (ROWS + 15) / 16;             // interior: 16 lanes x 16 B per row, 16 rows per pass
	constexpr int NHS = (ROWS * O + 255) / 256;      // halo dwords per thread and side
	extern __shared__ __attribute__((aligned(16))) char smem[];
	uint32_t* raw = reinterpret_cast<uint32_t*>(smem);
	int* T16 = reinterpret_cast<int*>(raw + ROWS * PITCH); // [16][64] column totals of the 4-row sub-segments
	int* T16P = T16 + 16 * TW;                             // [16][64] exclusive prefix of T16 over sub-segments (wave 3, during C/D)
	int* Rc = T16P + 16 * TW;                              // [64] right column of the sub-segment-local SAT
	int* Av = Rc + TH;
	int* Bv = Av + TW;
	int* misc = Bv + TH; // [0] first ticket, [1] K, [2] Q, [3] next ticket

	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int w = a.w, h = a.h;
	const int perFrame = a.tilesX * a.tilesY;
	const int f = (int)(blockIdx.x % (unsigned)a.batch);
	uint32_t* queue = a.ctrl + 16 + 16 * f;
	const size_t frame = (size_t)f * w * h;
	const bool loader = tid == 3 * 64;

	if(a.zero && blockIdx.x < (unsigned)a.batch && tid < 3) a.zero[3 * blockIdx.x + tid] = 0;
	uint32_t pending = 0;
	if(loader) {
		const uint32_t t0 = atomicAdd(queue, 1u);
		uint32_t base, epoch;
		sat_launch_state(t0, a.tickets_per_launch, base, epoch);
		// (with one workgroup per tile a second ticket could only be out of range: not asking for it halves the burst on the
		// frame's ticket counter, which is what a single-frame launch spends a third of its time on)
		pending = a.single ? base + (uint32_t)perFrame : atomicAdd(queue, 1u);
		misc[0] = (int)(t0 - base);
		misc[4] = (int)base;
		misc[5] = (int)epoch;
	}
	__syncthreads();
	int rem = misc[0];
	a.ticket_base = (uint32_t)__builtin_amdgcn_readfirstlane(misc[4]);
	a.epoch = (uint32_t)__builtin_amdgcn_readfirstlane(misc[5]);
	if(rem < 0 || rem >= perFrame) return;

	// buffer descriptors of this frame (host guarantees w * h * 4 < 2^31): 32-bit lane offsets + a scalar tile offset
	const uint32_t frame_bytes = (uint32_t)w * (uint32_t)h * 4u, row_bytes = (uint32_t)w * 4u;
	const __amdgpu_buffer_rsrc_t rs_flat = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(in + frame), 0, (int)frame_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(sat + frame, 0, (int)frame_bytes, 0x00020000);
	const bool use_strips = a.strips.base != nullptr;
	const uint32_t strip_bytes = (uint32_t)(a.strips.frame_words * 4);
	const __amdgpu_buffer_rsrc_t rs_strip = __builtin_amdgcn_make_buffer_rsrc(use_strips ? a.strips.base + (size_t)f * a.strips.frame_words : const_cast<uint32_t*>(in + frame), 0,
			(int)(use_strips ? strip_bytes : frame_bytes), 0x00020000);

	// ---- register image of the next tile: NI x 16 B of interior + NHS dwords of halo per side ------------------------------
	typedef unsigned int v4u __attribute__((ext_vector_type(4)));
	v4u pv0, pv1, pv2, pv3, pv4;
	uint32_t hl0 = 0, hl1 = 0, hr0 = 0, hr1 = 0;
	// The index constants of the prefetch (c4, r16, halo row / column) are recomputed from an opaque copy of tid wherever they
	// are used: hoisted out of the tile loop they would occupy ~10 VGPRs for the whole kernel.
	int tidv = tid;
#define VP2_IDX()                                                                                          \
	asm volatile("" : "+v"(tidv));                                                                         \
	const int c4 = tidv & 15, r16 = tidv >> 4;                                                             \
	/* halo element i of a side: row i / O, column k = i % O (compile-time O) */                           \
	const int hi0 = min(tidv, ROWS * O - 1), hi1 = min(tidv + 256, ROWS * O - 1);                          \
	const int hrow0 = hi0 / O, hk0 = hi0 - hrow0 * O, hrow1 = hi1 / O, hk1 = hi1 - hrow1 * O;              \
	(void)hrow1; (void)hk1
	auto clamp_row = [&](int Y0, int row) { return (uint32_t)vp_clampi(Y0 - O + row, 0, h - 1); };
	// VP_SAT_ABLATE (tools/abbuild.sh builds for the traffic attribution of profiles/r06_pmc_sat_attribution.json; WRONG RESULTS BY DESIGN, never defined in the product):
	// bit 0: the 2 O halo ROWS of a tile are not loaded, bit 1: the halo COLUMNS are not loaded
#ifndef VP_SAT_ABLATE
#define VP_SAT_ABLATE 0
#endif
	auto tile_load = [&](int P, int X0, int Y0, int c4, int r16) -> v4u {
		if((VP_SAT_ABLATE & 1) && (min(r16 + 16 * P, ROWS - 1) < O || min(r16 + 16 * P, ROWS - 1) >= TH + O)) return v4u{0, 0, 0, 0};
		const uint32_t rowoff = clamp_row(Y0, min(r16 + 16 * P, ROWS - 1)) * row_bytes;
		if(!RAG || X0 + TW <= w) return (v4u)__builtin_amdgcn_raw_buffer_load_b128(rs_flat, rowoff + 16u * (uint32_t)c4, (uint32_t)X0 * 4u, LDC ? 0 : 2); // aux 2 = nt: read once
		// the last tile column of a width that is not a multiple of 64 (workgroup-uniform): columns past the image take the edge
		// pixel, clamp-to-edge as gradientDot.cl:20 reads them
		v4u v;
		const int gx = X0 + 4 * c4;
		v.x = __builtin_amdgcn_raw_buffer_load_b32(rs_flat, rowoff + 4u * (uint32_t)min(gx, w - 1), 0, 0);
		v.y = __builtin_amdgcn_raw_buffer_load_b32(rs_flat, rowoff + 4u * (uint32_t)min(gx + 1, w - 1), 0, 0);
		v.z = __builtin_amdgcn_raw_buffer_load_b32(rs_flat, rowoff + 4u * (uint32_t)min(gx + 2, w - 1), 0, 0);
		v.w = __builtin_amdgcn_raw_buffer_load_b32(rs_flat, rowoff + 4u * (uint32_t)min(gx + 3, w - 1), 0, 0);
		return v;
	};
	// left side: pixels X0 - O + k; right side: X0 + 64 + k (clamp-to-edge at the image border, gradientDot.cl:20)
	auto halo_load = [&](bool right, int row, int k, int X0, int Y0) -> uint32_t {
		if(VP_SAT_ABLATE & 2) return 0u;
		const uint32_t gy = clamp_row(Y0, row);
		const int tjx = X0 >> 6;
		const bool border = right ? tjx + 1 >= a.tilesX : tjx == 0;
		if(use_strips && !border) { // right halo: FIRST pixels of tile tjx + 1; left halo: LAST pixels of tile tjx - 1 (VpStrips::rec)
			const int j = right ? tjx + 1 : tjx - 1;
			return __builtin_amdgcn_raw_buffer_load_b32(rs_strip, (gy * 2u * (uint32_t)O + (right ? 0u : (uint32_t)O) + (uint32_t)k) * 4u, (uint32_t)j * (uint32_t)a.strips.hp * 2u * (uint32_t)O * 4u, 0);
		}
		if(border) return __builtin_amdgcn_raw_buffer_load_b32(rs_flat, gy * row_bytes, right ? row_bytes - 4u : 0u, 0); // edge column replicated
		// (the column clamp only acts next to a partial last tile column: its first pixels may already lie past the image)
		if(RAG) return __builtin_amdgcn_raw_buffer_load_b32(rs_flat, gy * row_bytes + 4u * (uint32_t)min((right ? X0 + TW : X0 - O) + k, w - 1), 0, 0);
		return __builtin_amdgcn_raw_buffer_load_b32(rs_flat, gy * row_bytes + 4u * (uint32_t)k, (uint32_t)(right ? X0 + TW : X0 - O) * 4u, 0);
	};
#define VP2_PREFETCH(X0, Y0)                                           \
	do {                                                                 \
		VP2_IDX();                                                       \
		pv0 = tile_load(0, X0, Y0, c4, r16);                             \
		pv1 = tile_load(1, X0, Y0, c4, r16);                             \
		pv2 = tile_load(2, X0, Y0, c4, r16);                             \
		pv3 = tile_load(3, X0, Y0, c4, r16);                             \
		if(NI > 4) pv4 = tile_load(4, X0, Y0, c4, r16);                  \
		hl0 = halo_load(false, hrow0, hk0, X0, Y0);                      \
		hr0 = halo_load(true, hrow0, hk0, X0, Y0);                       \
		if(NHS > 1) {                                                    \
			hl1 = halo_load(false, hrow1, hk1, X0, Y0);                  \
			hr1 = halo_load(true, hrow1, hk1, X0, Y0);                   \
		}                                                                \
	} while(0)
#define VP2_COMMIT_ROW(P, V)                                                                                          \
	do {                                                                                                                \
		if(r16 + 16 * (P) < ROWS) *reinterpret_cast<v4u*>(raw + (r16 + 16 * (P)) * PITCH + 4 + 4 * c4) = (V);           \
	} while(0)
#define VP2_COMMIT_ALL()                                                                     \
	do {                                                                                       \
		VP2_IDX();                                                                             \
		VP2_COMMIT_ROW(0, pv0);                                                                \
		VP2_COMMIT_ROW(1, pv1);                                                                \
		VP2_COMMIT_ROW(2, pv2);                                                                \
		VP2_COMMIT_ROW(3, pv3);                                                                \
		if(NI > 4) VP2_COMMIT_ROW(4, pv4);                                                     \
		if(tidv < ROWS * O) {                                                                  \
			raw[hrow0 * PITCH + 4 - O + hk0] = hl0;                                            \
			raw[hrow0 * PITCH + 4 + TW + hk0] = hr0;                                           \
		}                                                                                      \
		if(NHS > 1 && tidv + 256 < ROWS * O) {                                                 \
			raw[hrow1 * PITCH + 4 - O + hk1] = hl1;                                            \
			raw[hrow1 * PITCH + 4 + TW + hk1] = hr1;                                           \
		}                                                                                      \
	} while(0)
	pv4 = v4u{0, 0, 0, 0};
	{
		const uint32_t t0 = a.order[__builtin_amdgcn_readfirstlane(rem)];
		VP2_PREFETCH((int)(t0 & 0xFFFF) * TW, (int)(t0 >> 16) * TH);
	}

	// phase B / E mapping: lane = 16 ql + m, columns 4m..4m+3, rows 16 wv + 4 q + it, q = bit-reversed ql: a ds_read_b128 is serviced in
	// four 16-lane groups that each mix lanes of two adjacent 16-lane rows ({0-3, 12-15, 20-27}, ... MI355X_MICROARCH.md, LDS); with the
	// 72-dword pitch those two rows must lie a multiple of 8 tile rows apart (72 * 8 = 0 mod 64 banks) for the group to cover 64
	// distinct banks.  ql = 0,1,2,3 -> row blocks 0, 8, 4, 12: all five tap reads are conflict-free (with q = ql each was 2-way conflicted).
	const int m = lane & 15, ql = lane >> 4, q = ((ql & 1) << 1) | (ql >> 1);
	const int sg_mine = 4 * wv + q;
	const uint32_t out_voff = ((uint32_t)(16 * wv + 4 * q) * (uint32_t)w + 4u * (uint32_t)m) * 4u;

	for(;;) {
		// (the tile-order table is written by the host before the launch and only read here: a SCALAR load.  As a vector load at the top of the tile loop it made
		// the wavefront wait for vmcnt(0) -- for the previous tile's stores to retire -- before it could commit the prefetched tile; now the wait is vmcnt(4):
		// - 0.4 % per 256-frame launch, - 3 % per 8-frame and single-frame launch)
		typedef __attribute__((address_space(4))) const uint32_t c_u32;
		const uint32_t tcur = ((c_u32*)a.order)[__builtin_amdgcn_readfirstlane(rem)];
		const int ti = (int)(tcur >> 16), tj = (int)(tcur & 0xFFFF);
		const int x0 = tj * TW, y0 = ti * TH;

		// ---- A: registers -> LDS; publish the next ticket, request the one after --------------------------------------
		VP2_COMMIT_ALL();
		if(loader) misc[3] = (int)(pending - a.ticket_base);
		lds_barrier();
		if(loader && !a.single) pending = atomicAdd(queue, 1u);
		const int next = misc[3];
		const bool more = next >= 0 && next < perFrame;

		// early polls: the carries of the two nearest predecessors of each chain, requested before our own sums exist
		const size_t tile = (size_t)f * perFrame + (size_t)ti * a.tilesX + tj;
		u64 pre0 = 0, pre1 = 0;
		int lanev = lane; // opaque copy: lane-dependent 64-bit granule pointers must not be hoisted out of the tile loop (4 VGPRs each)
		asm volatile("" : "+v"(lanev));
		if(wv == 0 && ti > 0) {
			const u64* g = a.V + tile * TW + lanev;
			pre0 = ld_granule(g - (long)a.tilesX * TW);
			if(ti > 1) pre1 = ld_granule(g - 2 * (long)a.tilesX * TW);
		} else if(wv == 1 && tj > 0) {
			const u64* g = a.H + tile * TH + lanev;
			pre0 = ld_granule(g - TH);
			if(tj > 1) pre1 = ld_granule(g - 2 * TH);
		} else if(wv == 2 && ti > 0) {
			const u64* g = a.K + tile;
			pre0 = ld_granule(g - a.tilesX);
			if(ti > 1) pre1 = ld_granule(g - 2 * (long)a.tilesX);
		}

		// ---- B: local SAT of the lane's 4 x 4 pixels, kept in registers -------------------------------------------------
		const bool bottom = y0 + TH > h, ragged = RAG && x0 + TW > w; // RAG: width not a multiple of 64 (host-selected instantiation)
		int4 sres[4];
		{
			int run0 = 0, run1 = 0, run2 = 0, run3 = 0;
#pragma unroll
			for(int it = 0; it < 4; it++) {
				const int r = 16 * wv + 4 * q + it;
				const uint32_t* c = raw + (r + O) * PITCH + 4 + 4 * m;
				const uint4 L = *reinterpret_cast<const uint4*>(c - 4), C = *reinterpret_cast<const uint4*>(c), R = *reinterpret_cast<const uint4*>(c + 4);
				const uint4 U = *reinterpret_cast<const uint4*>(c - O * PITCH), D = *reinterpret_cast<const uint4*>(c + O * PITCH);
				const uint32_t hx[12] = {L.x, L.y, L.z, L.w, C.x, C.y, C.z, C.w, R.x, R.y, R.z, R.w};
				int g0 = grad_dot_pk(hx[4 + O], hx[4 - O], D.x, U.x);
				int g1 = grad_dot_pk(hx[5 + O], hx[5 - O], D.y, U.y);
				int g2 = grad_dot_pk(hx[6 + O], hx[6 - O], D.z, U.z);
				int g3 = grad_dot_pk(hx[7 + O], hx[7 - O], D.w, U.w);
				if(bottom) { // only tiles that overhang the image mask rows / columns (workgroup-uniform)
					if(y0 + r >= h) g0 = g1 = g2 = g3 = 0;
				}
				if(ragged) {
					const int gx = x0 + 4 * m;
					if(gx >= w) g0 = 0;
					if(gx + 1 >= w) g1 = 0;
					if(gx + 2 >= w) g2 = 0;
					if(gx + 3 >= w) g3 = 0;
				}
				if(GD && y0 + r < h) {
					float* gp = gd_out + frame + (size_t)(y0 + r) * w + x0 + 4 * m;
					if(!ragged) *reinterpret_cast<float4*>(gp) = make_float4((float)g0, (float)g1, (float)g2, (float)g3);
					else {
						const int gx = x0 + 4 * m;
						if(gx < w) gp[0] = (float)g0;
						if(gx + 1 < w) gp[1] = (float)g1;
						if(gx + 2 < w) gp[2] = (float)g2;
						if(gx + 3 < w) gp[3] = (float)g3;
					}
				}
				g1 += g0;
				g2 += g1;
				g3 += g2;
				int sc = g3; // inclusive scan of the lane totals over the 16 lanes of this row
				sc += __builtin_amdgcn_update_dpp(0, sc, 0x111, 0xf, 0xf, false);
				sc += __builtin_amdgcn_update_dpp(0, sc, 0x112, 0xf, 0xf, false);
				sc += __builtin_amdgcn_update_dpp(0, sc, 0x114, 0xf, 0xf, false);
				sc += __builtin_amdgcn_update_dpp(0, sc, 0x118, 0xf, 0xf, false);
				const int ex = sc - g3;
				run0 += g0 + ex;
				run1 += g1 + ex;
				run2 += g2 + ex;
				run3 += g3 + ex;
				sres[it] = make_int4(run0, run1, run2, run3);
				if(m == 15) Rc[r] = run3;
				__builtin_amdgcn_sched_barrier(0); // one row at a time: overlapping the taps of several rows costs ~20 VGPRs
			}
			*reinterpret_cast<int4*>(T16 + sg_mine * TW + 4 * m) = make_int4(run0, run1, run2, run3);
		}
		lds_barrier();

		// ---- C/D: aggregates, publication, look-back --------------------------------------------------------------------------
		if(wv == 0) {
			int pc = 0;
#pragma unroll
			for(int sg = 0; sg < 16; sg++) pc += T16[sg * TW + lane];
			asm volatile("" : "+v"(lanev));
			u64* g = a.V + tile * TW + lanev;
			int carry = 0;
			const bool drop = (long)tile == (long)a.drop_tile; // fault injection, see TiledArgs (scalar: tile is workgroup-uniform)
			if(ti > 0 && !early_resolve(pre0, pre1, ti, a, carry)) {
				if(!drop) st_granule(g, a.epoch, ST_AGG, pc);
				lookback(g, (long)a.tilesX * TW, ti, a, carry);
			}
			if(!drop) st_granule(g, a.epoch, ST_INC, carry + pc);
			Av[lane] = carry;
		} else if(wv == 1) {
			const int t = T16[(lane >> 2) * TW + TW - 1];
			const int pr = Rc[lane] + scan_incl_dpp((lane & 3) == 0 ? t : 0) - t;
			asm volatile("" : "+v"(lanev));
			u64* g = a.H + tile * TH + lanev;
			int carry = 0;
			if(tj > 0 && !early_resolve(pre0, pre1, tj, a, carry)) {
				st_granule(g, a.epoch, ST_AGG, pr);
				lookback(g, TH, tj, a, carry);
			}
			st_granule(g, a.epoch, ST_INC, carry + pr);
			Bv[lane] = carry;
			if(lane == TH - 1) {
				misc[2] = carry;
				st_granule(a.K + tile, a.epoch, ti > 0 ? ST_AGG : ST_INC, carry);
			}
		} else if(wv == 2) {
			int carry = 0;
			if(ti > 0 && !early_resolve(pre0, pre1, ti, a, carry)) lookback(a.K + tile, a.tilesX, ti, a, carry);
			if(lane == 0) misc[1] = carry;
		} else { // wave 3 has no chain: it turns the sub-segment totals into their exclusive prefix for phase E
			int acc = 0;
#pragma unroll
			for(int sg = 0; sg < 16; sg++) {
				T16P[sg * TW + lane] = acc;
				acc += T16[sg * TW + lane];
			}
		}
		lds_barrier();
		const int K = misc[1];
		if(wv == 1 && lane == 0 && ti > 0) st_granule(a.K + tile, a.epoch, ST_INC, K + misc[2]);
		if(more) { // next tile -> registers: in flight across phase E and the loop barrier
			const uint32_t tn = a.order[__builtin_amdgcn_readfirstlane(next)];
			VP2_PREFETCH((int)(tn & 0xFFFF) * TW, (int)(tn >> 16) * TH);
		}

		// ---- E: out = local + (sub-segment totals above) + A(x) + B(y) + K, for the lane's own 16 pixels --------------------
		{
			const int4 av = *reinterpret_cast<const int4*>(Av + 4 * m);
			int4 add = make_int4(av.x + K, av.y + K, av.z + K, av.w + K);
			{
				const int4 t = *reinterpret_cast<const int4*>(T16P + sg_mine * TW + 4 * m);
				add = make_int4(add.x + t.x, add.y + t.y, add.z + t.z, add.w + t.w);
			}
			const uint32_t tile_off = ((uint32_t)y0 * (uint32_t)w + (uint32_t)x0) * 4u;
			typedef int v4i __attribute__((ext_vector_type(4)));
#pragma unroll
			for(int it = 0; it < 4; it++) {
				const int r = 16 * wv + 4 * q + it;
				const int b = Bv[r];
				const v4i o4 = {sres[it].x + add.x + b, sres[it].y + add.y + b, sres[it].z + add.z + b, sres[it].w + add.w + b};
				if(y0 + r < h) {
					if(!ragged) __builtin_amdgcn_raw_buffer_store_b128(o4, rs_out, out_voff + (uint32_t)it * row_bytes, tile_off, 2); // nt: streamed out
					else {
						const int gx = x0 + 4 * m;
						const uint32_t vo = out_voff + (uint32_t)it * row_bytes;
						if(gx < w) __builtin_amdgcn_raw_buffer_store_b32(o4.x, rs_out, vo, tile_off, 2);
						if(gx + 1 < w) __builtin_amdgcn_raw_buffer_store_b32(o4.y, rs_out, vo + 4u, tile_off, 2);
						if(gx + 2 < w) __builtin_amdgcn_raw_buffer_store_b32(o4.z, rs_out, vo + 8u, tile_off, 2);
						if(gx + 3 < w) __builtin_amdgcn_raw_buffer_store_b32(o4.w, rs_out, vo + 12u, tile_off, 2);
					}
				}
			}
		}
		if(!more) break;
		rem = next;
		lds_barrier(); // T16 / Rc / Av / Bv / misc of this tile are dead only now (phase E read them)
	}
#undef VP2_IDX
#undef VP2_PREFETCH
#undef VP2_COMMIT_ROW
#undef VP2_COMMIT_ALL
}

} // namespace

int vp_check_timeout(VpContext* ctx, int stream);

int vp_sat_tiled(VpContext* ctx, int stream, const void* in, bool fused, int w, int h, int offset, int batch, int32_t* sat, float* graddot_out, const uint32_t* strips, int32_t* zero3) {
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
	a.zero = zero3;
	a.spin_limit = ctx->tune.sat_spin_limit > 0 ? (uint32_t)ctx->tune.sat_spin_limit : SPIN_LIMIT;
	a.drop_tile = ctx->tune.sat_debug_drop_tile;
	const int rawRows = TH + 2 * o;
	a.pitch = (rawRows * 2 * o + 3) & ~3; // halo array (dwords), 16-byte multiple
	const size_t lds = sizeof(int) * ((size_t)rawRows * TW + a.pitch + TH * TW + NWAVES * TW + TW + TH + 8);

	// persistent carry state (slot 1): [control: abort word + one 64-B ticket line per frame][tile order][V][H][K]
	const size_t bCtl = 64 * ((size_t)batch + 1);
	const size_t bOrd = (sizeof(uint32_t) * (size_t)a.tilesX * a.tilesY + 255) & ~size_t(255);
	const size_t bV = sizeof(u64) * (size_t)a.nTiles * TW, bH = sizeof(u64) * (size_t)a.nTiles * TH, bK = sizeof(u64) * (size_t)a.nTiles;
	void* base = nullptr;
	bool grew = false;
	VP_TRY(vp_workspace(ctx, stream, 1, bCtl + bOrd + bV + bH + bK, &base, &grew));
	if(!ctx->host_flag[stream]) {
		VP_REQUIRE(!ctx->capturing[stream], "first use of the single-pass SAT on this stream inside a graph capture");
		VP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->host_flag[stream]), 64, hipHostMallocMapped));
		*ctx->host_flag[stream] = 0;
	}
	const bool poisoned = *reinterpret_cast<volatile int*>(ctx->host_flag[stream]) != 0; // an earlier launch timed out (reported by vp_sync)
	// the per-frame ticket counters advance by tilesX*tilesY per launch: a different tiling or batch restarts the count
	// persistent grid: P workgroups per frame queue; every workgroup overshoots its queue by exactly two tickets (one workgroup per tile: one ticket each)
	const int perFrame = a.tilesX * a.tilesY;
	// resident workgroups per CU of k_sat_fast2 (vp_ctx_tune; 5 or 6 have their own register budget).  Small launches do better with fewer,
	// longer-lived workgroups (each tile's prefetch rides on the previous tile): 8 frames 80.0 -> 78.6 us, 32 frames 205 -> 196 us; no effect at 128
	const int wpc = ctx->tune.sat_workgroups_per_cu > 0 ? ctx->tune.sat_workgroups_per_cu : (batch < 64 ? 4 : 5);
	// k_sat_fast2 takes any width (the last tile column clamps / masks per pixel); rows need not be 16-byte aligned for buffer accesses
	const bool aligned16 = (w & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(sat) & 15) == 0 &&
			(!graddot_out || (reinterpret_cast<uintptr_t>(graddot_out) & 15) == 0);
	const bool fast = fused && o >= 1 && o <= 4 && ((w % TW) == 0 ? aligned16 : true) && (size_t)w * h * 4 < 0x7fffffffu && (size_t)2 * a.tilesX * h * o * 4 < 0x7fffffffu;
	int P = ((fast ? wpc : 4) * ctx->cu_count + batch - 1) / batch;
	if(P < 1) P = 1;
	if(P > perFrame) P = perFrame;
	a.single = fast && P >= perFrame ? 1 : 0;
	a.tickets_per_launch = a.single ? (uint32_t)P : (uint32_t)(perFrame + 2 * P);
	// The kernels tell launches apart by first ticket / tickets_per_launch (sat_launch_state): every launch since the last reset must have drawn
	// the same number of tickets per frame queue, so a different tiling, batch or grid restarts the count.  The host only keeps count of the
	// launches to restart before the 32-bit ticket counters (and the 30-bit epochs) wrap.
	const uint64_t shape = ((uint64_t)a.tilesX << 44) | ((uint64_t)a.tilesY << 24) | (uint64_t)batch;
	const bool reshaped = ctx->sat_shape[stream] != shape || ctx->sat_tickets[stream] != a.tickets_per_launch;
	const bool reset = grew || poisoned || reshaped || ctx->sat_epoch[stream] >= (1u << 30) - 2 || ((uint64_t)ctx->sat_epoch[stream] + 2) * a.tickets_per_launch >= (uint64_t(1) << 32);
	if(reset && ctx->capturing[stream]) {
		vp_set_error("the single-pass SAT state would have to be reset inside a graph capture"); // refused before any host state changes
		return VP_ERR_INVALID;
	}
	ctx->sat_shape[stream] = shape;
	ctx->sat_tickets[stream] = a.tickets_per_launch;
	if(reset) {
		if(!grew) {
			VP_CHECK_HIP(hipStreamSynchronize(s));
			VP_CHECK_HIP(hipMemsetAsync(base, 0, ctx->ws[1][stream].bytes, s));
		}
		*ctx->host_flag[stream] = 0;
		ctx->sat_epoch[stream] = 0;
	}
	++ctx->sat_epoch[stream]; // launches since the reset (this one included) = the epoch the kernel derives
	a.epoch = a.ticket_base = 0; // derived in the kernel
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
		// left neighbour one ticket ahead: measured ~3.7 re-polls per chain against 0.24).
		std::vector<uint32_t> ord;
		ord.reserve((size_t)a.tilesX * a.tilesY);
		for(int d = 0; d < a.tilesX + a.tilesY - 1; d++)
			for(int y = 0; y < a.tilesY; y++) {
				const int x = d - y;
				if(x >= 0 && x < a.tilesX) ord.push_back(((uint32_t)y << 16) | (uint32_t)x);
			}
		VP_CHECK_HIP(hipMemcpyAsync(p + bCtl, ord.data(), sizeof(uint32_t) * ord.size(), hipMemcpyHostToDevice, s));
		VP_CHECK_HIP(hipStreamSynchronize(s)); // ord is a temporary; this path runs only when the shape changes
	}
	int* dflag = nullptr;
	VP_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dflag), ctx->host_flag[stream], 0));
	a.host_flag = dflag;
	a.strips = vp_make_strips((fused && strips && vp_strips_supported(w, o)) ? const_cast<uint32_t*>(strips) : nullptr, vp_strips_supported(w, o) ? w : 64, h, o >= 1 ? o : 1);

	const unsigned grid = (unsigned)P * (unsigned)batch;
	if(fast) {
		const size_t lds2 = sizeof(int) * ((size_t)(TH + 2 * o) * (TW + 8) + 32 * TW + TH + TW + TH + 8);
		const uint32_t* fin = reinterpret_cast<const uint32_t*>(in);
		const bool cached_loads = (size_t)w * h * 4 * batch <= (size_t(112) << 20); // flat still sits in the Infinity Cache (see LDC)
#define VP_FAST2(OO) \
		do { \
			if((w % TW) != 0) { \
				if(graddot_out) k_sat_fast2<OO, 5, true, true><<<grid, 256, lds2, s>>>(fin, sat, graddot_out, a); \
				else k_sat_fast2<OO, 5, false, true><<<grid, 256, lds2, s>>>(fin, sat, graddot_out, a); \
			} else if(graddot_out) k_sat_fast2<OO, 5, true, false><<<grid, 256, lds2, s>>>(fin, sat, graddot_out, a); \
			else if(wpc == 6) k_sat_fast2<OO, 6, false, false><<<grid, 256, lds2, s>>>(fin, sat, graddot_out, a); \
			else if(cached_loads) k_sat_fast2<OO, 5, false, false, true><<<grid, 256, lds2, s>>>(fin, sat, graddot_out, a); \
			else k_sat_fast2<OO, 5, false, false><<<grid, 256, lds2, s>>>(fin, sat, graddot_out, a); \
		} while(0)
		if(o == 1) VP_FAST2(1);
		else if(o == 2) VP_FAST2(2);
		else if(o == 3) VP_FAST2(3);
		else VP_FAST2(4);
#undef VP_FAST2
	} else if(fused) {
		if(lds > 48 * 1024) VP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sat_tiled<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		k_sat_tiled<true><<<grid, 256, lds, s>>>(in, sat, graddot_out, a);
	} else {
		k_sat_tiled<false><<<grid, 256, lds, s>>>(in, sat, graddot_out, a);
	}
	return vp_launch_ok("tiled sat");
}
