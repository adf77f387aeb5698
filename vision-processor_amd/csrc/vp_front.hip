// Front of the chain: Bayer de-interleave and field-plane reprojection.
//   raw2quad   (raw2quad.cl:21-39, launched at Resources.cpp:142)
//   resampling (resampling.cl:29-100, launched at Resources.cpp:159)
//   + a fused form that reads the raw sensor buffer directly so the four planes never hit HBM.
//
// Numerics contract: field2image is evaluated in IEEE binary32, one rounding per source operation
// in the order of resampling.cl:29-47,53; this file is built with -ffp-contract=off and HIP's default correctly
// rounded fp32 division.  Integer-linear sampling follows VpSamplerMode (frozen in oracle/ref_shim.c).
#include "vp_common.h"
#include <type_traits>
#include <utility>
#include <algorithm>

// ---------------------------------------------------------------- raw2quad ----------------------------------------
// Bayer, vector path: one thread de-interleaves four 2x2 cells = 8 B of each of two sensor rows into 4 B per plane.
struct Planes4 {
	uint8_t* p[4];
	size_t bstride; // bytes between consecutive frames of one plane
};

__global__ __launch_bounds__(256) void k_raw2quad_bayer4(const uint8_t* __restrict__ raw, const Planes4 pl, int qw, int qh) {
	const int x4 = blockIdx.x * 64 + threadIdx.x;
	const int y = blockIdx.y * 4 + threadIdx.y;
	if(x4 * 4 >= qw || y >= qh) return;
	const size_t q = (size_t)qw * qh;
	const uint8_t* fr = raw + (size_t)blockIdx.z * 4 * q;
	const size_t row = 2 * (size_t)qw;
	const uint2 a = *reinterpret_cast<const uint2*>(fr + (size_t)(2 * y) * row + 8 * (size_t)x4);
	const uint2 b = *reinterpret_cast<const uint2*>(fr + (size_t)(2 * y + 1) * row + 8 * (size_t)x4);
	// v_perm_b32: selector bytes 0-3 pick from the 2nd operand, 4-7 from the 1st
	const uint32_t p0 = __builtin_amdgcn_perm(a.y, a.x, 0x06040200u), p1 = __builtin_amdgcn_perm(a.y, a.x, 0x07050301u);
	const uint32_t p2 = __builtin_amdgcn_perm(b.y, b.x, 0x06040200u), p3 = __builtin_amdgcn_perm(b.y, b.x, 0x07050301u);
	const size_t o = (size_t)blockIdx.z * pl.bstride + (size_t)y * qw + 4 * (size_t)x4;
	*reinterpret_cast<uint32_t*>(pl.p[0] + o) = p0;
	*reinterpret_cast<uint32_t*>(pl.p[1] + o) = p1;
	*reinterpret_cast<uint32_t*>(pl.p[2] + o) = p2;
	*reinterpret_cast<uint32_t*>(pl.p[3] + o) = p3;
}

// any width / BGR: one thread per cell
template<bool BGR>
__global__ __launch_bounds__(256) void k_raw2quad_scalar(const uint8_t* __restrict__ raw, const Planes4 pl, int qw, int qh) {
	const int x = blockIdx.x * 64 + threadIdx.x;
	const int y = blockIdx.y * 4 + threadIdx.y;
	if(x >= qw || y >= qh) return;
	const size_t q = (size_t)qw * qh;
	const size_t o = (size_t)blockIdx.z * pl.bstride + (size_t)y * qw + x;
	if(BGR) {
		const uint8_t* p = raw + (size_t)blockIdx.z * 3 * q + 3 * ((size_t)y * qw + x);
		pl.p[0][o] = p[0];
		pl.p[1][o] = p[1];
		pl.p[2][o] = p[2];
	} else {
		const size_t row = 2 * (size_t)qw;
		const uint8_t* p = raw + (size_t)blockIdx.z * 4 * q + (size_t)(2 * y) * row + 2 * (size_t)x;
		pl.p[0][o] = p[0];
		pl.p[1][o] = p[1];
		pl.p[2][o] = p[row];
		pl.p[3][o] = p[row + 1];
	}
}

static int raw2quad_launch(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, int batch, const Planes4& pl) {
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "raw2quad");
	hipStream_t s = ctx->streams[stream];
	const dim3 block(64, 4);
	bool al = (reinterpret_cast<uintptr_t>(raw) & 7) == 0 && (pl.bstride & 3) == 0;
	for(int c = 0; c < 4; c++) al = al && (reinterpret_cast<uintptr_t>(pl.p[c]) & 3) == 0;
	if(layout == VP_LAYOUT_BGR) {
		k_raw2quad_scalar<true><<<dim3(vp_div_up(qw, 64), vp_div_up(qh, 4), batch), block, 0, s>>>(raw, pl, qw, qh);
	} else if(qw % 4 == 0 && al) {
		k_raw2quad_bayer4<<<dim3(vp_div_up(qw / 4, 64), vp_div_up(qh, 4), batch), block, 0, s>>>(raw, pl, qw, qh);
	} else {
		k_raw2quad_scalar<false><<<dim3(vp_div_up(qw, 64), vp_div_up(qh, 4), batch), block, 0, s>>>(raw, pl, qw, qh);
	}
	return vp_launch_ok("raw2quad");
}

extern "C" int vp_raw2quad(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, int batch, uint8_t* planes) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(raw && planes && qw > 0 && qh > 0 && batch > 0 && layout >= 0 && layout <= 2, "vp_raw2quad: bad arguments");
	VP_REQUIRE(batch <= 65535, "vp_raw2quad: batch %d > 65535", batch);
	const size_t q = (size_t)qw * qh;
	const Planes4 pl = {{planes, planes + q, planes + 2 * q, planes + 3 * q}, 4 * q};
	return raw2quad_launch(ctx, stream, raw, layout, qw, qh, batch, pl);
}

extern "C" int vp_raw2quad4(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, uint8_t* plane0, uint8_t* plane1, uint8_t* plane2, uint8_t* plane3) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(raw && plane0 && plane1 && plane2 && (plane3 || layout == VP_LAYOUT_BGR) && qw > 0 && qh > 0 && layout >= 0 && layout <= 2, "vp_raw2quad4: bad arguments");
	const Planes4 pl = {{plane0, plane1, plane2, plane3 ? plane3 : plane2}, 0};
	return raw2quad_launch(ctx, stream, raw, layout, qw, qh, 1, pl);
}

// ---------------------------------------------------------------- resampling --------------------------------------
// While writing a flat pixel, also drop it into the halo strips if it sits within `o` pixels of a 64-pixel tile boundary.
__device__ __forceinline__ void emit_strip(const VpStrips& st, int frame, int gx, int gy, uint32_t px4) {
	if(!st.base) return;
	const int xm = gx & 63, j = gx >> 6;
	uint32_t* f = st.base + (size_t)frame * st.frame_words;
	if(xm < st.o) f[st.rec(j, gy, xm)] = px4;
	else if(xm >= 64 - st.o) f[st.rec(j, gy, xm - 64 + 2 * st.o)] = px4;
}

struct ResampleArgs {
	VpCameraModel m;
	float height, fs, ox, oy;
	int qw, qh, fw, fh;
	const uint8_t* pl[4]; // plane sources (unused when sampling the raw buffer)
	size_t pl_bstride;
	VpStrips strips;      // optional halo-strip output
};

// resampling.cl:29-47
__device__ __forceinline__ void field2image(const VpCameraModel& m, float fx, float fy, float fz, float& ox, float& oy) {
	fx = fx - m.c[0];
	fy = fy - m.c[1];
	fz = fz - m.c[2];
	const float cx = m.r[0] * fx + m.r[1] * fy + m.r[2] * fz;
	const float cy = m.r[3] * fx + m.r[4] * fy + m.r[5] * fz;
	const float cz = m.r[6] * fx + m.r[7] * fy + m.r[8] * fz;
	const float x2 = cx / cz, y2 = cy / cz;
	float ux = x2, uy = y2;
#pragma unroll
	for(int i = 0; i < 8; i++) {
		const float rx = ux * ux, ry = uy * uy;
		const float dr = 1.0f + m.d * (rx + ry);
		ux = x2 / dr;
		uy = y2 / dr;
	}
	ox = m.f * ux + m.p[0];
	oy = m.f * uy + m.p[1];
}

// one frame's source: the raw sensor buffer, or the four materialised quad planes
struct Src {
	const uint8_t* raw;
	const uint8_t* p[4];
};

// texel (i, j) of quad plane `ch`, either from materialised planes or straight from the raw sensor buffer
template<int LAYOUT, bool FROM_RAW>
__device__ __forceinline__ uint32_t texel(const Src& src, int ch, int i, int j, int qw, int qh) {
	if(!FROM_RAW) return src.p[ch][(size_t)j * qw + i];
	if(LAYOUT == VP_LAYOUT_BGR) return src.raw[3 * ((size_t)j * qw + i) + ch];
	return src.raw[(size_t)(2 * j + (ch >> 1)) * (2 * (size_t)qw) + 2 * (size_t)i + (ch & 1)]; // raw2quad.cl:31-37
}

template<int LAYOUT, int MODE, bool FROM_RAW>
__device__ __forceinline__ uint32_t sample(const Src& src, int ch, float u, float v, int qw, int qh) {
	if(MODE == VP_SAMPLER_NEAREST) {
		const int x = vp_clampi((int)floorf(u), 0, qw - 1), y = vp_clampi((int)floorf(v), 0, qh - 1);
		return texel<LAYOUT, FROM_RAW>(src, ch, x, y, qw, qh);
	}
	const float um = u - 0.5f, vm = v - 0.5f;
	const float fu = floorf(um), fv = floorf(vm);
	const float a = um - fu, b = vm - fv;
	const int i0 = vp_clampi((int)fu, 0, qw - 1), i1 = vp_clampi((int)fu + 1, 0, qw - 1);
	const int j0 = vp_clampi((int)fv, 0, qh - 1), j1 = vp_clampi((int)fv + 1, 0, qh - 1);
	const float t00 = (float)texel<LAYOUT, FROM_RAW>(src, ch, i0, j0, qw, qh), t10 = (float)texel<LAYOUT, FROM_RAW>(src, ch, i1, j0, qw, qh);
	const float t01 = (float)texel<LAYOUT, FROM_RAW>(src, ch, i0, j1, qw, qh), t11 = (float)texel<LAYOUT, FROM_RAW>(src, ch, i1, j1, qw, qh);
	const float oma = 1.0f - a, omb = 1.0f - b;
	const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
	float t = w00 * t00;
	t = t + w10 * t10;
	t = t + w01 * t01;
	t = t + w11 * t11;
	return (uint32_t)t;
}

// ---- Bayer fast sampling: 4 window loads instead of 16 byte gathers -------------------------------------------------------
// The four positional channels of a 2x2 Bayer cell are sampled at (u +- 0.25, v +- 0.25) (resampling.cl:64-79).  Every texel the
// four bilinear footprints touch lies in 4 raw rows (two even, two odd) and, horizontally, within 4 consecutive cells = 8
// bytes.  One unaligned 8-byte load per row replaces 16 single-byte gathers; the arithmetic on the texel values is unchanged.
typedef uint2 __attribute__((aligned(2))) uint2_a2;

struct Axis { // spec 8.2 linear filter along one axis: clamped indices and weight
	int i0, i1;
	float a;
};
__device__ __forceinline__ Axis axis_lin(float u, int n) {
	const float um = u - 0.5f, fu = floorf(um);
	Axis r;
	r.a = um - fu;
	r.i0 = vp_clampi((int)fu, 0, n - 1);
	r.i1 = vp_clampi((int)fu + 1, 0, n - 1);
	return r;
}
// same evaluation order as sample<>(): w00*t00 + w10*t10 + w01*t01 + w11*t11, truncated
__device__ __forceinline__ uint32_t bilin(float t00, float t10, float t01, float t11, float a, float b) {
	const float oma = 1.0f - a, omb = 1.0f - b;
	const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
	float t = w00 * t00;
	t = t + w10 * t10;
	t = t + w01 * t01;
	t = t + w11 * t11;
	return (uint32_t)t;
}
// Frame-independent part of the four Bayer footprints of one flat pixel: window offsets, byte selectors, weights.
struct Bayer4Geom {
	uint32_t off_e0, off_e1, off_o0, off_o1; // byte offsets of the four 8-byte windows inside a sensor frame
	uint32_t sa0, sa1, sb0, sb1;             // v_perm_b32 selectors of the two texels per axis (even / odd columns)
	float xa, xb, ya, yb;                    // interpolation weights along x (even / odd columns) and y (even / odd rows)
};
__device__ __forceinline__ Bayer4Geom bayer4_geom(float px, float py, int qw, int qh) {
	const Axis xa = axis_lin(px + 0.25f, qw), xb = axis_lin(px - 0.25f, qw);
	const Axis ya = axis_lin(py + 0.25f, qh), yb = axis_lin(py - 0.25f, qh);
	const int ibase = vp_clampi(min(xa.i0, xb.i0), 0, qw - 4);
	// 32-bit offsets from the (wave-uniform) frame base: one sensor frame is far below 4 GiB
	const uint32_t row = 2u * (uint32_t)qw, col = 2u * (uint32_t)ibase;
	Bayer4Geom g;
	g.off_e0 = (uint32_t)(2 * ya.i0) * row + col;
	g.off_e1 = (uint32_t)(2 * ya.i1) * row + col;
	g.off_o0 = (uint32_t)(2 * yb.i0 + 1) * row + col;
	g.off_o1 = (uint32_t)(2 * yb.i1 + 1) * row + col;
	g.sa0 = 0x0c0c0c00u | (uint32_t)(2 * (xa.i0 - ibase));
	g.sa1 = 0x0c0c0c00u | (uint32_t)(2 * (xa.i1 - ibase));
	g.sb0 = 0x0c0c0c00u | (uint32_t)(2 * (xb.i0 - ibase) + 1);
	g.sb1 = 0x0c0c0c00u | (uint32_t)(2 * (xb.i1 - ibase) + 1);
	g.xa = xa.a;
	g.xb = xb.a;
	g.ya = ya.a;
	g.yb = yb.a;
	return g;
}
__device__ __forceinline__ float win_sel(uint2 w, uint32_t sel) { return (float)__builtin_amdgcn_perm(w.y, w.x, sel); }
// p[0..3]: positional channels (even row/even col, even/odd, odd/even, odd/odd) of one frame, BILINEAR_RTZ
__device__ __forceinline__ void bayer4_combine(const uint2 e0, const uint2 e1, const uint2 o0, const uint2 o1, const Bayer4Geom& g, uint32_t p[4]);
__device__ __forceinline__ void bayer4_eval(const uint8_t* __restrict__ raw, const Bayer4Geom& g, uint32_t p[4]) {
	const uint2 e0 = *reinterpret_cast<const uint2_a2*>(raw + g.off_e0), e1 = *reinterpret_cast<const uint2_a2*>(raw + g.off_e1);
	const uint2 o0 = *reinterpret_cast<const uint2_a2*>(raw + g.off_o0), o1 = *reinterpret_cast<const uint2_a2*>(raw + g.off_o1);
	bayer4_combine(e0, e1, o0, o1, g, p);
}
__device__ __forceinline__ void bayer4_combine(const uint2 e0, const uint2 e1, const uint2 o0, const uint2 o1, const Bayer4Geom& g, uint32_t p[4]) {
	p[0] = bilin(win_sel(e0, g.sa0), win_sel(e0, g.sa1), win_sel(e1, g.sa0), win_sel(e1, g.sa1), g.xa, g.ya);
	p[1] = bilin(win_sel(e0, g.sb0), win_sel(e0, g.sb1), win_sel(e1, g.sb0), win_sel(e1, g.sb1), g.xb, g.ya);
	p[2] = bilin(win_sel(o0, g.sa0), win_sel(o0, g.sa1), win_sel(o1, g.sa0), win_sel(o1, g.sa1), g.xa, g.yb);
	p[3] = bilin(win_sel(o0, g.sb0), win_sel(o0, g.sb1), win_sel(o1, g.sb0), win_sel(o1, g.sb1), g.xb, g.yb);
}
__device__ __forceinline__ void sample_bayer4(const uint8_t* __restrict__ raw, float px, float py, int qw, int qh, uint32_t p[4]) {
	bayer4_eval(raw, bayer4_geom(px, py, qw, qh), p);
}
// delta-RGB in uint arithmetic (resampling.cl:86-91).  r, g, b are sampled bytes (<= 255), so every channel
// (2c - d - e + 510) / 4 lies in [0, 255] and the reference's convert_uchar saturation never acts; written as
// (3c + (510 - (r+g+b))) / 4 to share the sum.
__device__ __forceinline__ uint32_t drgb_pack(uint32_t r, uint32_t g, uint32_t b) {
	const uint32_t k = 510u - (r + g + b);
	const uint32_t dr = (__umul24(r, 3u) + k) >> 2, dg = (__umul24(g, 3u) + k) >> 2, db = (__umul24(b, 3u) + k) >> 2;
	return dr | (dg << 8) | (db << 16) | 0xFF000000u;
}

// dRGB + RGBA8 packing in 7 instructions instead of 11.  Each channel is (3c + k) >> 2 with k = 510 - (r+g+b), a byte (0 <= 2c - d - e + 510 <= 1020).
// Scaled by 64 the byte sits in bits 8..15: 64 (3c + k) = 192 c + (32640 - 64 s), one 24-bit multiply-add per channel on a common addend (32-bit
// wrap-around: the addend may be negative, the sum never is), and two v_perm_b32 gather byte 1 of the three words behind an 0xFF alpha.  Same integers
// as drgb_pack, no rounding involved.
__device__ __forceinline__ uint32_t drgb_pack7(uint32_t r, uint32_t g, uint32_t b) {
	const uint32_t s = r + g + b;
	const uint32_t k64 = 32640u - (s << 6);
	const uint32_t x = __umul24(r, 192u) + k64, y = __umul24(g, 192u) + k64, z = __umul24(b, 192u) + k64;
	const uint32_t xy = __builtin_amdgcn_perm(y, x, 0x0c0c0501u); // [x.b1, y.b1, 0, 0]
	return __builtin_amdgcn_perm(z, xy, 0x0d050100u);             // [x.b1, y.b1, z.b1, 0xFF]
}
typedef float vp_f2 __attribute__((ext_vector_type(2)));
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), written out
template<class F, int... I> __device__ __forceinline__ void vp_unroll_impl(F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>()), ...); }
template<int N, class F> __device__ __forceinline__ void vp_unroll(F& f) { vp_unroll_impl(f, std::make_integer_sequence<int, N>()); }
typedef float vp_f4 __attribute__((ext_vector_type(4)));

// Staged variant of the Bayer footprint: all texels of one sensor row that a flat pixel needs (two even-column and two
// odd-column bytes) lie within 5 consecutive bytes starting at an even column, hence inside TWO aligned dwords.  One
// v_perm_b32 per row gathers them, v_cvt_f32_ubyte0..3 converts them.
struct Bayer4Stage {
	int d0;                  // first dword (within the sensor row) of the two-dword window
	int re0, re1, ro0, ro1;  // sensor rows of the four windows
	uint32_t sel;            // perm selector: bytes [a.i0, a.i1, b.i0, b.i1]
	float xa, xb, ya, yb;
	int ia0, ia1, ib0, ib1;  // the four cell columns themselves (float-staged variant)
};
__device__ __forceinline__ Bayer4Stage bayer4_stage(float px, float py, int qw, int qh) {
	const Axis xa = axis_lin(px + 0.25f, qw), xb = axis_lin(px - 0.25f, qw);
	const Axis ya = axis_lin(py + 0.25f, qh), yb = axis_lin(py - 0.25f, qh);
	Bayer4Stage g;
	// leftmost byte needed is column 2*min(i0) (even), rightmost at most 4 bytes further; keep both dwords inside the row
	g.d0 = min((2 * min(xa.i0, xb.i0)) >> 2, (qw >> 1) - 2);
	const int b0 = 4 * g.d0;
	g.sel = (uint32_t)(2 * xa.i0 - b0) | ((uint32_t)(2 * xa.i1 - b0) << 8) | ((uint32_t)(2 * xb.i0 + 1 - b0) << 16) | ((uint32_t)(2 * xb.i1 + 1 - b0) << 24);
	g.ia0 = xa.i0;
	g.ia1 = xa.i1;
	g.ib0 = xb.i0;
	g.ib1 = xb.i1;
	g.re0 = 2 * ya.i0;
	g.re1 = 2 * ya.i1;
	g.ro0 = 2 * yb.i0 + 1;
	g.ro1 = 2 * yb.i1 + 1;
	g.xa = xa.a;
	g.xb = xb.a;
	g.ya = ya.a;
	g.yb = yb.a;
	return g;
}
// two rows (w0: y index 0, w1: y index 1), both column classes: returns the a-column and b-column samples
__device__ __forceinline__ void bayer4_rows(uint32_t w0, uint32_t w1, float xa, float xb, float y, uint32_t& pa, uint32_t& pb) {
	pa = bilin((float)(w0 & 0xFF), (float)((w0 >> 8) & 0xFF), (float)(w1 & 0xFF), (float)((w1 >> 8) & 0xFF), xa, y);
	pb = bilin((float)((w0 >> 16) & 0xFF), (float)(w0 >> 24), (float)((w1 >> 16) & 0xFF), (float)(w1 >> 24), xb, y);
}

template<int LAYOUT, int MODE, bool FROM_RAW>
__global__ __launch_bounds__(256) void k_resample(const uint8_t* __restrict__ src, uint8_t* __restrict__ flat, const ResampleArgs a, size_t src_frame_bytes) {
	const int gx = blockIdx.x * 64 + threadIdx.x;
	const int gy = blockIdx.y * 4 + threadIdx.y;
	if(gx >= a.fw || gy >= a.fh) return;
	Src s;
	s.raw = src + (size_t)blockIdx.z * src_frame_bytes;
#pragma unroll
	for(int c = 0; c < 4; c++) s.p[c] = a.pl[c] + (size_t)blockIdx.z * a.pl_bstride;
	float px, py;
	field2image(a.m, (float)gx * a.fs + a.ox, (float)gy * a.fs + a.oy, a.height, px, py); // resampling.cl:53
	uint32_t r, g, b;
	if(LAYOUT == VP_LAYOUT_BGR) { // :56-61
		r = sample<LAYOUT, MODE, FROM_RAW>(s, 2, px, py, a.qw, a.qh);
		g = sample<LAYOUT, MODE, FROM_RAW>(s, 1, px, py, a.qw, a.qh);
		b = sample<LAYOUT, MODE, FROM_RAW>(s, 0, px, py, a.qw, a.qh);
	} else if(FROM_RAW && MODE == VP_SAMPLER_BILINEAR_RTZ && a.qw >= 4) {
		uint32_t p[4];
		sample_bayer4(s.raw, px, py, a.qw, a.qh, p);
		if(LAYOUT == VP_LAYOUT_RGGB) { r = p[0]; g = p[1] / 2 + p[2] / 2; b = p[3]; }
		else { r = p[1]; g = p[0] / 2 + p[3] / 2; b = p[2]; }
	} else if(LAYOUT == VP_LAYOUT_RGGB) { // :64-71
		r = sample<LAYOUT, MODE, FROM_RAW>(s, 0, px + 0.25f, py + 0.25f, a.qw, a.qh);
		g = sample<LAYOUT, MODE, FROM_RAW>(s, 1, px - 0.25f, py + 0.25f, a.qw, a.qh) / 2 + sample<LAYOUT, MODE, FROM_RAW>(s, 2, px + 0.25f, py - 0.25f, a.qw, a.qh) / 2;
		b = sample<LAYOUT, MODE, FROM_RAW>(s, 3, px - 0.25f, py - 0.25f, a.qw, a.qh);
	} else { // GRBG :74-79
		r = sample<LAYOUT, MODE, FROM_RAW>(s, 1, px - 0.25f, py + 0.25f, a.qw, a.qh);
		g = sample<LAYOUT, MODE, FROM_RAW>(s, 0, px + 0.25f, py + 0.25f, a.qw, a.qh) / 2 + sample<LAYOUT, MODE, FROM_RAW>(s, 3, px - 0.25f, py - 0.25f, a.qw, a.qh) / 2;
		b = sample<LAYOUT, MODE, FROM_RAW>(s, 2, px + 0.25f, py - 0.25f, a.qw, a.qh);
	}
	const uint32_t px4 = drgb_pack(r, g, b); // :86-91
	reinterpret_cast<uint32_t*>(flat)[(size_t)blockIdx.z * a.fw * a.fh + (size_t)gy * a.fw + gx] = px4;
	emit_strip(a.strips, blockIdx.z, gx, gy, px4);
}

__global__ __launch_bounds__(256) void k_halo_strips(const uint32_t* __restrict__ flat, const VpStrips st, int w) {
	// one thread per (boundary, row, side, k)
	const int per_row = 2 * st.o;
	const size_t total = (size_t)st.tiles_x * st.h * per_row;
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if(i >= total) return;
	const int e = (int)(i % per_row), y = (int)((i / per_row) % st.h), j = (int)(i / ((size_t)per_row * st.h));
	const uint32_t* f = flat + (size_t)blockIdx.y * w * st.h;
	uint32_t* out = st.base + (size_t)blockIdx.y * st.frame_words;
	out[st.rec(j, y, e)] = f[(size_t)y * w + 64 * j + (e < st.o ? e : 64 - 2 * st.o + e)];
}

extern "C" size_t vp_halo_strip_bytes(int w, int h, int offset) {
	if(!vp_strips_supported(w, offset) || h <= 0) return 0;
	return sizeof(uint32_t) * vp_make_strips(nullptr, w, h, offset).frame_words;
}

extern "C" int vp_halo_strips(VpContext* ctx, int stream, const uint8_t* flat, int w, int h, int offset, int batch, uint32_t* strips) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(flat && strips && h > 0 && batch > 0 && batch <= 65535 && vp_strips_supported(w, offset), "vp_halo_strips: needs w %% 64 == 0 and 1 <= offset <= 4");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "halo strips");
	const VpStrips st = vp_make_strips(strips, w, h, offset);
	const size_t total = (size_t)st.tiles_x * h * 2 * offset;
	if(total == 0) return VP_OK;
	k_halo_strips<<<dim3((unsigned)((total + 255) / 256), batch), 256, 0, ctx->streams[stream]>>>(reinterpret_cast<const uint32_t*>(flat), st, w);
	return vp_launch_ok("halo strips");
}

// ---- coordinate map ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resample_map(float2* __restrict__ map, const ResampleArgs a) {
	const int gx = blockIdx.x * 64 + threadIdx.x;
	const int gy = blockIdx.y * 4 + threadIdx.y;
	if(gx >= a.fw || gy >= a.fh) return;
	float px, py;
	field2image(a.m, (float)gx * a.fs + a.ox, (float)gy * a.fs + a.oy, a.height, px, py);
	map[(size_t)gy * a.fw + gx] = make_float2(px, py);
}

// FPT frames of the batch per thread: the coordinate and everything derived from it (tap offsets, selectors, weights) are
// computed once and reused, so the 8 B/px map is read once per FPT frames instead of once per frame.
constexpr int RESAMPLE_FPT = 16; // measured on the 128-frame workload: 8 -> 853 us, 16 -> 800 us, 32 -> 867 us (LDS stage 16 / 16 / 32 KB per workgroup)

// The 256 pixels of a workgroup (64 x 4 of flat) need only about a kilobyte of distinct sensor bytes per frame, yet every
// lane fetching its own four 8-byte windows keeps the texture addresser busy 85 % of the kernel (~36 TA cycles per unaligned
// wave load).  So the workgroup finds the bounding box of its windows once (frame independent), then per frame stages that box
// in LDS with ONE coalesced dword load per thread (double buffered, one barrier per frame) and reads the windows from LDS.
// Boxes too large for the stage (exotic geometries) take the direct path; results are identical either way.
constexpr int STAGE_DW = 256; // dwords per frame box in the stage

// packed-u16 min / max (v_pk_min_u16 / v_pk_max_u16) and their wave reductions in DPP (row_shr 1,2,4,8, row_bcast 15 and 31; the
// result is read from lane 63)
typedef unsigned short vp_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk16_min(uint32_t a, uint32_t b) {
	return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(vp_u16x2, a), __builtin_bit_cast(vp_u16x2, b)));
}
__device__ __forceinline__ uint32_t pk16_max(uint32_t a, uint32_t b) {
	return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(vp_u16x2, a), __builtin_bit_cast(vp_u16x2, b)));
}
template<bool MAX>
__device__ __forceinline__ uint32_t wave_reduce_pk16(uint32_t v) {
#define VP_DPP_STEP(ctrl, rowmask) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rowmask, 0xf, false); v = MAX ? pk16_max(v, t) : pk16_min(v, t); }
	VP_DPP_STEP(0x111, 0xf) VP_DPP_STEP(0x112, 0xf) VP_DPP_STEP(0x114, 0xf) VP_DPP_STEP(0x118, 0xf) VP_DPP_STEP(0x142, 0xa) VP_DPP_STEP(0x143, 0xc)
#undef VP_DPP_STEP
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Bayer + BILINEAR_RTZ, sensor rows dword aligned (host-checked): the staged kernel.  Workgroups whose box exceeds the stage
// sample directly (bayer4_eval); results are identical either way.
// VARIANT 0: BILINEAR_RTZ, box staged as bytes; 1: BILINEAR_RTZ, box staged as floats; 2: NEAREST, box staged as bytes; 3 (default): BILINEAR_RTZ, box staged as
// floats of two frames side by side (run_fp) where the workgroup's footprints allow it, as bytes elsewhere (the tiles on the left / right image border).
template<int LAYOUT, int VARIANT, bool NTS>
__global__ __launch_bounds__(256) void k_resample_staged(const uint8_t* __restrict__ raw, const float2* __restrict__ map, uint8_t* __restrict__ flat,
		int qw, int qh, int fw, int fh, size_t src_frame_bytes, const VpStrips strips, int batch, int xcd_map, int groups_per_wg) {
	__shared__ __attribute__((aligned(16))) uint32_t stage[RESAMPLE_FPT][STAGE_DW];
	__shared__ __attribute__((aligned(16))) uint32_t fp_rows[2][2][4][64]; // run_fp: the workgroup's 4 x 64 flat pixels of the two frames of a step (the strip records are gathered from them), double buffered
	uint32_t (*red)[2] = reinterpret_cast<uint32_t (*)[2]>(&fp_rows[0][0][0][0]); // the set-up's box reduction (done with before anything is parked): 20 480 B of LDS per workgroup, eight per CU
	const unsigned bz = blockIdx.z;
	// Workgroups are dealt round-robin over the 8 XCDs (private L2 each).  A workgroup's box shares its 64/128-byte lines with the
	// boxes left and right of it and two to four sensor rows with those above and below; with the plain mapping horizontal
	// neighbours always sit on different XCDs and every line is fetched from HBM by each of them.  xcd_map hands every XCD a
	// contiguous band of the frame group's tiles instead (placement only: results are identical).
	unsigned bxi = blockIdx.x, byi = blockIdx.y;
	if(xcd_map) {
		const unsigned l = vp_xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
		byi = l / gridDim.x;
		bxi = l - byi * gridDim.x;
	}
	const int gx = (int)bxi * 64 + threadIdx.x;
	const int gy = (int)byi * 4 + threadIdx.y;
	const bool valid = gx < fw && gy < fh;
	const float2 pos = map[(size_t)min(gy, fh - 1) * fw + min(gx, fw - 1)];
	const float px = pos.x, py = pos.y;
	const size_t npx = (size_t)fw * fh;
	const uint32_t rowb = 2u * (uint32_t)qw;
	{
		constexpr bool FLT = VARIANT == 1, NEAR = VARIANT == 2, FP = VARIANT == 3;
		Bayer4Stage g = {};
		// NEAREST (spec 8.2: texel = floor(u), clamp-to-edge): one texel per positional channel -- even row e / odd row o of the
		// sensor, even byte column ca / odd byte column cb (resampling.cl:64-79 with the +-0.25 offsets)
		int nca = 0, ncb = 0, ne = 0, no = 0;
		if(NEAR) {
			nca = 2 * vp_clampi((int)floorf(px + 0.25f), 0, qw - 1);
			ncb = 2 * vp_clampi((int)floorf(px - 0.25f), 0, qw - 1) + 1;
			ne = 2 * vp_clampi((int)floorf(py + 0.25f), 0, qh - 1);
			no = 2 * vp_clampi((int)floorf(py - 0.25f), 0, qh - 1) + 1;
		} else g = bayer4_stage(px, py, qw, qh);
		// bounding box of this workgroup's windows: dword columns [cmin, cmax), sensor rows [rmin, rmax]
		// (both fit 16 bits: checked by the host; one packed 16-bit min and max reduction over the workgroup, DPP within a wave)
		const uint32_t lo_w = wave_reduce_pk16<false>(NEAR ? (uint32_t)(min(nca, ncb) >> 2) | ((uint32_t)min(ne, no) << 16) : (uint32_t)g.d0 | ((uint32_t)min(g.re0, g.ro0) << 16));
		const uint32_t hi_w = wave_reduce_pk16<true>(NEAR ? (uint32_t)((max(nca, ncb) >> 2) + 1) | ((uint32_t)max(ne, no) << 16) : (uint32_t)(g.d0 + 2) | ((uint32_t)max(g.re1, g.ro1) << 16));
		const int tid = threadIdx.y * 64 + threadIdx.x;
		if(threadIdx.x == 0) {
			red[threadIdx.y][0] = lo_w;
			red[threadIdx.y][1] = hi_w;
		}
		__syncthreads();
		// (workgroup-uniform, and said to be: everything derived from the box lives in SGPRs and branches on it are scalar)
		const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)pk16_min(pk16_min(red[0][0], red[1][0]), pk16_min(red[2][0], red[3][0])));
		const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)pk16_max(pk16_max(red[0][1], red[1][1]), pk16_max(red[2][1], red[3][1])));
		const int cmin = (int)(lo & 0xFFFF), rmin = (int)(lo >> 16), cmax = (int)(hi & 0xFFFF), rmax = (int)(hi >> 16);
		const int wdw = cmax - cmin, hr = rmax - rmin + 1, ndw = wdw * hr;
		// The workgroup's first frame group (up to RESAMPLE_FPT frames [f0, f1)); run_fp goes on to the next groups_per_wg - 1 by itself.
		auto group = [&](const int f0, const int f1) {
		const size_t raw_span = (size_t)(f1 - f0) * src_frame_bytes, flat_span = (size_t)(f1 - f0) * npx * 4;
		if((FP ? hr * (wdw + 2) : ndw) <= STAGE_DW && raw_span < 0x7fffffffu && flat_span < 0x7fffffffu && (size_t)(f1 - f0) * strips.frame_words * 4 < 0x7fffffffu) {
			// buffer addressing: descriptor in SGPRs (rebased to this frame group), 32-bit per-lane offset, per-frame scalar
			// offset -- no 64-bit VALU address arithmetic inside the frame loop
			const __amdgpu_buffer_rsrc_t rs_raw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(raw) + (size_t)f0 * src_frame_bytes, 0, (int)raw_span, 0x00020000);
			const __amdgpu_buffer_rsrc_t rs_flat = __builtin_amdgcn_make_buffer_rsrc(flat + (size_t)f0 * npx * 4, 0, (int)flat_span, 0x00020000);
			const __amdgpu_buffer_rsrc_t rs_strip = __builtin_amdgcn_make_buffer_rsrc(strips.base ? strips.base + (size_t)f0 * strips.frame_words : reinterpret_cast<uint32_t*>(flat), 0,
					(int)((size_t)(f1 - f0) * strips.frame_words * 4), 0x00020000);
			const uint32_t fb_raw = (uint32_t)src_frame_bytes, fb_flat = (uint32_t)(npx * 4), fb_strip = (uint32_t)(strips.frame_words * 4);
			const uint32_t voff_out = 4u * (uint32_t)(min(gy, fh - 1) * fw + min(gx, fw - 1));
			// NTS: flat goes out non-temporal when the launch's flat is far larger than the Infinity Cache (the SAT pass will read it from HBM anyway, and what
			// the resampling kernel leaves dirty in the caches is written back during that pass): - 1.4 % per 128-frame vp_frame, - 1.5 % at 64; cached for
			// launches that fit (the SAT pass then reads flat from the cache: nt costs + 2.4..8 % at 8 frames).  A template parameter: as a run-time branch
			// it cost more than half of what it gains (profiles/r04_sat_cache_policy.json).
			auto store_flat = [&](uint32_t px4, uint32_t soff) { __builtin_amdgcn_raw_buffer_store_b32(px4, rs_flat, voff_out, soff, NTS ? 2 : 0); };
			// halo-strip destinations of this pixel (emit_strip(), hoisted out of the frame loop)
			const int xm = gx & 63, tb = gx >> 6;
			const uint32_t so = (uint32_t)strips.o, shp = (uint32_t)strips.hp; // VpStrips::rec in 32-bit arithmetic (frame_words * 4 < 2 GiB)
			const bool se = valid && strips.base && (xm < strips.o || xm >= 64 - strips.o); // this pixel is one of the first or last o pixels of its tile row
			const uint32_t se_e = (uint32_t)(xm < strips.o ? xm : xm - 64 + 2 * strips.o);  // position inside the row's record
			const uint32_t voff_se = se ? 4u * (((uint32_t)tb * shp + (uint32_t)gy) * 2u * so + se_e) : 0u;
			const int wcol = g.d0 - cmin;
			const int ie0 = (g.re0 - rmin) * wdw + wcol, ie1 = (g.re1 - rmin) * wdw + wcol, io0 = (g.ro0 - rmin) * wdw + wcol, io1 = (g.ro1 - rmin) * wdw + wcol;
			// which dwords of the box this thread fetches, NP pieces of 256 (lanes past the box re-fetch its first dword: harmless)
			const float inv_wdw = __builtin_amdgcn_rcpf((float)wdw);
			const int nf = f1 - f0;
			// On gfx9-family hardware loads and stores share one vmcnt and complete out of order with respect to each other, so a
			// wave that waits for a load while one of its stores is in flight drains both (measured: +185 us per 128 frames when
			// the box of frame f+1 was fetched while frame f was being stored).  Hence: fetch the boxes of ALL frames of the group
			// first (one wait, no store pending), park them in LDS, one barrier, then sample and store frame after frame with no
			// further memory waits.  Other workgroups on the CU cover the fetch phase.
			auto run = [&](auto np) {
				constexpr int NP = decltype(np)::value; // pieces of 256 dwords per frame box
				uint32_t goff[NP], pre[RESAMPLE_FPT][NP];
#pragma unroll
				for(int k = 0; k < NP; k++) {
					// r = i / wdw without the integer-division sequence: (i + 0.5) / wdw stays >= 0.5 / wdw away from every integer while
					// the product's error is below 1e-6 * STAGE_DW / wdw.  Lanes past the box re-fetch its first dword (harmless).
					const int i = tid + 256 * k < ndw ? tid + 256 * k : 0, r = (int)(((float)i + 0.5f) * inv_wdw), c = i - r * wdw;
					goff[k] = (uint32_t)(rmin + r) * rowb + 4u * (uint32_t)(cmin + c);
				}
#pragma unroll
				for(int j = 0; j < RESAMPLE_FPT; j++)
#pragma unroll
					for(int k = 0; k < NP; k++) pre[j][k] = __builtin_amdgcn_raw_buffer_load_b32(rs_raw, goff[k], (uint32_t)min(j, nf - 1) * fb_raw, 0);
#pragma unroll
				for(int j = 0; j < RESAMPLE_FPT; j++)
#pragma unroll
					for(int k = 0; k < NP; k++) stage[j][tid + 256 * k] = pre[j][k];
				__syncthreads();
#pragma unroll
				for(int j = 0; j < RESAMPLE_FPT; j++) {
					if(j >= nf) break;
					const uint32_t* st = stage[j];
					auto window = [&](int i) { return __builtin_amdgcn_perm(st[i + 1], st[i], g.sel); };
					uint32_t p[4];
					bayer4_rows(window(ie0), window(ie1), g.xa, g.xb, g.ya, p[0], p[1]);
					bayer4_rows(window(io0), window(io1), g.xa, g.xb, g.yb, p[2], p[3]);
					const uint32_t px4 = LAYOUT == VP_LAYOUT_RGGB ? drgb_pack(p[0], p[1] / 2 + p[2] / 2, p[3]) : drgb_pack(p[1], p[0] / 2 + p[3] / 2, p[2]);
					// (lanes past the right/bottom edge hold the clamped pixel's own value: the duplicate store is harmless)
					store_flat(px4, (uint32_t)j * fb_flat);
					if(se) __builtin_amdgcn_raw_buffer_store_b32(px4, rs_strip, voff_se, (uint32_t)j * fb_strip, 0);
				}
			};
			// FLT: the box is staged as FLOATS, one frame at a time (double buffered, one barrier per frame): every texel is
			// converted once by the thread that fetched it (4 converts per thread and frame) instead of once per use (16 per pixel
			// and frame, at the 4-cycle rate).  Even sensor columns go to plane A, odd ones to plane B of a box row, so the two
			// texels of a footprint row are neighbours.  Loads of all frames are still issued up front.
			auto run_flt = [&]() {
				float* F = reinterpret_cast<float*>(&stage[0][0]); // [2][1024]
				const int own = tid < ndw ? tid : 0, orow = (int)(((float)own + 0.5f) * inv_wdw), ocol = own - orow * wdw;
				const uint32_t goff = (uint32_t)(rmin + orow) * rowb + 4u * (uint32_t)(cmin + ocol);
				const int RP = 4 * wdw; // floats per box row: [A: 2 wdw | B: 2 wdw]
				float* wa = F + orow * RP + 2 * ocol;
				float* wb = wa + 2 * wdw;
				uint32_t pre[RESAMPLE_FPT];
#pragma unroll
				for(int j = 0; j < RESAMPLE_FPT; j++) pre[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_raw, goff, (uint32_t)min(j, nf - 1) * fb_raw, 0);
				auto put = [&](int buf, uint32_t v) {
					if(tid < ndw) {
						*reinterpret_cast<float2*>(wa + buf * 1024) = make_float2((float)(v & 0xFF), (float)((v >> 16) & 0xFF));
						*reinterpret_cast<float2*>(wb + buf * 1024) = make_float2((float)((v >> 8) & 0xFF), (float)(v >> 24));
					}
				};
				// float indices of the 16 texels (frame independent): plane A column of cell i is i - 2 cmin, plane B likewise + 2 wdw
				const int ca0 = g.ia0 - 2 * cmin, ca1 = g.ia1 - 2 * cmin, cb0 = g.ib0 - 2 * cmin + 2 * wdw, cb1 = g.ib1 - 2 * cmin + 2 * wdw;
				const int r0 = (g.re0 - rmin) * RP, r1 = (g.re1 - rmin) * RP, r2 = (g.ro0 - rmin) * RP, r3 = (g.ro1 - rmin) * RP;
				const float* t00 = F + r0 + ca0; const float* t01 = F + r0 + ca1; const float* t02 = F + r1 + ca0; const float* t03 = F + r1 + ca1;
				const float* t10 = F + r0 + cb0; const float* t11 = F + r0 + cb1; const float* t12 = F + r1 + cb0; const float* t13 = F + r1 + cb1;
				const float* t20 = F + r2 + ca0; const float* t21 = F + r2 + ca1; const float* t22 = F + r3 + ca0; const float* t23 = F + r3 + ca1;
				const float* t30 = F + r2 + cb0; const float* t31 = F + r2 + cb1; const float* t32 = F + r3 + cb0; const float* t33 = F + r3 + cb1;
				// Halo strips: the 8 o strip pixels of this workgroup and frame (4 rows x [first o | last o]) are one contiguous record.  Instead
				// of a masked 4-lane store per wave and frame (two 8-byte pieces each: as many write transactions as the flat stores, measured
				// +85 us per 128 frames) the pixels are parked in LDS and ONE wave stores the record of the previous frame as a whole.
				uint32_t* sbuf = &stage[8][0]; // [2][32] behind the two float boxes (8 KB)
				const bool agg = strips.base != nullptr;
				const uint32_t voff_rec = 4u * (((uint32_t)bxi * shp + (uint32_t)byi * 4u) * 2u * so + (uint32_t)threadIdx.x);
				const bool rec_lane = threadIdx.x < 8 * strips.o;
				uint32_t* sslot = se ? sbuf + threadIdx.y * 2 * so + se_e : &stage[9][tid]; // pixels off the strips go to a dump slot of their own (no branch in the loop)
				const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
				auto flush = [&](int j) { // frame j's record, by wave j % 4 (after the barrier that ended iteration j)
					if(agg && wave == (j & 3) && rec_lane)
						__builtin_amdgcn_raw_buffer_store_b32(sbuf[(j & 1) * 32 + threadIdx.x], rs_strip, voff_rec, (uint32_t)j * fb_strip, 0);
				};
				put(0, pre[0]);
				__syncthreads();
#pragma unroll
				for(int j = 0; j < RESAMPLE_FPT; j++) {
					if(j >= nf) break;
					const int o = (j & 1) * 1024; // compile-time after unrolling: folds into the ds_read offset field
					uint32_t p[4];
					p[0] = bilin(t00[o], t01[o], t02[o], t03[o], g.xa, g.ya);
					p[1] = bilin(t10[o], t11[o], t12[o], t13[o], g.xb, g.ya);
					p[2] = bilin(t20[o], t21[o], t22[o], t23[o], g.xa, g.yb);
					p[3] = bilin(t30[o], t31[o], t32[o], t33[o], g.xb, g.yb);
					if(j + 1 < nf) put((j + 1) & 1, pre[j + 1 < RESAMPLE_FPT ? j + 1 : j]);
					const uint32_t px4 = LAYOUT == VP_LAYOUT_RGGB ? drgb_pack(p[0], p[1] / 2 + p[2] / 2, p[3]) : drgb_pack(p[1], p[0] / 2 + p[3] / 2, p[2]);
					store_flat(px4, (uint32_t)j * fb_flat);
					sslot[(j & 1) * 32] = px4;
					if(j > 0) flush(j - 1);
					__syncthreads(); // frame j+1 is staged; everybody is done reading frame j's buffer before frame j+2 overwrites it
				}
				flush(nf - 1);
			};
			// FRAME-PAIR form of run_flt (VARIANT 3, the default).  The float-staged kernel is bound by LDS cycles, not by the VALU (taking a quarter of its VALU
			// instructions out changes nothing, adding LDS stores slows it down: profiles/r05_resample_lds_bound.json): 16 ds_read_b32 per pixel and frame at 2
			// cycles per wave instruction -- and a ds_read_b64 costs the same 2 cycles (MI355X_MICROARCH.md, LDS).  So the LDS image holds, for every cell of the
			// box, the texel of TWO consecutive frames side by side (8 bytes, always aligned: an 8-byte access off its alignment is replayed at 64 cycles), and a
			// pixel computes both frames at once: the weights are the same, so every product and every partial sum is one packed instruction for the two frames
			// (v_pk_mul_f32 taking one half of a weight pair for both lanes, v_pk_add_f32), in the order of the sampler definition.  Per pixel and frame: 8 LDS
			// reads instead of 16, 14 instead of 17 instructions for the four sums, half a barrier; per staged dword one ds_write_b128 per plane and frame PAIR.
			// [First built with CELL pairs (T[k], T[k+1]) per frame: every texel stored twice, the neighbour's dword fetched by a DPP wave shift, whole box rows per
			// staging wavefront; 1 % slower than this and more conditions.]
			// The second column's texel is read at + 8 bytes from the first one's address register, so a footprint must be two NEIGHBOURING cells -- also where
			// clamp-to-edge folds it onto one (sample<>(): i0 = clamp(fu), i1 = clamp(fu + 1)): a box row carries two spare cells on either side, and a box that
			// touches the left / right sensor border gets a copy of cell 0 in virtual cell -1 / of cell qw - 1 in virtual cell qw.  The pair (i0v, i0v + 1) with
			// i0v = clamp(fu, -1, qw - 1) then reads exactly (T[i0], T[i1]) for every fu.
			// The loop over the workgroup's frame groups sits BELOW the set-up (coordinate, footprints, box, weights, addresses: about a third of the kernel's VALU
			// instructions at one group per workgroup), which is paid once for all of them.
			auto run_fp = [&]() {
				// FPS frames per sub-group (their boxes are requested together: one register per lane and frame), NSETS register sets taking turns, NSUB sub-groups
				// in the written-out stream (profiles/r05_resample_lds_bound.json has the sweep: 16 frames x 1 set in a loop 1311 us per 256-frame launch, 8 x 1: 1271,
				// 4 x 2 sets written out: 1195)
#ifndef VP_RS_FPS
#define VP_RS_FPS 4
#endif
#ifndef VP_RS_NSETS
#define VP_RS_NSETS 2
#endif
#ifndef VP_RS_NSUB
#define VP_RS_NSUB 16 // sub-groups of the written-out (FULL) path: groups_per_wg x RESAMPLE_FPT / FPS must be <= it (4 groups of 16 frames); a workgroup with fewer leaves the stream early
#endif
				constexpr int FPS = VP_RS_FPS;
				static_assert(RESAMPLE_FPT % FPS == 0 && FPS % 2 == 0, "sub-groups of whole frame pairs");
				// the -D overrides were a sweep (profiles/r05_resample_lds_bound.json); what the pair stream supports: a set is requested NSETS - 1 sub-groups ahead of
				// its use and must not be the one in use, and a sub-group must span at least two pairs for the request to go out ahead of its first store
				static_assert(VP_RS_NSETS >= 2 && VP_RS_NSETS <= VP_RS_NSUB && FPS >= 4, "unsupported VP_RS_FPS / VP_RS_NSETS / VP_RS_NSUB combination");
				float* F = reinterpret_cast<float*>(&stage[0][0]); // [2][2048]: per box row [plane A: (2 wdw + 4) cells x (frame f, frame f + 1) | plane B likewise]
				const int PL = 4 * wdw + 8, RP = 2 * PL;            // floats per plane of a box row, per box row
				const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
				const int own = tid < ndw ? tid : 0, orow = (int)(((float)own + 0.5f) * inv_wdw), ocol = own - orow * wdw;
				const uint32_t goff = (uint32_t)(rmin + orow) * rowb + 4u * (uint32_t)(cmin + ocol);
				float* wa = F + orow * RP + 4 * ocol + 4; // real cells 2 ocol, 2 ocol + 1 of plane A (behind the two spare cells); plane B PL floats on
				const bool pad_l = cmin == 0, pad_r = 2 * cmax == qw;                     // workgroup-uniform: the box touches the left / right sensor border
				const bool own_l = pad_l && tid < ndw && ocol == 0, own_r = pad_r && tid < ndw && ocol == wdw - 1; // the threads that hold cell 0 / cell qw - 1 of a box row at the border
				const bool pad_any = pad_l || pad_r;
				// ONE predicated region for the two 16-byte stores, and a scalar branch around the border copies (taken by every workgroup that does not touch the
				// sensor's left / right border): the step's instruction stream is what a wavefront's time goes into (profiles/r05_resample_issue_stream.json)
				auto put = [&](int buf, uint32_t v0, uint32_t v1) { // frames f (v0) and f + 1 (v1) of this thread's dword
					float* w = wa + buf * 2048;
					// plane A: even sensor columns, plane B: odd ones
					const float a0 = (float)(v0 & 0xFF), a1 = (float)(v1 & 0xFF), c0 = (float)((v0 >> 16) & 0xFF), c1 = (float)((v1 >> 16) & 0xFF);
					const float b0 = (float)((v0 >> 8) & 0xFF), b1 = (float)((v1 >> 8) & 0xFF), d0 = (float)(v0 >> 24), d1 = (float)(v1 >> 24);
					if(tid < ndw) {
						*reinterpret_cast<vp_f4*>(w) = vp_f4{a0, a1, c0, c1};
						*reinterpret_cast<vp_f4*>(w + PL) = vp_f4{b0, b1, d0, d1};
					}
					if(__builtin_expect(pad_any, 0)) {
						if(own_l) { // virtual cell -1 := cell 0
							*reinterpret_cast<vp_f2*>(w - 2) = vp_f2{a0, a1};
							*reinterpret_cast<vp_f2*>(w + PL - 2) = vp_f2{b0, b1};
						}
						if(own_r) { // virtual cell qw := cell qw - 1
							*reinterpret_cast<vp_f2*>(w + 4) = vp_f2{c0, c1};
							*reinterpret_cast<vp_f2*>(w + PL + 4) = vp_f2{d0, d1};
						}
					}
				};
				// the first cell of a footprint in virtual coordinates: the clamped pair (0, 0) is (-1, 0), the clamped pair (qw - 1, qw - 1) is (qw - 1, qw)
				const int ia = g.ia0 == g.ia1 && g.ia0 == 0 ? -1 : g.ia0, ib = g.ib0 == g.ib1 && g.ib0 == 0 ? -1 : g.ib0;
				// cell i of plane A sits at float 2 (i - 2 cmin + 2) of its box row, plane B PL floats on; the footprint's second column is the next cell ([1])
				const int pa = 2 * (ia - 2 * cmin + 2), pb = 2 * (ib - 2 * cmin + 2) + PL;
				const int r0 = (g.re0 - rmin) * RP, r1 = (g.re1 - rmin) * RP, r2 = (g.ro0 - rmin) * RP, r3 = (g.ro1 - rmin) * RP;
				const vp_f2* q00 = reinterpret_cast<const vp_f2*>(F + r0 + pa); const vp_f2* q01 = reinterpret_cast<const vp_f2*>(F + r1 + pa);
				const vp_f2* q10 = reinterpret_cast<const vp_f2*>(F + r0 + pb); const vp_f2* q11 = reinterpret_cast<const vp_f2*>(F + r1 + pb);
				const vp_f2* q20 = reinterpret_cast<const vp_f2*>(F + r2 + pa); const vp_f2* q21 = reinterpret_cast<const vp_f2*>(F + r3 + pa);
				const vp_f2* q30 = reinterpret_cast<const vp_f2*>(F + r2 + pb); const vp_f2* q31 = reinterpret_cast<const vp_f2*>(F + r3 + pb);
				// the 16 weights; the two greens' carry a factor 0.5, so their sums come out halved as resampling.cl:66,76 halves them: scaling a weight by 0.5 scales every
				// product and partial sum EXACTLY (a weight is 0 or >= 8e-16: nothing underflows), and (uint)(t / 2) = (uint)t / 2 for t >= 0
				const float oxa = 1.0f - g.xa, oxb = 1.0f - g.xb, oya = 1.0f - g.ya, oyb = 1.0f - g.yb;
				const float h0 = LAYOUT == VP_LAYOUT_RGGB ? 1.0f : 0.5f, h1 = LAYOUT == VP_LAYOUT_RGGB ? 0.5f : 1.0f, h2 = h1, h3 = h0;
				// kept as register PAIRS (w00, w10), (w01, w11) per channel: a packed multiply takes one half of a pair for both of its lanes (op_sel), so the 16
				// weights cost 16 registers, not 32 broadcast copies
				const vp_f2 w0a = vp_f2{(oxa * oya) * h0, (g.xa * oya) * h0}, w0b = vp_f2{(oxa * g.ya) * h0, (g.xa * g.ya) * h0};
				const vp_f2 w1a = vp_f2{(oxb * oya) * h1, (g.xb * oya) * h1}, w1b = vp_f2{(oxb * g.ya) * h1, (g.xb * g.ya) * h1};
				const vp_f2 w2a = vp_f2{(oxa * oyb) * h2, (g.xa * oyb) * h2}, w2b = vp_f2{(oxa * g.yb) * h2, (g.xa * g.yb) * h2};
				const vp_f2 w3a = vp_f2{(oxb * oyb) * h3, (g.xb * oyb) * h3}, w3b = vp_f2{(oxb * g.yb) * h3, (g.xb * g.yb) * h3};
				// (volatile: two 8-byte reads 8 bytes apart must NOT be merged into one ds_read2_b64 -- that costs 8 LDS cycles against 2 + 2, MI355X_MICROARCH.md, LDS)
				auto sum2 = [&](vp_f2 wa_, vp_f2 wb_, const vp_f2* row0, const vp_f2* row1, int o, uint32_t& a, uint32_t& b) { // both frames (a, b): (uint)(w00 t00 + w10 t10 + w01 t01 + w11 t11), in that order
					typedef __attribute__((address_space(3))) const volatile vp_f2 lds_f2;
					lds_f2* v0 = (lds_f2*)(row0 + o);
					lds_f2* v1 = (lds_f2*)(row1 + o);
					const vp_f2 t00 = v0[0], t10 = v0[1], t01 = v1[0], t11 = v1[1];
					// one half of a weight pair times both frames' texels.  Written as asm because the compiler, given a broadcast, keeps 16 (w, w) register pairs alive
					// instead of selecting the half in the instruction (op_sel / op_sel_hi: [weight half for lane 0, texel lo], [weight half for lane 1, texel hi])
					vp_f2 m00, m10, m01, m11;
					asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(m00) : "v"(wa_), "v"(t00));
					asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(m10) : "v"(wa_), "v"(t10));
					asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(m01) : "v"(wb_), "v"(t01));
					asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(m11) : "v"(wb_), "v"(t11));
					vp_f2 t = m00 + m10;
					t = t + m01;
					t = t + m11;
					a = (uint32_t)t.x;
					b = (uint32_t)t.y;
					// finished HERE: left alone, the additions sink below the staging that follows the four channels and keep all 32 product registers alive through it
					// (76 instead of 60 VGPRs; at <= 64 there are eight wavefronts per SIMD)
					asm volatile("" : "+v"(a), "+v"(b));
				};
				// halo strips: every wavefront parks its 64 pixels of both frames (ds_write_addtid_b32), one wavefront per frame stores that frame's record a step later
				const bool agg = strips.base != nullptr;
				const uint32_t voff_rec = 4u * (((uint32_t)bxi * shp + (uint32_t)byi * 4u) * 2u * so + (uint32_t)threadIdx.x);
				const bool rec_lane = threadIdx.x < 8 * strips.o;
				const int rec_row = (int)(((float)threadIdx.x + 0.5f) * __builtin_amdgcn_rcpf((float)(2 * (int)so))), rec_e = (int)threadIdx.x - rec_row * 2 * (int)so; // lane / 2o (2o <= 8: exact)
				const uint32_t* rec_src = &fp_rows[0][0][rec_lane ? rec_row : 0][rec_lane ? (rec_e < (int)so ? rec_e : 64 - 2 * (int)so + rec_e) : 0];
				const uint32_t row_m0 = (uint32_t)(uintptr_t)&fp_rows[0][0][wave][0];
				auto park_row = [&](int j, uint32_t px4) { // frame j: buffer (j >> 1) & 1, half j & 1 (asm: hipcc neither counts the LDS store -- the s_waitcnt before the step's barrier covers it -- nor pads the hazard: an SALU write of M0 needs one wait state before an LDS add-TID instruction)
					asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0" : : "v"(px4), "s"(__builtin_amdgcn_readfirstlane((int)(row_m0 + (uint32_t)(((j >> 1) & 1) * 2 + (j & 1)) * 1024u))) : "m0", "memory");
				};
				// Sub-groups of FPS frames.  Vector memory operations of a wavefront retire in order: a load issued behind stores waits for them to be written out,
				// a load issued ahead of them returns as soon as its data is there.  So the boxes of sub-group k + 1 are requested at the START of sub-group k,
				// before any of k's stores -- taking the sensor loads off the critical path is worth 19 % of the kernel (ablation: 1272 -> 1027 us per 256 frames
				// when only the first sub-group loads at all).  hipcc sets the partial vmcnt waits this needs only in straight-line code (at a loop's back edge it
				// merges the wait-count state and waits for everything: 7 % SLOWER, profiles/r05_resample_lds_bound.json), hence: a workgroup whose sub-groups are
				// all complete runs them written out one after the other (FULL), two register sets of FPS dwords taking turns; anything else takes the loop.
				const int nsub = groups_per_wg * (RESAMPLE_FPT / FPS);
				auto request = [&](int gi, uint32_t (&pre)[FPS], int nfg) { // one dword per lane and frame of sub-group gi
					const int g0 = f0 + gi * FPS;
					const __amdgpu_buffer_rsrc_t grs_raw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(raw) + (size_t)g0 * src_frame_bytes, 0, (int)((size_t)nfg * src_frame_bytes), 0x00020000);
#pragma unroll
					for(int j = 0; j < FPS; j++) pre[j] = __builtin_amdgcn_raw_buffer_load_b32(grs_raw, goff, (uint32_t)min(j, nfg - 1) * fb_raw, 0);
				};
				// one sub-group of nfg <= FPS frames, as a loop body (partial workgroups, group settings the written-out stream does not cover)
				auto subgroup = [&](int gi, uint32_t (&pre)[FPS], int nfg) {
					const int g0 = f0 + gi * FPS; // this sub-group's first frame
					// this sub-group's buffer windows (32-bit lane offsets + a per-frame scalar offset; the spans were checked for a full group)
					const __amdgpu_buffer_rsrc_t grs_flat = __builtin_amdgcn_make_buffer_rsrc(flat + (size_t)g0 * npx * 4, 0, (int)((size_t)nfg * npx * 4), 0x00020000);
					const __amdgpu_buffer_rsrc_t grs_strip = __builtin_amdgcn_make_buffer_rsrc(strips.base ? strips.base + (size_t)g0 * strips.frame_words : reinterpret_cast<uint32_t*>(flat), 0,
							(int)((size_t)nfg * strips.frame_words * 4), 0x00020000);
					auto gstore = [&](uint32_t px4, uint32_t soff) { __builtin_amdgcn_raw_buffer_store_b32(px4, grs_flat, voff_out, soff, NTS ? 2 : 0); };
					auto flush = [&](int j) {
						if(agg && wave == (j & 3) && rec_lane)
							__builtin_amdgcn_raw_buffer_store_b32(rec_src[(((j >> 1) & 1) * 2 + (j & 1)) * 256], grs_strip, voff_rec, (uint32_t)j * fb_strip, 0);
					};
					// (no barrier before the first frames are parked: whatever ran before ended with one, and buffer 0 was last read a barrier earlier still)
					put(0, pre[0], pre[1]);
					__syncthreads();
#pragma unroll
					for(int jj = 0; jj < FPS / 2; jj++) {
						const int j = 2 * jj;
						if(j >= nfg) break;
						const int o = (jj & 1) * 1024; // in vp_f2 units: the second buffer starts 2048 floats on
						// (one channel's texels in flight at a time: all sixteen 8-byte reads up front cost 24 more VGPRs)
						uint32_t a0, b0, a1, b1, a2, b2, a3, b3; // a: frame j, b: frame j + 1; the greens halved
						sum2(w0a, w0b, q00, q01, o, a0, b0);
						__builtin_amdgcn_sched_barrier(0);
						sum2(w1a, w1b, q10, q11, o, a1, b1);
						__builtin_amdgcn_sched_barrier(0);
						sum2(w2a, w2b, q20, q21, o, a2, b2);
						__builtin_amdgcn_sched_barrier(0);
						sum2(w3a, w3b, q30, q31, o, a3, b3);
						__builtin_amdgcn_sched_barrier(0); // (the staging below needs registers of its own: only once the products are dead)
						if(j + 2 < nfg) put((jj + 1) & 1, pre[j + 2 < FPS ? j + 2 : j], pre[j + 3 < FPS ? j + 3 : j + 1]);
						const uint32_t pxa = LAYOUT == VP_LAYOUT_RGGB ? drgb_pack7(a0, a1 + a2, a3) : drgb_pack7(a1, a0 + a3, a2);
						const uint32_t pxb = LAYOUT == VP_LAYOUT_RGGB ? drgb_pack7(b0, b1 + b2, b3) : drgb_pack7(b1, b0 + b3, b2);
						gstore(pxa, (uint32_t)j * fb_flat);
						if(j + 1 < nfg) gstore(pxb, (uint32_t)(j + 1) * fb_flat);
						if(agg) {
							park_row(j, pxa);
							park_row(j + 1, pxb);
						}
						if(j > 0) {
							flush(j - 2);
							flush(j - 1);
						}
						asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
						__syncthreads();
					}
					const int last = (nfg - 1) & ~1; // the last pair's records
					flush(last);
					if(last + 1 < nfg) flush(last + 1);
				};
				uint32_t pre_a[FPS];
				const int fr_wg = min(batch - f0, nsub * FPS), nsub_full = fr_wg / FPS; // this workgroup's frames; its complete sub-groups
				if(nsub <= VP_RS_NSUB && nsub_full > 0 && (uint64_t)fr_wg * fb_flat <= 0xffffffffull && (uint64_t)fr_wg * fb_raw <= 0xffffffffull && (uint64_t)fr_wg * fb_strip <= 0xffffffffull) {
					// The workgroup's complete sub-groups (up to VP_RS_NSUB) as ONE written-out stream of frame pairs.  Pair p = frames 2p, 2p + 1 of the workgroup;
					// stage buffer p & 1; sub-group p / (FPS / 2) holds its dwords in register set (p / (FPS / 2)) % NSETS; the first pair of a sub-group requests a
					// later sub-group's boxes before it stores anything (past the workgroup's last complete sub-group: that one again, harmless); every pair parks
					// the next pair's texels (from whichever set holds them) and ends with the one barrier.  The stream is LEFT, never skipped over, at the first
					// sub-group the workgroup does not have (a forward exit: no wait-count state is merged on the path that goes on); a last incomplete sub-group
					// takes the loop body below.
					// What the stream's time is made of (ablations, profiles/r05_resample_issue_stream.json): the flat stores alone run at the chip's write rate, the
					// dword-per-lane loads alone are bound by the CU's address unit, and together they take the SUM of their times whatever the loads' lead -- one
					// in-order memory queue per CU; the four sums cost 4 %, the barrier nothing.  Tried against that and removed: chunked loads by one wavefront in
					// turn, a fifth loading wavefront, two frames per load (v_permlane32_swap), predicated / out-of-range lanes past the box, deeper prefetch.
					constexpr int PPS = FPS / 2, NP = VP_RS_NSUB * PPS; // pairs per sub-group, pairs in all
					static_assert(VP_RS_NSUB % 2 == 0 && NP <= 64, "sub-groups come in pairs");
					constexpr int NSETS = VP_RS_NSETS; // register sets of FPS dwords: sub-group s lives in set s % NSETS and is requested NSETS - 1 sub-groups ahead
					uint32_t pre[NSETS][FPS];
					auto dw = [&](int pp, int k) -> uint32_t { // dword k (0 / 1) of pair pp: sub-group pp / PPS, frame 2 (pp % PPS) + k of it
						return pre[(pp / PPS) % NSETS][2 * (pp % PPS) + k];
					};
					// ONE descriptor per buffer for the workgroup's frames, a frame = a scalar offset (the condition above keeps every offset inside 32 bits);
					// a descriptor per pair cost a 64-bit multiply-add in SALU instructions per pair and buffer
					const __amdgpu_buffer_rsrc_t wrs_flat = __builtin_amdgcn_make_buffer_rsrc(flat + (size_t)f0 * npx * 4, 0, (int)((uint32_t)fr_wg * fb_flat), 0x00020000);
					const __amdgpu_buffer_rsrc_t wrs_raw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(raw) + (size_t)f0 * src_frame_bytes, 0, (int)((uint32_t)fr_wg * fb_raw), 0x00020000);
					const __amdgpu_buffer_rsrc_t wrs_strip = __builtin_amdgcn_make_buffer_rsrc(strips.base ? strips.base + (size_t)f0 * strips.frame_words : reinterpret_cast<uint32_t*>(flat), 0,
							(int)((uint32_t)fr_wg * fb_strip), 0x00020000);
					auto request_s = [&](int sg_rt, uint32_t (&pre)[FPS]) { // the boxes of sub-group sg_rt (< nsub_full)
						const uint32_t so0 = (uint32_t)(sg_rt * FPS) * fb_raw;
#pragma unroll
						for(int j = 0; j < FPS; j++) pre[j] = __builtin_amdgcn_raw_buffer_load_b32(wrs_raw, goff, so0 + (uint32_t)j * fb_raw, 0);
					};
					// The two strip records of pair pp: by wavefronts 0 and 1 (even pairs) / 2 and 3 (odd pairs), one frame each -- one predicated region, the
					// wavefront's part (which half of the parked pair, which frame) in scalars set up once
					const bool flush_even = agg && rec_lane && wave < 2, flush_odd = agg && rec_lane && wave >= 2;
					const uint32_t* rec_src_w = rec_src + (wave & 1) * 256; // the wavefront's frame of a parked pair
					const uint32_t rec_soff_w = (uint32_t)(wave & 1) * fb_strip;
					auto flush_pair = [&](int pp) {
						if(pp & 1 ? flush_odd : flush_even)
							__builtin_amdgcn_raw_buffer_store_b32(rec_src_w[(pp & 1) * 512], wrs_strip, voff_rec, (uint32_t)(2 * pp) * fb_strip + rec_soff_w, 0);
					};
#pragma unroll
					for(int sg = 0; sg < NSETS - 1 && sg < VP_RS_NSUB; sg++) request_s(min(sg, nsub_full - 1), pre[sg]);
					put(0, pre[0][0], pre[0][1]);
					__syncthreads();
					auto pair_step = [&](auto pc) {
						constexpr int pp = decltype(pc)::value;
						if(pp % PPS == 0 && pp / PPS + NSETS - 1 < VP_RS_NSUB) { // first pair of a sub-group: request the boxes of the sub-group NSETS - 1 ahead, before this one's stores
							request_s(min(pp / PPS + NSETS - 1, nsub_full - 1), pre[(pp / PPS + NSETS - 1) % NSETS]);
							__builtin_amdgcn_sched_barrier(0); // (the requests stay where they are written)
						}
						constexpr int o = (pp & 1) * 1024;
						uint32_t a0, b0, a1, b1, a2, b2, a3, b3; // a: frame 2 pp, b: frame 2 pp + 1; the greens halved
						sum2(w0a, w0b, q00, q01, o, a0, b0);
						__builtin_amdgcn_sched_barrier(0);
						sum2(w1a, w1b, q10, q11, o, a1, b1);
						__builtin_amdgcn_sched_barrier(0);
						sum2(w2a, w2b, q20, q21, o, a2, b2);
						__builtin_amdgcn_sched_barrier(0);
						sum2(w3a, w3b, q30, q31, o, a3, b3);
						__builtin_amdgcn_sched_barrier(0);
						if(pp + 1 < NP) put((pp + 1) & 1, dw(pp + 1, 0), dw(pp + 1, 1));
						const uint32_t pxa = LAYOUT == VP_LAYOUT_RGGB ? drgb_pack7(a0, a1 + a2, a3) : drgb_pack7(a1, a0 + a3, a2);
						const uint32_t pxb = LAYOUT == VP_LAYOUT_RGGB ? drgb_pack7(b0, b1 + b2, b3) : drgb_pack7(b1, b0 + b3, b2);
						__builtin_amdgcn_raw_buffer_store_b32(pxa, wrs_flat, voff_out, (uint32_t)(2 * pp) * fb_flat, NTS ? 2 : 0);
						__builtin_amdgcn_raw_buffer_store_b32(pxb, wrs_flat, voff_out, (uint32_t)(2 * pp + 1) * fb_flat, NTS ? 2 : 0);
						if(agg) {
							park_row(2 * pp, pxa);
							park_row(2 * pp + 1, pxb);
						}
						if(pp > 0) flush_pair(pp - 1);
						asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
						__syncthreads();
					};
					auto run_from = [&](auto self, auto pc) -> void { // pair pp and everything after it
						constexpr int pp = decltype(pc)::value;
						if constexpr(pp < NP) {
							if constexpr(pp % PPS == 0 && pp > 0) {
								if(pp / PPS >= nsub_full) { // the workgroup has no such sub-group: leave
									flush_pair(pp - 1);
									return;
								}
							}
							pair_step(pc);
							self(self, std::integral_constant<int, pp + 1>());
						} else flush_pair(NP - 1);
					};
					run_from(run_from, std::integral_constant<int, 0>());
					if(fr_wg % FPS) { // a last incomplete sub-group
						request(nsub_full, pre_a, fr_wg % FPS);
						subgroup(nsub_full, pre_a, fr_wg % FPS);
					}
				} else {
					for(int gi = 0; gi < nsub; gi++) {
						const int g0 = f0 + gi * FPS;
						if(g0 >= batch) break;
						const int nfg = min(batch - g0, FPS);
						request(gi, pre_a, nfg); // (what is left of the previous sub-group's stores is drained by the first wait: once per sub-group)
						subgroup(gi, pre_a, nfg);
					}
				}
			};
			// NEAREST: the boxes of all frames up front (as `run`), then four byte picks per pixel and frame
			auto run_near = [&]() {
				const int own = tid < ndw ? tid : 0, orow = (int)(((float)own + 0.5f) * inv_wdw), ocol = own - orow * wdw;
				const uint32_t goff = (uint32_t)(rmin + orow) * rowb + 4u * (uint32_t)(cmin + ocol);
				uint32_t pre[RESAMPLE_FPT];
#pragma unroll
				for(int j = 0; j < RESAMPLE_FPT; j++) pre[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_raw, goff, (uint32_t)min(j, nf - 1) * fb_raw, 0);
#pragma unroll
				for(int j = 0; j < RESAMPLE_FPT; j++) stage[j][tid] = pre[j];
				__syncthreads();
				const int iea = (ne - rmin) * wdw + (nca >> 2) - cmin, ieb = (ne - rmin) * wdw + (ncb >> 2) - cmin;
				const int ioa = (no - rmin) * wdw + (nca >> 2) - cmin, iob = (no - rmin) * wdw + (ncb >> 2) - cmin;
				const uint32_t sa = (uint32_t)(nca & 3) * 8u, sb = (uint32_t)(ncb & 3) * 8u;
#pragma unroll
				for(int j = 0; j < RESAMPLE_FPT; j++) {
					if(j >= nf) break;
					const uint32_t* st = stage[j];
					const uint32_t tea = (st[iea] >> sa) & 0xFF, teb = (st[ieb] >> sb) & 0xFF, toa = (st[ioa] >> sa) & 0xFF, tob = (st[iob] >> sb) & 0xFF;
					const uint32_t px4 = LAYOUT == VP_LAYOUT_RGGB ? drgb_pack(tea, teb / 2 + toa / 2, tob) : drgb_pack(teb, tea / 2 + tob / 2, toa);
					store_flat(px4, (uint32_t)j * fb_flat);
					if(se) __builtin_amdgcn_raw_buffer_store_b32(px4, rs_strip, voff_se, (uint32_t)j * fb_strip, 0);
				}
			};
			// (two float-staged bodies in one instantiation cost 86 VGPRs instead of 70 -- five wavefronts per SIMD instead of seven: VARIANT 3 falls back to bytes)
			if(NEAR) run_near();
			else if(FP) run_fp();
			else if(FLT) run_flt();
			else if(ndw <= 256) run(std::integral_constant<int, 1>());
			else run(std::integral_constant<int, STAGE_DW / 256>());
			return;
		}
	if(!valid) return;
	uint32_t* out = reinterpret_cast<uint32_t*>(flat) + (size_t)gy * fw + gx;
	if(VARIANT == 2) {
		for(int f = f0; f < f1; f++) {
			Src sr;
			sr.raw = raw + (size_t)f * src_frame_bytes;
			sr.p[0] = sr.p[1] = sr.p[2] = sr.p[3] = nullptr;
			const uint32_t a0 = sample<LAYOUT, VP_SAMPLER_NEAREST, true>(sr, 0, px + 0.25f, py + 0.25f, qw, qh), a1 = sample<LAYOUT, VP_SAMPLER_NEAREST, true>(sr, 1, px - 0.25f, py + 0.25f, qw, qh);
			const uint32_t a2 = sample<LAYOUT, VP_SAMPLER_NEAREST, true>(sr, 2, px + 0.25f, py - 0.25f, qw, qh), a3 = sample<LAYOUT, VP_SAMPLER_NEAREST, true>(sr, 3, px - 0.25f, py - 0.25f, qw, qh);
			const uint32_t px4 = LAYOUT == VP_LAYOUT_RGGB ? drgb_pack(a0, a1 / 2 + a2 / 2, a3) : drgb_pack(a1, a0 / 2 + a3 / 2, a2);
			out[(size_t)f * npx] = px4;
			emit_strip(strips, f, gx, gy, px4);
		}
		return;
	}
	const Bayer4Geom gd = bayer4_geom(px, py, qw, qh);
	const int f1d = min(batch, f0 + groups_per_wg * RESAMPLE_FPT); // every pixel fetches its own windows: ALL the workgroup's groups (only run_fp loops over them itself)
	for(int f = f0; f < f1d; f++) {
		uint32_t p[4];
		bayer4_eval(raw + (size_t)f * src_frame_bytes, gd, p);
		const uint32_t px4 = LAYOUT == VP_LAYOUT_RGGB ? drgb_pack(p[0], p[1] / 2 + p[2] / 2, p[3]) : drgb_pack(p[1], p[0] / 2 + p[3] / 2, p[2]);
		out[(size_t)f * npx] = px4;
		emit_strip(strips, f, gx, gy, px4);
	}
		}; // group
		// (run_fp loops over the workgroup's groups itself, below its set-up; the other bodies take one group per workgroup: the host passes groups_per_wg = 1)
		const int f0 = (int)bz * groups_per_wg * RESAMPLE_FPT;
		group(f0, min(batch, f0 + RESAMPLE_FPT));
	}
}

// BGR8 sensor (OpenCV driver layout, resampling.cl:56-61: the three channels are sampled at the same position): the staged
// kernel for interleaved 3-byte pixels.  Same plan as k_resample_staged: workgroup bounding box of the footprints, the boxes of
// all frames of the group requested up front, BILINEAR_RTZ staged as floats one frame at a time (every byte converted once by
// the thread that fetched it), NEAREST staged as bytes.  Needs 3 * qw % 4 == 0 (dword-aligned sensor rows).
template<int MODE>
__global__ __launch_bounds__(256) void k_resample_staged_bgr(const uint8_t* __restrict__ raw, const float2* __restrict__ map, uint8_t* __restrict__ flat,
		int qw, int qh, int fw, int fh, size_t src_frame_bytes, const VpStrips strips, int batch, int xcd_map) {
	__shared__ uint32_t stage[RESAMPLE_FPT][STAGE_DW];
	__shared__ uint32_t red[4][2];
	constexpr bool NEAR = MODE == VP_SAMPLER_NEAREST;
	const unsigned bz = blockIdx.z;
	unsigned bxi = blockIdx.x, byi = blockIdx.y; // XCD-contiguous tile bands, see k_resample_staged
	if(xcd_map) {
		const unsigned l = vp_xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
		byi = l / gridDim.x;
		bxi = l - byi * gridDim.x;
	}
	const int gx = (int)bxi * 64 + threadIdx.x;
	const int gy = (int)byi * 4 + threadIdx.y;
	const bool valid = gx < fw && gy < fh;
	const float2 pos = map[(size_t)min(gy, fh - 1) * fw + min(gx, fw - 1)];
	const float px = pos.x, py = pos.y;
	const int f0 = (int)bz * RESAMPLE_FPT, f1 = min(batch, f0 + RESAMPLE_FPT), nf = f1 - f0;
	const size_t npx = (size_t)fw * fh;
	const uint32_t rowb = 3u * (uint32_t)qw;
	// footprint: pixel columns i0, i1 and rows j0, j1 (NEAREST: i1 = i0 = floor(u), j likewise)
	Axis ax, ay;
	if(NEAR) {
		ax.i0 = ax.i1 = vp_clampi((int)floorf(px), 0, qw - 1);
		ay.i0 = ay.i1 = vp_clampi((int)floorf(py), 0, qh - 1);
		ax.a = ay.a = 0.0f;
	} else {
		ax = axis_lin(px, qw);
		ay = axis_lin(py, qh);
	}
	const uint32_t lo_w = wave_reduce_pk16<false>((uint32_t)((3 * ax.i0) >> 2) | ((uint32_t)ay.i0 << 16));
	const uint32_t hi_w = wave_reduce_pk16<true>((uint32_t)(((3 * ax.i1 + 2) >> 2) + 1) | ((uint32_t)ay.i1 << 16));
	const int tid = threadIdx.y * 64 + threadIdx.x;
	if(threadIdx.x == 0) {
		red[threadIdx.y][0] = lo_w;
		red[threadIdx.y][1] = hi_w;
	}
	__syncthreads();
	const uint32_t lo = pk16_min(pk16_min(red[0][0], red[1][0]), pk16_min(red[2][0], red[3][0]));
	const uint32_t hi = pk16_max(pk16_max(red[0][1], red[1][1]), pk16_max(red[2][1], red[3][1]));
	const int cmin = (int)(lo & 0xFFFF), rmin = (int)(lo >> 16), cmax = (int)(hi & 0xFFFF), rmax = (int)(hi >> 16);
	const int wdw = cmax - cmin, hr = rmax - rmin + 1, ndw = wdw * hr;
	uint32_t* outp = reinterpret_cast<uint32_t*>(flat) + (size_t)min(gy, fh - 1) * fw + min(gx, fw - 1);
	if(ndw <= STAGE_DW) { // workgroup-uniform
		const __amdgpu_buffer_rsrc_t rs_raw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(raw) + (size_t)f0 * src_frame_bytes, 0, (int)((size_t)nf * src_frame_bytes), 0x00020000);
		const float inv_wdw = __builtin_amdgcn_rcpf((float)wdw);
		const int own = tid < ndw ? tid : 0, orow = (int)(((float)own + 0.5f) * inv_wdw), ocol = own - orow * wdw;
		const uint32_t goff = (uint32_t)(rmin + orow) * rowb + 4u * (uint32_t)(cmin + ocol);
		uint32_t pre[RESAMPLE_FPT];
#pragma unroll
		for(int j = 0; j < RESAMPLE_FPT; j++) pre[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_raw, goff, (uint32_t)min(j, nf - 1) * (uint32_t)src_frame_bytes, 0);
		// byte index (within a box row) of channel 0 of the two footprint columns
		const int b0 = 3 * ax.i0 - 4 * cmin, b1 = 3 * ax.i1 - 4 * cmin;
		if(NEAR) {
#pragma unroll
			for(int j = 0; j < RESAMPLE_FPT; j++) stage[j][tid] = pre[j];
			__syncthreads();
			const int bi = (ay.i0 - rmin) * 4 * wdw + b0;
#pragma unroll
			for(int j = 0; j < RESAMPLE_FPT; j++) {
				if(j >= nf) break;
				const uint8_t* sb = reinterpret_cast<const uint8_t*>(stage[j]) + bi;
				const uint32_t px4 = drgb_pack(sb[2], sb[1], sb[0]); // resampling.cl:56-61: r = channel 2, g = 1, b = 0
				if(valid) {
					outp[(size_t)(f0 + j) * npx] = px4;
					emit_strip(strips, f0 + j, gx, gy, px4);
				}
			}
		} else {
			float* F = reinterpret_cast<float*>(&stage[0][0]); // [2][1024]: one float per staged byte
			const int RP = 4 * wdw;
			float* wq = F + orow * RP + 4 * ocol;
			auto put = [&](int buf, uint32_t v) {
				if(tid < ndw) *reinterpret_cast<float4*>(wq + buf * 1024) = make_float4((float)(v & 0xFF), (float)((v >> 8) & 0xFF), (float)((v >> 16) & 0xFF), (float)(v >> 24));
			};
			const float* t00 = F + (ay.i0 - rmin) * RP + b0;
			const float* t10 = F + (ay.i0 - rmin) * RP + b1;
			const float* t01 = F + (ay.i1 - rmin) * RP + b0;
			const float* t11 = F + (ay.i1 - rmin) * RP + b1;
			put(0, pre[0]);
			__syncthreads();
#pragma unroll
			for(int j = 0; j < RESAMPLE_FPT; j++) {
				if(j >= nf) break;
				const int o = (j & 1) * 1024;
				const uint32_t cb = bilin(t00[o], t10[o], t01[o], t11[o], ax.a, ay.a);
				const uint32_t cg = bilin(t00[o + 1], t10[o + 1], t01[o + 1], t11[o + 1], ax.a, ay.a);
				const uint32_t cr = bilin(t00[o + 2], t10[o + 2], t01[o + 2], t11[o + 2], ax.a, ay.a);
				if(j + 1 < nf) put((j + 1) & 1, pre[j + 1 < RESAMPLE_FPT ? j + 1 : j]);
				const uint32_t px4 = drgb_pack(cr, cg, cb);
				if(valid) {
					outp[(size_t)(f0 + j) * npx] = px4;
					emit_strip(strips, f0 + j, gx, gy, px4);
				}
				__syncthreads();
			}
		}
		return;
	}
	if(!valid) return;
	for(int f = f0; f < f1; f++) { // box too large for the stage: sample directly
		Src sr;
		sr.raw = raw + (size_t)f * src_frame_bytes;
		sr.p[0] = sr.p[1] = sr.p[2] = sr.p[3] = nullptr;
		const uint32_t r = sample<VP_LAYOUT_BGR, MODE, true>(sr, 2, px, py, qw, qh), g = sample<VP_LAYOUT_BGR, MODE, true>(sr, 1, px, py, qw, qh), b = sample<VP_LAYOUT_BGR, MODE, true>(sr, 0, px, py, qw, qh);
		const uint32_t px4 = drgb_pack(r, g, b);
		outp[(size_t)f * npx] = px4;
		emit_strip(strips, f, gx, gy, px4);
	}
}

// Unstaged coordinate-map kernel (odd quad widths, single frames, sensors the staged kernels do not take).  A thread owns MAPPED_ROWS
// vertically adjacent flat pixels: one flat pixel per thread made this a 49 152-wavefront launch for one 2048x1536 frame, bound by the
// rate at which wavefronts can be started (~11 us of a 22 us kernel).
constexpr int MAPPED_ROWS = 4;
template<int LAYOUT, int MODE>
__global__ __launch_bounds__(256) void k_resample_mapped(const uint8_t* __restrict__ raw, const float2* __restrict__ map, uint8_t* __restrict__ flat,
		int qw, int qh, int fw, int fh, size_t src_frame_bytes, const VpStrips strips, int batch) {
	// 1-D grid, XCD-aware (vp_xcd_remap): consecutive logical ids walk along x, then y, then the frame group
	const unsigned nbx = (unsigned)(fw + 63) / 64, nby = (unsigned)(fh + 4 * MAPPED_ROWS - 1) / (4 * MAPPED_ROWS);
	const unsigned lb = vp_xcd_remap(blockIdx.x, gridDim.x);
	const unsigned bz = lb / (nbx * nby), brem = lb - bz * nbx * nby;
	const int gx = (int)(brem % nbx) * 64 + threadIdx.x;
	const int f0 = (int)bz * RESAMPLE_FPT, f1 = min(batch, f0 + RESAMPLE_FPT);
	const size_t npx = (size_t)fw * fh;
	if(gx >= fw) return;
	float2 posr[MAPPED_ROWS]; // all rows' coordinates requested before the first one is used
#pragma unroll
	for(int rr = 0; rr < MAPPED_ROWS; rr++) posr[rr] = map[(size_t)min(((int)(brem / nbx) * MAPPED_ROWS + rr) * 4 + (int)threadIdx.y, fh - 1) * fw + gx];
#pragma unroll
	for(int rr = 0; rr < MAPPED_ROWS; rr++) {
		const int gy = ((int)(brem / nbx) * MAPPED_ROWS + rr) * 4 + threadIdx.y;
		if(gy >= fh) break;
		const float2 pos = posr[rr];
		const float px = pos.x, py = pos.y;
		uint32_t* out = reinterpret_cast<uint32_t*>(flat) + (size_t)gy * fw + gx;
		if(LAYOUT != VP_LAYOUT_BGR && MODE == VP_SAMPLER_BILINEAR_RTZ && qw >= 4) {
			const Bayer4Geom g = bayer4_geom(px, py, qw, qh);
#pragma unroll 2
			for(int f = f0; f < f1; f++) {
				uint32_t p[4];
				bayer4_eval(raw + (size_t)f * src_frame_bytes, g, p);
				const uint32_t px4 = LAYOUT == VP_LAYOUT_RGGB ? drgb_pack(p[0], p[1] / 2 + p[2] / 2, p[3]) : drgb_pack(p[1], p[0] / 2 + p[3] / 2, p[2]);
				out[(size_t)f * npx] = px4;
				emit_strip(strips, f, gx, gy, px4);
			}
			continue;
		}
		for(int f = f0; f < f1; f++) {
			Src s;
			s.raw = raw + (size_t)f * src_frame_bytes;
			s.p[0] = s.p[1] = s.p[2] = s.p[3] = nullptr;
			uint32_t r, g, b;
			if(LAYOUT == VP_LAYOUT_BGR) {
				r = sample<LAYOUT, MODE, true>(s, 2, px, py, qw, qh);
				g = sample<LAYOUT, MODE, true>(s, 1, px, py, qw, qh);
				b = sample<LAYOUT, MODE, true>(s, 0, px, py, qw, qh);
			} else if(LAYOUT == VP_LAYOUT_RGGB) {
				r = sample<LAYOUT, MODE, true>(s, 0, px + 0.25f, py + 0.25f, qw, qh);
				g = sample<LAYOUT, MODE, true>(s, 1, px - 0.25f, py + 0.25f, qw, qh) / 2 + sample<LAYOUT, MODE, true>(s, 2, px + 0.25f, py - 0.25f, qw, qh) / 2;
				b = sample<LAYOUT, MODE, true>(s, 3, px - 0.25f, py - 0.25f, qw, qh);
			} else {
				r = sample<LAYOUT, MODE, true>(s, 1, px - 0.25f, py + 0.25f, qw, qh);
				g = sample<LAYOUT, MODE, true>(s, 0, px + 0.25f, py + 0.25f, qw, qh) / 2 + sample<LAYOUT, MODE, true>(s, 3, px - 0.25f, py - 0.25f, qw, qh) / 2;
				b = sample<LAYOUT, MODE, true>(s, 2, px + 0.25f, py - 0.25f, qw, qh);
			}
			const uint32_t px4 = drgb_pack(r, g, b);
			out[(size_t)f * npx] = px4;
			emit_strip(strips, f, gx, gy, px4);
		}
	}
}

extern "C" int vp_resample_map(VpContext* ctx, int stream, const VpCameraModel* model, float max_robot_height, float field_scale, float off_x, float off_y,
		int fw, int fh, float* map) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(model && map && fw > 0 && fh > 0 && (reinterpret_cast<uintptr_t>(map) & 7) == 0, "vp_resample_map: bad arguments");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "resampling map");
	ResampleArgs a;
	a.m = *model;
	a.height = max_robot_height;
	a.fs = field_scale;
	a.ox = off_x;
	a.oy = off_y;
	a.qw = a.qh = 0;
	a.fw = fw;
	a.fh = fh;
	a.strips = vp_make_strips(nullptr, 64, 1, 1);
	k_resample_map<<<dim3(vp_div_up(fw, 64), vp_div_up(fh, 4)), dim3(64, 4), 0, ctx->streams[stream]>>>(reinterpret_cast<float2*>(map), a);
	return vp_launch_ok("resampling map");
}

int vp_resample_raw_mapped_impl(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, const float* map, int sampler_mode,
		int fw, int fh, int batch, uint8_t* flat, const VpStrips& strips);

extern "C" int vp_resample_raw_mapped(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, const float* map, int sampler_mode,
		int fw, int fh, int batch, uint8_t* flat) {
	VpStrips none = vp_make_strips(nullptr, 64, 1, 1);
	return vp_resample_raw_mapped_impl(ctx, stream, raw, layout, qw, qh, map, sampler_mode, fw, fh, batch, flat, none);
}

int vp_resample_raw_mapped_impl(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, const float* map, int sampler_mode,
		int fw, int fh, int batch, uint8_t* flat, const VpStrips& strips) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(raw && map && flat && qw > 0 && qh > 0 && fw > 0 && fh > 0 && batch > 0 && batch <= 65535, "vp_resample_raw_mapped: bad arguments");
	VP_REQUIRE(layout >= 0 && layout <= 2 && (sampler_mode == VP_SAMPLER_NEAREST || sampler_mode == VP_SAMPLER_BILINEAR_RTZ), "vp_resample_raw_mapped: bad layout/sampler mode");
	VP_REQUIRE((reinterpret_cast<uintptr_t>(flat) & 3) == 0 && (reinterpret_cast<uintptr_t>(map) & 7) == 0, "vp_resample_raw_mapped: misaligned buffer");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "raw2quad+resampling");
	const size_t q = (size_t)qw * qh, frame_bytes = layout == VP_LAYOUT_BGR ? 3 * q : 4 * q;
	const dim3 grid(vp_div_up(fw, 64) * vp_div_up(fh, 4 * MAPPED_ROWS) * vp_div_up(batch, RESAMPLE_FPT)), block(64, 4); // k_resample_mapped
	hipStream_t s = ctx->streams[stream];
	const float2* m2 = reinterpret_cast<const float2*>(map);
	// staged kernel: Bayer + BILINEAR_RTZ with dword-aligned sensor rows; a frame group's spans must fit the 32-bit buffer offsets
	const size_t span = (size_t)RESAMPLE_FPT * std::max(std::max(frame_bytes, (size_t)fw * fh * 4), strips.base ? strips.frame_words * 4 : (size_t)0);
	if(layout == VP_LAYOUT_BGR && (qw & 3) == 0 && batch >= 2 && (size_t)RESAMPLE_FPT * frame_bytes < 0x7fffffffu && (reinterpret_cast<uintptr_t>(raw) & 3) == 0 &&
			qw < 21845 && qh < 65536 && vp_div_up(fh, 4) <= 65535 && ctx->tune.resample_bgr_staged) {
		const dim3 grid3(vp_div_up(fw, 64), vp_div_up(fh, 4), vp_div_up(batch, RESAMPLE_FPT));
		const int xcd_map = ctx->tune.resample_xcd_bands >= 0 ? ctx->tune.resample_xcd_bands : (grid3.x * grid3.y >= 4096u); // see the Bayer launch below
		if(sampler_mode == VP_SAMPLER_NEAREST) k_resample_staged_bgr<VP_SAMPLER_NEAREST><<<grid3, block, 0, s>>>(raw, m2, flat, qw, qh, fw, fh, frame_bytes, strips, batch, xcd_map);
		else k_resample_staged_bgr<VP_SAMPLER_BILINEAR_RTZ><<<grid3, block, 0, s>>>(raw, m2, flat, qw, qh, fw, fh, frame_bytes, strips, batch, xcd_map);
		return vp_launch_ok("raw2quad+resampling (staged, BGR)");
	}
	if(layout != VP_LAYOUT_BGR && qw >= 4 && (qw & 1) == 0 && batch >= 2 && span < 0x7fffffffu
			&& ((reinterpret_cast<uintptr_t>(raw) | frame_bytes) & 3) == 0 && qw < 65536 && qh < 32768 && vp_div_up(fh, 4) <= 65535) {
		const dim3 grid(vp_div_up(fw, 64), vp_div_up(fh, 4), vp_div_up(batch, RESAMPLE_FPT));
		const int fs = ctx->tune.resample_float_stage; // 2 (default): floats, two frames side by side; 1: floats; 0: bytes (vp_ctx_tune "resample_float_stage")
		const int variant = sampler_mode == VP_SAMPLER_NEAREST ? 2 : (fs >= 2 ? 3 : fs == 1 ? 1 : 0);
		// XCD-contiguous tile bands for large frames (measured at 2048x1536: HBM reads 1.38 -> 0.60 GB per 128 frames, time unchanged;
		// at 640x480 the bands unbalance the XCDs: 87 -> 99 us, so small grids keep the plain mapping).  vp_ctx_tune "resample_xcd_bands" forces it.
		const int xcd_map = ctx->tune.resample_xcd_bands >= 0 ? ctx->tune.resample_xcd_bands : (grid.x * grid.y >= 4096u);
		const bool flat_nt = (size_t)fw * fh * 4 * batch > (size_t(112) << 20); // see store_flat in k_resample_staged
		// frame groups per workgroup: the set-up is paid once for all of them; only where enough workgroups remain to fill the chip several times over
		const int n_groups = (int)vp_div_up(batch, RESAMPLE_FPT);
		// (measured at 256 frames with 16 frames in flight: 1 -> 1367, 2 -> 1305, 4 -> 1304, 8 -> 1308 us per launch; with 8 in flight: 2 -> 1292, 4 -> 1271;
		// only the frame-pair body loops over groups)
		const int gpw = variant != 3 ? 1 : std::min(ctx->tune.resample_groups_per_wg > 0 ? ctx->tune.resample_groups_per_wg : 4, n_groups);
		const dim3 gridg(grid.x, grid.y, vp_div_up(n_groups, gpw));
#define VP_RST(L, V) \
		do { \
			if(flat_nt) k_resample_staged<L, V, true><<<gridg, block, 0, s>>>(raw, m2, flat, qw, qh, fw, fh, frame_bytes, strips, batch, xcd_map, gpw); \
			else k_resample_staged<L, V, false><<<gridg, block, 0, s>>>(raw, m2, flat, qw, qh, fw, fh, frame_bytes, strips, batch, xcd_map, gpw); \
		} while(0)
		if(layout == VP_LAYOUT_RGGB) {
			if(variant == 3) VP_RST(VP_LAYOUT_RGGB, 3);
			else if(variant == 2) VP_RST(VP_LAYOUT_RGGB, 2);
			else if(variant == 1) VP_RST(VP_LAYOUT_RGGB, 1);
			else VP_RST(VP_LAYOUT_RGGB, 0);
		} else {
			if(variant == 3) VP_RST(VP_LAYOUT_GRBG, 3);
			else if(variant == 2) VP_RST(VP_LAYOUT_GRBG, 2);
			else if(variant == 1) VP_RST(VP_LAYOUT_GRBG, 1);
			else VP_RST(VP_LAYOUT_GRBG, 0);
		}
#undef VP_RST
		return vp_launch_ok("raw2quad+resampling (staged)");
	}
#define VP_RM(L, M) k_resample_mapped<L, M><<<grid, block, 0, s>>>(raw, m2, flat, qw, qh, fw, fh, frame_bytes, strips, batch)
	if(sampler_mode == VP_SAMPLER_NEAREST) {
		if(layout == VP_LAYOUT_RGGB) VP_RM(VP_LAYOUT_RGGB, VP_SAMPLER_NEAREST);
		else if(layout == VP_LAYOUT_GRBG) VP_RM(VP_LAYOUT_GRBG, VP_SAMPLER_NEAREST);
		else VP_RM(VP_LAYOUT_BGR, VP_SAMPLER_NEAREST);
	} else {
		if(layout == VP_LAYOUT_RGGB) VP_RM(VP_LAYOUT_RGGB, VP_SAMPLER_BILINEAR_RTZ);
		else if(layout == VP_LAYOUT_GRBG) VP_RM(VP_LAYOUT_GRBG, VP_SAMPLER_BILINEAR_RTZ);
		else VP_RM(VP_LAYOUT_BGR, VP_SAMPLER_BILINEAR_RTZ);
	}
#undef VP_RM
	return vp_launch_ok("raw2quad+resampling (mapped)");
}

template<bool FROM_RAW>
static int resample_dispatch(VpContext* ctx, int stream, const uint8_t* src, const uint8_t* const* planes4, int layout, int qw, int qh, const VpCameraModel* model,
		float height, float fs, float ox, float oy, int mode, int fw, int fh, int batch, uint8_t* flat, const char* name, const VpStrips* strips = nullptr) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE((src || planes4) && flat && model, "%s: null pointer", name);
	VP_REQUIRE(qw > 0 && qh > 0 && fw > 0 && fh > 0 && batch > 0 && batch <= 65535, "%s: bad shape", name);
	VP_REQUIRE(layout >= 0 && layout <= 2 && (mode == VP_SAMPLER_NEAREST || mode == VP_SAMPLER_BILINEAR_RTZ), "%s: bad layout/sampler mode", name);
	VP_REQUIRE((reinterpret_cast<uintptr_t>(flat) & 3) == 0, "%s: flat must be 4-byte aligned", name);
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, name);
	ResampleArgs a;
	a.m = *model;
	a.height = height;
	a.fs = fs;
	a.ox = ox;
	a.oy = oy;
	a.qw = qw;
	a.qh = qh;
	a.fw = fw;
	a.fh = fh;
	const size_t q = (size_t)qw * qh;
	const size_t frame_bytes = FROM_RAW ? (layout == VP_LAYOUT_BGR ? 3 * q : 4 * q) : 4 * q;
	a.strips = strips ? *strips : vp_make_strips(nullptr, 64, 1, 1);
	for(int c = 0; c < 4; c++) a.pl[c] = FROM_RAW ? nullptr : (planes4 ? planes4[c] : src + c * q);
	a.pl_bstride = planes4 ? 0 : 4 * q;
	const dim3 grid(vp_div_up(fw, 64), vp_div_up(fh, 4), batch), block(64, 4);
	hipStream_t s = ctx->streams[stream];
#define VP_RS(L, M) k_resample<L, M, FROM_RAW><<<grid, block, 0, s>>>(src, flat, a, frame_bytes)
	if(mode == VP_SAMPLER_NEAREST) {
		if(layout == VP_LAYOUT_RGGB) VP_RS(VP_LAYOUT_RGGB, VP_SAMPLER_NEAREST);
		else if(layout == VP_LAYOUT_GRBG) VP_RS(VP_LAYOUT_GRBG, VP_SAMPLER_NEAREST);
		else VP_RS(VP_LAYOUT_BGR, VP_SAMPLER_NEAREST);
	} else {
		if(layout == VP_LAYOUT_RGGB) VP_RS(VP_LAYOUT_RGGB, VP_SAMPLER_BILINEAR_RTZ);
		else if(layout == VP_LAYOUT_GRBG) VP_RS(VP_LAYOUT_GRBG, VP_SAMPLER_BILINEAR_RTZ);
		else VP_RS(VP_LAYOUT_BGR, VP_SAMPLER_BILINEAR_RTZ);
	}
#undef VP_RS
	return vp_launch_ok(name);
}

extern "C" int vp_resample(VpContext* ctx, int stream, const uint8_t* planes, int layout, int qw, int qh, const VpCameraModel* model,
		float max_robot_height, float field_scale, float off_x, float off_y, int sampler_mode, int fw, int fh, int batch, uint8_t* flat) {
	return resample_dispatch<false>(ctx, stream, planes, nullptr, layout, qw, qh, model, max_robot_height, field_scale, off_x, off_y, sampler_mode, fw, fh, batch, flat, "resampling");
}

extern "C" int vp_resample4(VpContext* ctx, int stream, const uint8_t* plane0, const uint8_t* plane1, const uint8_t* plane2, const uint8_t* plane3, int layout, int qw, int qh,
		const VpCameraModel* model, float max_robot_height, float field_scale, float off_x, float off_y, int sampler_mode, int fw, int fh, uint8_t* flat) {
	VP_REQUIRE(plane0 && plane1 && plane2 && (plane3 || layout == VP_LAYOUT_BGR), "vp_resample4: null plane");
	const uint8_t* pl[4] = {plane0, plane1, plane2, plane3 ? plane3 : plane2};
	return resample_dispatch<false>(ctx, stream, nullptr, pl, layout, qw, qh, model, max_robot_height, field_scale, off_x, off_y, sampler_mode, fw, fh, 1, flat, "resampling");
}

int vp_resample_impl(VpContext* ctx, int stream, const uint8_t* src, bool from_raw, int layout, int qw, int qh, const VpCameraModel* model,
		float max_robot_height, float field_scale, float off_x, float off_y, int sampler_mode, int fw, int fh, int batch, uint8_t* flat, const VpStrips& strips) {
	if(from_raw) return resample_dispatch<true>(ctx, stream, src, nullptr, layout, qw, qh, model, max_robot_height, field_scale, off_x, off_y, sampler_mode, fw, fh, batch, flat, "raw2quad+resampling", &strips);
	return resample_dispatch<false>(ctx, stream, src, nullptr, layout, qw, qh, model, max_robot_height, field_scale, off_x, off_y, sampler_mode, fw, fh, batch, flat, "resampling", &strips);
}

extern "C" int vp_resample_raw(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, const VpCameraModel* model,
		float max_robot_height, float field_scale, float off_x, float off_y, int sampler_mode, int fw, int fh, int batch, uint8_t* flat) {
	return resample_dispatch<true>(ctx, stream, raw, nullptr, layout, qw, qh, model, max_robot_height, field_scale, off_x, off_y, sampler_mode, fw, fh, batch, flat, "raw2quad+resampling");
}

// ---------------------------------------------------------------- debug / stream colour kernels ----------
// quad2rgba.cl:24-53, quad2nv12.cl:24-58, rgba2nv12.cl:22-31, f2nv12.cl:22-26 (Resources.cpp:145-149, 166-186).
// NV12 frame: w*h luma bytes, then ceil(h/2) rows of w interleaved U,V bytes.  In the reference all four pixels of a 2x2 block
// store the block's UV pair (a race); here ONE thread owns a 2x2 block and stores the value of its last pixel in raster order,
// which is what sequential execution of the reference produces (the oracle's definition).

// colour of quad pixel (x, y): quad2rgba.cl:27-50.  BGR reads the texel itself (int2 coordinates), Bayer samples at +-0.25.
template<int LAYOUT, int MODE, bool FROM_RAW>
__device__ __forceinline__ void quad_colour(const Src& s, int x, int y, int qw, int qh, uint32_t& r, uint32_t& g, uint32_t& b) {
	const float fx = (float)x, fy = (float)y;
	if(LAYOUT == VP_LAYOUT_BGR) {
		r = texel<LAYOUT, FROM_RAW>(s, 2, x, y, qw, qh);
		g = texel<LAYOUT, FROM_RAW>(s, 1, x, y, qw, qh);
		b = texel<LAYOUT, FROM_RAW>(s, 0, x, y, qw, qh);
	} else if(LAYOUT == VP_LAYOUT_RGGB) {
		r = sample<LAYOUT, MODE, FROM_RAW>(s, 0, fx + 0.25f, fy + 0.25f, qw, qh);
		g = sample<LAYOUT, MODE, FROM_RAW>(s, 1, fx - 0.25f, fy + 0.25f, qw, qh) / 2 + sample<LAYOUT, MODE, FROM_RAW>(s, 2, fx + 0.25f, fy - 0.25f, qw, qh) / 2;
		b = sample<LAYOUT, MODE, FROM_RAW>(s, 3, fx - 0.25f, fy - 0.25f, qw, qh);
	} else {
		r = sample<LAYOUT, MODE, FROM_RAW>(s, 1, fx - 0.25f, fy + 0.25f, qw, qh);
		g = sample<LAYOUT, MODE, FROM_RAW>(s, 0, fx + 0.25f, fy + 0.25f, qw, qh) / 2 + sample<LAYOUT, MODE, FROM_RAW>(s, 3, fx - 0.25f, fy - 0.25f, qw, qh) / 2;
		b = sample<LAYOUT, MODE, FROM_RAW>(s, 2, fx + 0.25f, fy - 0.25f, qw, qh);
	}
}

// BT.601 limited range in integers (quad2nv12.cl:52,56-57): Y in uint arithmetic, U/V in int arithmetic with truncating division
__device__ __forceinline__ uint32_t nv12_y(uint32_t r, uint32_t g, uint32_t b) { return min((66u * r + 129u * g + 25u * b) / 256u + 16u, 255u); }
__device__ __forceinline__ uint32_t nv12_u(uint32_t r, uint32_t g, uint32_t b) { return (uint32_t)vp_clampi((-38 * (int)r - 74 * (int)g + 112 * (int)b) / 256 + 128, 0, 255); }
__device__ __forceinline__ uint32_t nv12_v(uint32_t r, uint32_t g, uint32_t b) { return (uint32_t)vp_clampi((112 * (int)r - 94 * (int)g - 18 * (int)b) / 256 + 128, 0, 255); }

extern "C" size_t vp_nv12_bytes(int w, int h) { return w > 0 && h > 0 ? (size_t)w * h + (size_t)w * ((h + 1) / 2) : 0; }

struct QuadSrcArgs {
	const uint8_t* raw;  // FROM_RAW: batch x frame_bytes
	const uint8_t* pl[4]; // planes
	size_t bstride;      // bytes between frames (raw: frame bytes; planes: 4*qw*qh or 0 for separately allocated planes)
};

template<int LAYOUT, int MODE, bool FROM_RAW>
__global__ __launch_bounds__(256) void k_quad2rgba(const QuadSrcArgs a, uint32_t* __restrict__ rgba, int qw, int qh) {
	const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
	if(x >= qw || y >= qh) return;
	Src s;
	s.raw = a.raw + (size_t)blockIdx.z * a.bstride;
#pragma unroll
	for(int c = 0; c < 4; c++) s.p[c] = a.pl[c] + (size_t)blockIdx.z * a.bstride;
	uint32_t r, g, b;
	quad_colour<LAYOUT, MODE, FROM_RAW>(s, x, y, qw, qh, r, g, b);
	rgba[(size_t)blockIdx.z * qw * qh + (size_t)y * qw + x] = r | (g << 8) | (b << 16) | 0xFF000000u; // channels are bytes: write_imageui cannot saturate
}

// one thread per 2x2 block (w even): two 2-byte luma stores, one 2-byte chroma store
template<int LAYOUT, int MODE, bool FROM_RAW>
__global__ __launch_bounds__(256) void k_quad2nv12(const QuadSrcArgs a, uint8_t* __restrict__ nv12, int qw, int qh, size_t out_stride) {
	const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
	if(2 * bx >= qw || 2 * by >= qh) return;
	Src s;
	s.raw = a.raw + (size_t)blockIdx.z * a.bstride;
#pragma unroll
	for(int c = 0; c < 4; c++) s.p[c] = a.pl[c] + (size_t)blockIdx.z * a.bstride;
	uint8_t* out = nv12 + (size_t)blockIdx.z * out_stride;
	uint32_t r = 0, g = 0, b = 0;
#pragma unroll
	for(int dy = 0; dy < 2; dy++) {
		const int y = 2 * by + dy;
		if(y >= qh) break;
		uint32_t r0, g0, b0;
		quad_colour<LAYOUT, MODE, FROM_RAW>(s, 2 * bx, y, qw, qh, r0, g0, b0);
		quad_colour<LAYOUT, MODE, FROM_RAW>(s, 2 * bx + 1, y, qw, qh, r, g, b); // the last one evaluated is the block's last pixel in raster order
		*reinterpret_cast<uint16_t*>(out + (size_t)y * qw + 2 * bx) = (uint16_t)(nv12_y(r0, g0, b0) | (nv12_y(r, g, b) << 8));
	}
	*reinterpret_cast<uint16_t*>(out + (size_t)qw * qh + (size_t)by * qw + 2 * bx) = (uint16_t)(nv12_u(r, g, b) | (nv12_v(r, g, b) << 8));
}

__global__ __launch_bounds__(256) void k_rgba2nv12(const uint32_t* __restrict__ rgba, uint8_t* __restrict__ nv12, int w, int h, size_t out_stride) {
	const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
	if(2 * bx >= w || 2 * by >= h) return;
	const uint32_t* in = rgba + (size_t)blockIdx.z * w * h;
	uint8_t* out = nv12 + (size_t)blockIdx.z * out_stride;
	uint32_t last = 0;
#pragma unroll
	for(int dy = 0; dy < 2; dy++) {
		const int y = 2 * by + dy;
		if(y >= h) break;
		const uint2 v = *reinterpret_cast<const uint2*>(in + (size_t)y * w + 2 * bx);
		last = v.y;
		*reinterpret_cast<uint16_t*>(out + (size_t)y * w + 2 * bx) =
				(uint16_t)(nv12_y(v.x & 0xFF, (v.x >> 8) & 0xFF, (v.x >> 16) & 0xFF) | (nv12_y(v.y & 0xFF, (v.y >> 8) & 0xFF, (v.y >> 16) & 0xFF) << 8));
	}
	const uint32_t r = last & 0xFF, g = (last >> 8) & 0xFF, b = (last >> 16) & 0xFF;
	*reinterpret_cast<uint16_t*>(out + (size_t)w * h + (size_t)by * w + 2 * bx) = (uint16_t)(nv12_u(r, g, b) | (nv12_v(r, g, b) << 8));
}

// convert_uchar_sat(float): round toward zero, saturate, NaN -> 0 (OpenCL C 6.4.3)
__device__ __forceinline__ uint32_t f2u8_sat(float v) { return !(v > 0.0f) ? 0u : (v >= 255.0f ? 255u : (uint32_t)v); }

__global__ __launch_bounds__(256) void k_f2nv12(const float* __restrict__ img, uint8_t* __restrict__ nv12, int w, int h, size_t out_stride) {
	const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
	if(2 * bx >= w || 2 * by >= h) return;
	const float* in = img + (size_t)blockIdx.z * w * h;
	uint8_t* out = nv12 + (size_t)blockIdx.z * out_stride;
#pragma unroll
	for(int dy = 0; dy < 2; dy++) {
		const int y = 2 * by + dy;
		if(y >= h) break;
		const float2 v = *reinterpret_cast<const float2*>(in + (size_t)y * w + 2 * bx);
		*reinterpret_cast<uint16_t*>(out + (size_t)y * w + 2 * bx) = (uint16_t)(f2u8_sat(v.x + 127) | (f2u8_sat(v.y + 127) << 8)); // f2nv12.cl:24
	}
	*reinterpret_cast<uint16_t*>(out + (size_t)w * h + (size_t)by * w + 2 * bx) = 0x7F7F; // :25
}

template<bool FROM_RAW, bool NV12>
static int quad_colour_dispatch(VpContext* ctx, int stream, const uint8_t* src, const uint8_t* const* planes4, int layout, int qw, int qh, int mode, int batch, uint8_t* out,
		const char* name) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE((src || planes4) && out && qw > 0 && qh > 0 && batch > 0 && batch <= 65535, "%s: bad arguments", name);
	VP_REQUIRE(layout >= 0 && layout <= 2 && (mode == VP_SAMPLER_NEAREST || mode == VP_SAMPLER_BILINEAR_RTZ), "%s: bad layout/sampler mode", name);
	VP_REQUIRE(!NV12 || (qw & 1) == 0, "%s: NV12 needs an even width", name);
	VP_REQUIRE(NV12 ? (reinterpret_cast<uintptr_t>(out) & 1) == 0 : (reinterpret_cast<uintptr_t>(out) & 3) == 0, "%s: misaligned output", name);
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, name);
	const size_t q = (size_t)qw * qh;
	QuadSrcArgs a;
	a.raw = FROM_RAW ? src : nullptr;
	for(int c = 0; c < 4; c++) a.pl[c] = FROM_RAW ? nullptr : (planes4 ? planes4[c] : src + c * q);
	a.bstride = FROM_RAW ? (layout == VP_LAYOUT_BGR ? 3 * q : 4 * q) : (planes4 ? 0 : 4 * q);
	hipStream_t s = ctx->streams[stream];
	const dim3 block(64, 4);
	const dim3 grid(NV12 ? vp_div_up((qw + 1) / 2, 64) : vp_div_up(qw, 64), NV12 ? vp_div_up((qh + 1) / 2, 4) : vp_div_up(qh, 4), batch);
	const size_t ostride = vp_nv12_bytes(qw, qh);
#define VP_QC(L, M) \
	do { \
		if(NV12) k_quad2nv12<L, M, FROM_RAW><<<grid, block, 0, s>>>(a, out, qw, qh, ostride); \
		else k_quad2rgba<L, M, FROM_RAW><<<grid, block, 0, s>>>(a, reinterpret_cast<uint32_t*>(out), qw, qh); \
	} while(0)
	if(mode == VP_SAMPLER_NEAREST) {
		if(layout == VP_LAYOUT_RGGB) VP_QC(VP_LAYOUT_RGGB, VP_SAMPLER_NEAREST);
		else if(layout == VP_LAYOUT_GRBG) VP_QC(VP_LAYOUT_GRBG, VP_SAMPLER_NEAREST);
		else VP_QC(VP_LAYOUT_BGR, VP_SAMPLER_NEAREST);
	} else {
		if(layout == VP_LAYOUT_RGGB) VP_QC(VP_LAYOUT_RGGB, VP_SAMPLER_BILINEAR_RTZ);
		else if(layout == VP_LAYOUT_GRBG) VP_QC(VP_LAYOUT_GRBG, VP_SAMPLER_BILINEAR_RTZ);
		else VP_QC(VP_LAYOUT_BGR, VP_SAMPLER_BILINEAR_RTZ);
	}
#undef VP_QC
	return vp_launch_ok(name);
}

extern "C" int vp_quad2rgba(VpContext* ctx, int stream, const uint8_t* planes, int layout, int qw, int qh, int sampler_mode, int batch, uint8_t* rgba) {
	return quad_colour_dispatch<false, false>(ctx, stream, planes, nullptr, layout, qw, qh, sampler_mode, batch, rgba, "quad2rgba");
}
extern "C" int vp_quad2rgba4(VpContext* ctx, int stream, const uint8_t* plane0, const uint8_t* plane1, const uint8_t* plane2, const uint8_t* plane3, int layout, int qw, int qh,
		int sampler_mode, uint8_t* rgba) {
	VP_REQUIRE(plane0 && plane1 && plane2 && (plane3 || layout == VP_LAYOUT_BGR), "vp_quad2rgba4: null plane");
	const uint8_t* pl[4] = {plane0, plane1, plane2, plane3 ? plane3 : plane2};
	return quad_colour_dispatch<false, false>(ctx, stream, nullptr, pl, layout, qw, qh, sampler_mode, 1, rgba, "quad2rgba");
}
extern "C" int vp_raw2rgba(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, int sampler_mode, int batch, uint8_t* rgba) {
	return quad_colour_dispatch<true, false>(ctx, stream, raw, nullptr, layout, qw, qh, sampler_mode, batch, rgba, "raw2quad+quad2rgba");
}
extern "C" int vp_quad2nv12(VpContext* ctx, int stream, const uint8_t* planes, int layout, int qw, int qh, int sampler_mode, int batch, uint8_t* nv12) {
	return quad_colour_dispatch<false, true>(ctx, stream, planes, nullptr, layout, qw, qh, sampler_mode, batch, nv12, "quad2nv12");
}
extern "C" int vp_quad2nv124(VpContext* ctx, int stream, const uint8_t* plane0, const uint8_t* plane1, const uint8_t* plane2, const uint8_t* plane3, int layout, int qw, int qh,
		int sampler_mode, uint8_t* nv12) {
	VP_REQUIRE(plane0 && plane1 && plane2 && (plane3 || layout == VP_LAYOUT_BGR), "vp_quad2nv124: null plane");
	const uint8_t* pl[4] = {plane0, plane1, plane2, plane3 ? plane3 : plane2};
	return quad_colour_dispatch<false, true>(ctx, stream, nullptr, pl, layout, qw, qh, sampler_mode, 1, nv12, "quad2nv12");
}
extern "C" int vp_raw2nv12(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, int sampler_mode, int batch, uint8_t* nv12) {
	return quad_colour_dispatch<true, true>(ctx, stream, raw, nullptr, layout, qw, qh, sampler_mode, batch, nv12, "raw2quad+quad2nv12");
}

extern "C" int vp_rgba2nv12(VpContext* ctx, int stream, const uint8_t* rgba, int w, int h, int batch, uint8_t* nv12) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(rgba && nv12 && w > 0 && h > 0 && batch > 0 && batch <= 65535, "vp_rgba2nv12: bad arguments");
	VP_REQUIRE((w & 1) == 0 && (reinterpret_cast<uintptr_t>(rgba) & 7) == 0 && (reinterpret_cast<uintptr_t>(nv12) & 1) == 0, "vp_rgba2nv12: needs an even width, 8-byte aligned input");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "rgba2nv12");
	k_rgba2nv12<<<dim3(vp_div_up(w / 2, 64), vp_div_up((h + 1) / 2, 4), batch), dim3(64, 4), 0, ctx->streams[stream]>>>(reinterpret_cast<const uint32_t*>(rgba), nv12, w, h, vp_nv12_bytes(w, h));
	return vp_launch_ok("rgba2nv12");
}

extern "C" int vp_f2nv12(VpContext* ctx, int stream, const float* img, int w, int h, int batch, uint8_t* nv12) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(img && nv12 && w > 0 && h > 0 && batch > 0 && batch <= 65535, "vp_f2nv12: bad arguments");
	VP_REQUIRE((w & 1) == 0 && (reinterpret_cast<uintptr_t>(img) & 7) == 0 && (reinterpret_cast<uintptr_t>(nv12) & 1) == 0, "vp_f2nv12: needs an even width, 8-byte aligned input");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "f2nv12");
	k_f2nv12<<<dim3(vp_div_up(w / 2, 64), vp_div_up((h + 1) / 2, 4), batch), dim3(64, 4), 0, ctx->streams[stream]>>>(img, nv12, w, h, vp_nv12_bytes(w, h));
	return vp_launch_ok("f2nv12");
}
