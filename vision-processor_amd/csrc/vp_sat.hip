// gradientDot + summed-area table.
//   gradient_dotproduct (gradientDot.cl:22-30, Resources.cpp:160)
//   sat_horizontal      (satHorizontal.cl:22-32, Resources.cpp:161)
//   sat_vertical        (satVertical.cl:22-32, Resources.cpp:162)
//
// gradDot values are exact integers (|v| <= 3*255^2), so the table is built in wrapping int32: box sums
// taken from it are exact and equal the reference's binary32 table whenever max|SAT| < 2^24.
//
// This file holds the standalone gradDot kernel and the TWOPASS ("global-recursive") SAT: a row pass that fuses the
// gradient stencil with a block-wide row prefix, then an in-place column pass.  The single-pass LDS-tiled variant
// lives in vp_sat_tiled.hip.
#include "vp_common.h"

// clamp-to-edge tap of the RGBA8 flat image (gradientDot.cl:20)
__device__ __forceinline__ uint32_t flat_tap(const uint32_t* __restrict__ f, int w, int h, int x, int y) {
	return f[(size_t)vp_clampi(y, 0, h - 1) * w + vp_clampi(x, 0, w - 1)];
}

// sum over r,g,b of (xp - xn) * (yp - yn); alpha is 255 on all taps and contributes 0
__device__ __forceinline__ int grad_dot(uint32_t xp, uint32_t xn, uint32_t yp, uint32_t yn) {
	int acc = 0;
#pragma unroll
	for(int k = 0; k < 3; k++) {
		const int gx = (int)((xp >> (8 * k)) & 0xFF) - (int)((xn >> (8 * k)) & 0xFF);
		const int gy = (int)((yp >> (8 * k)) & 0xFF) - (int)((yn >> (8 * k)) & 0xFF);
		acc += gx * gy;
	}
	return acc;
}

__global__ __launch_bounds__(256) void k_graddot(const uint32_t* __restrict__ flat, float* __restrict__ out, int w, int h, int o) {
	const int x = blockIdx.x * 64 + threadIdx.x;
	const int y = blockIdx.y * 4 + threadIdx.y;
	if(x >= w || y >= h) return;
	const uint32_t* f = flat + (size_t)blockIdx.z * w * h;
	const int g = grad_dot(flat_tap(f, w, h, x + o, y), flat_tap(f, w, h, x - o, y), flat_tap(f, w, h, x, y + o), flat_tap(f, w, h, x, y - o));
	out[(size_t)blockIdx.z * w * h + (size_t)y * w + x] = (float)g;
}

extern "C" int vp_graddot(VpContext* ctx, int stream, const uint8_t* flat, int w, int h, int offset, int batch, float* graddot) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(flat && graddot && w > 0 && h > 0 && batch > 0 && batch <= 65535 && offset >= 0, "vp_graddot: bad arguments");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "gradientDot");
	k_graddot<<<dim3(vp_div_up(w, 64), vp_div_up(h, 4), batch), dim3(64, 4), 0, ctx->streams[stream]>>>(reinterpret_cast<const uint32_t*>(flat), graddot, w, h, offset);
	return vp_launch_ok("gradientDot");
}

// ---------------------------------------------------------------- TWOPASS: row pass --------------------------------
// One 256-thread block per image row.  Each thread owns 4 consecutive pixels of a 1024-pixel chunk: local prefix,
// wave-64 shuffle scan, 4-wave LDS combine, running carry between chunks.
template<bool FUSED>
__global__ __launch_bounds__(256) void k_sat_rowpass(const void* __restrict__ in, int32_t* __restrict__ sat, float* __restrict__ gd_out, int w, int h, int o) {
	__shared__ int wave_tot[4];
	const int y = blockIdx.x;
	const size_t frame = (size_t)blockIdx.y * w * h;
	const uint32_t* flat = reinterpret_cast<const uint32_t*>(in) + frame;
	const float* gin = reinterpret_cast<const float*>(in) + frame;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	int carry = 0;
	for(int x0 = 0; x0 < w; x0 += 1024) {
		const int xb = x0 + tid * 4;
		int g[4];
#pragma unroll
		for(int i = 0; i < 4; i++) {
			const int x = xb + i;
			if(x < w) {
				if(FUSED) g[i] = grad_dot(flat_tap(flat, w, h, x + o, y), flat_tap(flat, w, h, x - o, y), flat_tap(flat, w, h, x, y + o), flat_tap(flat, w, h, x, y - o));
				else g[i] = (int)gin[(size_t)y * w + x];
			} else g[i] = 0;
		}
		if(FUSED && gd_out) {
#pragma unroll
			for(int i = 0; i < 4; i++)
				if(xb + i < w) gd_out[frame + (size_t)y * w + xb + i] = (float)g[i];
		}
		g[1] += g[0];
		g[2] += g[1];
		g[3] += g[2];
		const int incl = vp_wave_scan_incl(g[3]);
		if(lane == 63) wave_tot[wv] = incl;
		__syncthreads();
		int base = carry + incl - g[3];
		int chunk_tot = 0;
#pragma unroll
		for(int k = 0; k < 4; k++) {
			const int t = wave_tot[k];
			if(k < wv) base += t;
			chunk_tot += t;
		}
		__syncthreads();
#pragma unroll
		for(int i = 0; i < 4; i++)
			if(xb + i < w) sat[frame + (size_t)y * w + xb + i] = base + g[i];
		carry += chunk_tot;
	}
}

// ---------------------------------------------------------------- TWOPASS: column pass (in place) ------------------
__global__ __launch_bounds__(256) void k_sat_colpass(int32_t* __restrict__ sat, int w, int h) {
	const int x = blockIdx.x * 256 + threadIdx.x;
	if(x >= w) return;
	int32_t* p = sat + (size_t)blockIdx.y * w * h + x;
	int sum = 0;
	int y = 0;
	for(; y + 8 <= h; y += 8) {
		int v[8];
#pragma unroll
		for(int k = 0; k < 8; k++) v[k] = p[(size_t)(y + k) * w];
#pragma unroll
		for(int k = 0; k < 8; k++) {
			sum += v[k];
			p[(size_t)(y + k) * w] = sum;
		}
	}
	for(; y < h; y++) {
		sum += p[(size_t)y * w];
		p[(size_t)y * w] = sum;
	}
}

int vp_sat_twopass(VpContext* ctx, int stream, const void* in, bool fused, int w, int h, int offset, int batch, int32_t* sat, float* graddot_out) {
	hipStream_t s = ctx->streams[stream];
	if(fused) k_sat_rowpass<true><<<dim3(h, batch), 256, 0, s>>>(in, sat, graddot_out, w, h, offset);
	else k_sat_rowpass<false><<<dim3(h, batch), 256, 0, s>>>(in, sat, nullptr, w, h, 0);
	VP_TRY(vp_launch_ok("sat row pass"));
	k_sat_colpass<<<dim3(vp_div_up(w, 256), batch), 256, 0, s>>>(sat, w, h);
	return vp_launch_ok("sat column pass");
}

int vp_sat_tiled(VpContext* ctx, int stream, const void* in, bool fused, int w, int h, int offset, int batch, int32_t* sat, float* graddot_out, const uint32_t* strips, int32_t* zero3);

// zero3 (vp_frame only): the blob list's counter[3 x batch], cleared by the single-pass kernel's first workgroup per frame so that
// the tail needs no clearing launch; *zeroed tells the caller whether that happened (the two-pass variant does not do it).
int vp_sat_common(VpContext* ctx, int stream, const void* in, bool fused, int w, int h, int offset, int batch, int variant, int32_t* sat, float* gd, const char* name, const uint32_t* strips,
		int32_t* zero3, bool* zeroed) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(in && sat && w > 0 && h > 0 && batch > 0 && batch <= 65535 && offset >= 0, "%s: bad arguments", name);
	VP_REQUIRE(variant == VP_SAT_TILED || variant == VP_SAT_TWOPASS, "%s: unknown SAT variant %d", name, variant);
	VP_REQUIRE((reinterpret_cast<uintptr_t>(in) & 3) == 0 && (reinterpret_cast<uintptr_t>(sat) & 3) == 0, "%s: buffers must be 4-byte aligned", name);
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, name);
	if(zeroed) *zeroed = false;
	if(variant == VP_SAT_TWOPASS) return vp_sat_twopass(ctx, stream, in, fused, w, h, offset, batch, sat, gd);
	if(zeroed) *zeroed = zero3 != nullptr;
	return vp_sat_tiled(ctx, stream, in, fused, w, h, offset, batch, sat, gd, strips, zero3);
}

extern "C" int vp_sat(VpContext* ctx, int stream, const float* graddot, int w, int h, int batch, int variant, int32_t* sat) {
	return vp_sat_common(ctx, stream, graddot, false, w, h, 0, batch, variant, sat, nullptr, "sat", nullptr, nullptr, nullptr);
}

extern "C" int vp_graddot_sat(VpContext* ctx, int stream, const uint8_t* flat, int w, int h, int offset, int batch, int variant, int32_t* sat, float* graddot_out) {
	return vp_sat_common(ctx, stream, flat, true, w, h, offset, batch, variant, sat, graddot_out, "gradientDot+sat", nullptr, nullptr, nullptr);
}

extern "C" int vp_graddot_sat_strips(VpContext* ctx, int stream, const uint8_t* flat, const uint32_t* strips, int w, int h, int offset, int batch, int variant,
		int32_t* sat, float* graddot_out) {
	return vp_sat_common(ctx, stream, flat, true, w, h, offset, batch, variant, sat, graddot_out, "gradientDot+sat", strips, nullptr, nullptr);
}
