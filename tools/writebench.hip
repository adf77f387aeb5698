// Which WRITE pattern does a large hipMalloc block take fastest?  (tools only)
// hipMemsetAsync fills every block at 6.4-6.7 TB/s, while the library's probe (nt 16-byte stores, grid-stride over 4096 workgroups) sees 3.8-4.4 or
// 5.0-5.3 TB/s depending on the block.  This sweeps store flavour x addressing x grid size over a number of blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// MODE 0: nt stores, 1: plain stores, 2: sc1 (write-through) stores
template<int MODE>
__device__ __forceinline__ void st(v4u* p, v4u v) {
	if(MODE == 0) __builtin_nontemporal_store(v, p);
	else if(MODE == 1) *p = v;
	else {
		__amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, 16, 0x00020000);
		__builtin_amdgcn_raw_buffer_store_b128(v, rs, 0, 0, 16);
	}
}
// grid-stride: thread t writes elements t, t + T, t + 2T ...   (T = all threads)
template<int MODE>
__global__ __launch_bounds__(256) void k_gridstride(v4u* out, size_t n16) {
	for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) st<MODE>(out + i, v4u{0u, 0u, 0u, 0u});
}
// block-contiguous: workgroup b owns [b * chunk, (b + 1) * chunk) and sweeps it 4 KB at a time
template<int MODE>
__global__ __launch_bounds__(256) void k_blockchunk(v4u* out, size_t n16) {
	const size_t chunk = (n16 + gridDim.x - 1) / gridDim.x;
	const size_t b0 = (size_t)blockIdx.x * chunk, b1 = b0 + chunk < n16 ? b0 + chunk : n16;
	for(size_t i = b0 + threadIdx.x; i < b1; i += 256) st<MODE>(out + i, v4u{0u, 0u, 0u, 0u});
}
// grid-stride, four consecutive 4 KB pieces per workgroup and step (16 KB contiguous per workgroup)
template<int MODE>
__global__ __launch_bounds__(256) void k_gridstride4(v4u* out, size_t n16) {
	for(size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) {
#pragma unroll
		for(int k = 0; k < 4; k++)
			if(i + 256 * k < n16) st<MODE>(out + i + 256 * k, v4u{0u, 0u, 0u, 0u});
	}
}

// tile writes as the single-pass SAT makes them: workgroup = one TW x TH int32 tile (16 KB) of one of `B` frames of W x H, rows W * 4 bytes apart,
// consecutive workgroups on different frames (the SAT's persistent workgroups serve frame b % B)
template<int TW, int TH, int MODE>
__global__ __launch_bounds__(256) void k_tilewrite(v4u* out, int w4, int h, int tilesX, int tilesY, int B) {
	constexpr int L = TW / 4, RPP = 256 / L;
	const int b = blockIdx.x, perFrame = tilesX * tilesY;
	const int f = b % B, rem = b / B;
	// anti-diagonal-ish scatter is not needed for the point: a frame's tiles are visited in raster order, frames interleaved
	const int ti = rem / tilesX, tj = rem % tilesX;
	const size_t base = (size_t)f * w4 * h + (size_t)(ti * TH) * w4 + (size_t)tj * L;
	const int c = threadIdx.x % L, r0 = threadIdx.x / L;
#pragma unroll
	for(int p = 0; p < TH / RPP; p++) st<MODE>(out + base + (size_t)(r0 + p * RPP) * w4 + c, v4u{0u, 0u, 0u, 0u});
	(void)perFrame;
}

template<typename F>
static float timeit(F f) {
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	f();
	CK(hipEventRecord(e0));
	for(int i = 0; i < 3; i++) f();
	CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
	return ms / 3;
}

int main(int argc, char** argv) {
	const int nblocks = argc > 1 ? atoi(argv[1]) : 10;
	const size_t bytes = (size_t)128 * 2048 * 1536 * 4, n16 = bytes / 16;
	std::vector<void*> blocks;
	printf("%-4s %8s | %27s | %27s | %27s | %17s | %17s\n", "blk", "memset", "gridstride g=4096 nt/pl/sc1", "gridstride g=1024 nt/pl/sc1", "gridstride g=16384 nt/pl/sc1", "chunk g=4096 nt/pl", "stride4 g=4096 nt/pl");
	for(int b = 0; b < nblocks; b++) {
		void* p; CK(hipMalloc(&p, bytes)); blocks.push_back(p);
		v4u* o = static_cast<v4u*>(p);
		auto gbs = [&](float ms) { return bytes / ms / 1e6; };
		const float ms0 = timeit([&] { CK(hipMemsetAsync(p, 0, bytes, 0)); });
		float r[3][3];
		const int grids[3] = {4096, 1024, 16384};
		for(int gi = 0; gi < 3; gi++) {
			const int g = grids[gi];
			r[gi][0] = gbs(timeit([&] { k_gridstride<0><<<g, 256>>>(o, n16); }));
			r[gi][1] = gbs(timeit([&] { k_gridstride<1><<<g, 256>>>(o, n16); }));
			r[gi][2] = gbs(timeit([&] { k_gridstride<2><<<g, 256>>>(o, n16); }));
		}
		const float c0 = gbs(timeit([&] { k_blockchunk<0><<<4096, 256>>>(o, n16); })), c1 = gbs(timeit([&] { k_blockchunk<1><<<4096, 256>>>(o, n16); }));
		const float s0 = gbs(timeit([&] { k_gridstride4<0><<<4096, 256>>>(o, n16); })), s1 = gbs(timeit([&] { k_gridstride4<1><<<4096, 256>>>(o, n16); }));
		printf("%-4d %8.0f | %8.0f %8.0f %8.0f | %8.0f %8.0f %8.0f | %8.0f %8.0f %8.0f | %8.0f %8.0f | %8.0f %8.0f\n", b, gbs(ms0), r[0][0], r[0][1], r[0][2], r[1][0], r[1][1], r[1][2], r[2][0], r[2][1], r[2][2], c0, c1, s0, s1);
		{
			const int W = 2048, H = 1536, B = 128;
			float t[5][2];
#define TILE(I, TW_, TH_) \
			t[I][0] = gbs(timeit([&] { k_tilewrite<TW_, TH_, 0><<<(W / TW_) * (H / TH_) * B, 256>>>(o, W / 4, H, W / TW_, H / TH_, B); })); \
			t[I][1] = gbs(timeit([&] { k_tilewrite<TW_, TH_, 1><<<(W / TW_) * (H / TH_) * B, 256>>>(o, W / 4, H, W / TW_, H / TH_, B); }));
			TILE(0, 64, 64) TILE(1, 128, 32) TILE(2, 256, 16) TILE(3, 512, 8) TILE(4, 1024, 4)
#undef TILE
			printf("      tiles (16 KB per workgroup, 128 frames interleaved) nt/plain:  64x64 %5.0f %5.0f | 128x32 %5.0f %5.0f | 256x16 %5.0f %5.0f | 512x8 %5.0f %5.0f | 1024x4 %5.0f %5.0f\n",
					t[0][0], t[0][1], t[1][0], t[1][1], t[2][0], t[2][1], t[3][0], t[3][1], t[4][0], t[4][1]);
		}
		fflush(stdout);
	}
	for(void* p : blocks) CK(hipFree(p));
	return 0;
}
