// Access-pattern micro-benchmark (not part of the product): copy a batch of W x H uint32 images tile by tile,
// one workgroup per TWxTH tile, rows of a tile are W*4 bytes apart.  Tells what HBM bandwidth a tiled SAT can hope for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

template<int TW, int TH>
__global__ __launch_bounds__(256) void k_tilecopy(const uint4* __restrict__ in, uint4* __restrict__ out, int w4, int h, int tilesX, int tilesY, int order) {
	constexpr int L = TW / 4;              // uint4 per tile row
	constexpr int RPP = 256 / L;           // rows per pass
	int b = blockIdx.x;
	const int perFrame = tilesX * tilesY;
	int f, rem;
	if(order == 0) { f = b / perFrame; rem = b % perFrame; }            // frame-major
	else { const int B = gridDim.x / perFrame; f = b % B; rem = b / B; } // frame-interleaved (b % B)
	const int ti = rem / tilesX, tj = rem % tilesX;
	const size_t base = (size_t)f * w4 * h + (size_t)(ti * TH) * w4 + tj * L;
	const int c = threadIdx.x % L, r0 = threadIdx.x / L;
	uint4 v[TH / RPP];
#pragma unroll
	for(int p = 0; p < TH / RPP; p++) v[p] = in[base + (size_t)(r0 + p * RPP) * w4 + c];
#pragma unroll
	for(int p = 0; p < TH / RPP; p++) { v[p].x += 1; out[base + (size_t)(r0 + p * RPP) * w4 + c] = v[p]; }
}

template<int TW, int TH>
void run(const uint4* in, uint4* out, int W, int H, int B, int order) {
	const int tilesX = W / TW, tilesY = H / TH;
	const int grid = tilesX * tilesY * B;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for(int i = 0; i < 3; i++) k_tilecopy<TW, TH><<<grid, 256>>>(in, out, W / 4, H, tilesX, tilesY, order);
	CK(hipEventRecord(e0));
	const int N = 20;
	for(int i = 0; i < N; i++) k_tilecopy<TW, TH><<<grid, 256>>>(in, out, W / 4, H, tilesX, tilesY, order);
	CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= N;
	const double bytes = 2.0 * W * H * 4.0 * B;
	printf("tile %4dx%-3d order %d B=%d: %8.1f us  %7.0f GB/s (read+write)\n", TW, TH, order, B, ms * 1e3, bytes / ms / 1e6);
}

// read-only and write-only streams of the same footprint (one 16-byte access per thread, grid-stride): what a pure consumer (the circle + blobList
// walker reads the table once) or a pure producer can hope for
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
	uint4 acc = {0, 0, 0, 0};
	for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
		typedef unsigned int v4u __attribute__((ext_vector_type(4)));
		const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(in) + i);
		acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
	}
	if((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc; // never true for the memset pattern: keeps the loads alive
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ out, size_t n16) {
	for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
		typedef unsigned int v4u __attribute__((ext_vector_type(4)));
		__builtin_nontemporal_store(v4u{1, 2, 3, (unsigned)i}, reinterpret_cast<v4u*>(out) + i);
	}
}
static void run_stream(const uint4* in, uint4* out, size_t bytes, int blocks_per_cu) {
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int grid = 256 * blocks_per_cu, N = 20;
	for(int w = 0; w < 2; w++) {
		for(int i = 0; i < 3; i++) { if(w) k_write<<<grid, 256>>>(out, bytes / 16); else k_read<<<grid, 256>>>(in, out, bytes / 16); }
		CK(hipEventRecord(e0));
		for(int i = 0; i < N; i++) { if(w) k_write<<<grid, 256>>>(out, bytes / 16); else k_read<<<grid, 256>>>(in, out, bytes / 16); }
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= N;
		printf("%s only, %d workgroups per CU, %.0f MB: %8.1f us  %7.0f GB/s\n", w ? "write" : "read ", blocks_per_cu, bytes / 1e6, ms * 1e3, bytes / ms / 1e6);
	}
}

int main(int argc, char** argv) {
	const int W = 2048, H = 1536, B = argc > 1 ? atoi(argv[1]) : 32;
	uint4 *in, *out;
	const size_t n = (size_t)W * H * 4 * B;
	CK(hipMalloc(&in, n)); CK(hipMalloc(&out, n));
	CK(hipMemset(in, 1, n)); CK(hipMemset(out, 0, n));
	for(int order = 0; order < 2; order++) {
		run<64, 64>(in, out, W, H, B, order);
		run<128, 32>(in, out, W, H, B, order);
		run<256, 16>(in, out, W, H, B, order);
		run<512, 8>(in, out, W, H, B, order);
		run<1024, 4>(in, out, W, H, B, order);
	}
	for(int bpc : {4, 8, 16, 32}) run_stream(in, out, n, bpc);
	return 0;
}
