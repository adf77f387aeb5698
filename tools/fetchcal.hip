// FETCH_SIZE calibration per access class (tools only): rocprofv3 --pmc FETCH_SIZE on gfx950 reports half of the bytes of a 16 B/lane
// streaming read (MI355X_MICROARCH.md, HBM); the single-pass SAT also issues dword loads (halo strips), 8-byte agent-scope granule loads and a scalar order
// table.  Each kernel below reads a known number of DISTINCT bytes once, from a buffer far larger than the Infinity Cache, so that
// FETCH_SIZE / bytes is the counter's factor for that class.   usage: fetchcal [MiB = 2048]
//   k_r16    16 B per lane, a wavefront reads 1 KB contiguous                 (tile rows, the reference class: 0.5)
//   k_r8     8 B per lane, 512 B contiguous per wavefront
//   k_r4     4 B per lane, 256 B contiguous per wavefront                     (halo strips, halo columns)
//   k_g8     8 B per lane agent-scope atomic loads (sc1), 512 B per wavefront  (carry granules)
//   k_r4s    4 B per lane, 16 lanes x 4 B = 64 B used of every 128 B          (half-used strip records: FIRST | LAST halves)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_r16(const v4u* __restrict__ in, unsigned* out, size_t n) {
	unsigned acc = 0;
	for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const v4u v = __builtin_nontemporal_load(in + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
	if(acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_r8(const v2u* __restrict__ in, unsigned* out, size_t n) {
	unsigned acc = 0;
	for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const v2u v = __builtin_nontemporal_load(in + i); acc ^= v.x ^ v.y; }
	if(acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_r4(const unsigned* __restrict__ in, unsigned* out, size_t n) {
	unsigned acc = 0;
	for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= __builtin_nontemporal_load(in + i);
	if(acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_g8(const unsigned long long* __restrict__ in, unsigned* out, size_t n) {
	unsigned long long acc = 0;
	typedef __attribute__((address_space(1))) unsigned long long gu64;
	for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= __hip_atomic_load((gu64*)(in + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if(acc == 0x12345678u) out[0] = (unsigned)acc;
}
// every lane group of 16 reads the first 64 B of its own 128 B line: n = number of 128-byte lines
__global__ __launch_bounds__(256) void k_r4s(const unsigned* __restrict__ in, unsigned* out, size_t nlines) {
	unsigned acc = 0;
	const int sub = threadIdx.x & 15;
	for(size_t l = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4; l < nlines; l += ((size_t)gridDim.x * 256) >> 4) acc ^= in[l * 32 + sub];
	if(acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
	const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 2048, bytes = mib << 20;
	void* buf; unsigned* out;
	CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
	CK(hipMemset(buf, 1, bytes));
	const int grid = 256 * 8;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto timed = [&](const char* name, double useful, auto launch) {
		launch(); CK(hipDeviceSynchronize());
		CK(hipEventRecord(e0)); for(int i = 0; i < 3; i++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
		printf("%-6s %10.1f MB read per launch  %8.1f us  %7.0f GB/s\n", name, useful / 1e6, ms * 1e3, useful / ms / 1e6);
	};
	timed("k_r16", (double)bytes, [&] { k_r16<<<grid, 256>>>((const v4u*)buf, out, bytes / 16); });
	timed("k_r8", (double)bytes, [&] { k_r8<<<grid, 256>>>((const v2u*)buf, out, bytes / 8); });
	timed("k_r4", (double)bytes, [&] { k_r4<<<grid, 256>>>((const unsigned*)buf, out, bytes / 4); });
	timed("k_g8", (double)bytes, [&] { k_g8<<<grid, 256>>>((const unsigned long long*)buf, out, bytes / 8); });
	timed("k_r4s", (double)bytes / 2, [&] { k_r4s<<<grid, 256>>>((const unsigned*)buf, out, bytes / 128); });
	return 0;
}
