// Issue rate of the VALU instructions the resampling kernel is made of (tools only): one wavefront per SIMD and eight, a long unrolled run of INDEPENDENT
// instructions of one kind, cycles per wave instruction from s_memtime.   hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template<int KIND>
__global__ __launch_bounds__(256) void k(uint64_t* out, float seed, int iters) {
	f2 a0 = {seed, seed + 1}, a1 = {seed + 2, seed}, a2 = {seed, 3}, a3 = {1, seed}, w = {seed * 0.5f, seed * 0.25f};
	f2 r0 = a0, r1 = a1, r2 = a2, r3 = a3;
	float s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
	uint32_t u0 = (uint32_t)seed, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
	const uint64_t t0 = __builtin_readcyclecounter();
	for(int i = 0; i < iters; i++) {
		if(KIND == 0) { REP64(asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(seed));) }
		if(KIND == 1) { REP64(asm volatile("v_pk_mul_f32 %0, %4, %0\n v_pk_mul_f32 %1, %4, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_mul_f32 %3, %4, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(w));) }
		if(KIND == 2) { REP64(asm volatile("v_pk_mul_f32 %0, %4, %0 op_sel:[0,0] op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %4, %1 op_sel:[1,0] op_sel_hi:[1,1]\n v_pk_mul_f32 %2, %4, %2 op_sel:[0,0] op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %4, %3 op_sel:[1,0] op_sel_hi:[1,1]" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(w));) }
		if(KIND == 3) { REP64(asm volatile("v_pk_add_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_add_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(w));) }
		if(KIND == 4) { REP64(asm volatile("v_pk_fma_f32 %0, %4, %0, %4\n v_pk_fma_f32 %1, %4, %1, %4\n v_pk_fma_f32 %2, %4, %2, %4\n v_pk_fma_f32 %3, %4, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(w));) }
		if(KIND == 5) { REP64(asm volatile("v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(seed));) }
		if(KIND == 6) { REP64(asm volatile("v_cvt_u32_f32 %0, %4\n v_cvt_u32_f32 %1, %4\n v_cvt_u32_f32 %2, %4\n v_cvt_u32_f32 %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(seed));) }
		if(KIND == 7) { REP64(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0));) }
		if(KIND == 8) { REP64(asm volatile("v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %1, %1, %2, %4\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %0, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(0x0c0c0501));) }
		if(KIND == 9) { REP64(asm volatile("v_cvt_f32_ubyte1 %0, %4\n v_cvt_f32_ubyte2 %1, %4\n v_cvt_f32_ubyte3 %2, %4\n v_cvt_f32_ubyte0 %3, %4" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(u0));) }
		if(KIND == 10) { REP64(asm volatile("v_add_lshl_u32 %0, %0, %1, 6\n v_add_lshl_u32 %1, %1, %2, 6\n v_add_lshl_u32 %2, %2, %3, 6\n v_add_lshl_u32 %3, %3, %0, 6" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
		if(KIND == 11) { REP64(asm volatile("v_pk_add_f32 %0, %4, %0\n s_nop 0\n v_pk_add_f32 %0, %4, %0\n s_nop 0\n v_pk_add_f32 %0, %4, %0\n s_nop 0\n v_pk_add_f32 %0, %4, %0\n s_nop 0" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(w));) } // dependent chain
		if(KIND == 12) { REP64(asm volatile("v_add_f32 %0, %4, %0\n v_add_f32 %0, %4, %0\n v_add_f32 %0, %4, %0\n v_add_f32 %0, %4, %0" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(seed));) } // dependent chain
		if(KIND == 13) { REP64(asm volatile("v_trunc_f32 %0, %4\n v_trunc_f32 %1, %4\n v_trunc_f32 %2, %4\n v_trunc_f32 %3, %4" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(seed));) }
		if(KIND == 14) { REP64(asm volatile("v_fma_f32 %0, %4, %0, %4\n v_fma_f32 %1, %4, %1, %4\n v_fma_f32 %2, %4, %2, %4\n v_fma_f32 %3, %4, %3, %4" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(seed));) }
		if(KIND == 15) { REP64(asm volatile("v_cvt_pk_u8_f32 %0, %4, 0, %0\n v_cvt_pk_u8_f32 %1, %4, 1, %1\n v_cvt_pk_u8_f32 %2, %4, 2, %2\n v_cvt_pk_u8_f32 %3, %4, 3, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(seed));) }
		if(KIND == 16) { REP64(asm volatile("v_floor_f32 %0, %4\n v_floor_f32 %1, %4\n v_floor_f32 %2, %4\n v_floor_f32 %3, %4" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(seed));) }
		if(KIND == 17) { REP64(asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0));) }
		if(KIND == 18) { REP64(asm volatile("v_max_f32 %0, %4, %0\n v_max_f32 %1, %4, %1\n v_max_f32 %2, %4, %2\n v_max_f32 %3, %4, %3" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(seed));) }
		if(KIND == 19) { REP64(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(seed));) }
		if(KIND == 20) { REP64(asm volatile("v_and_b32 %0, %4, %0\n v_and_b32 %1, %4, %1\n v_and_b32 %2, %4, %2\n v_and_b32 %3, %4, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0));) }
		if(KIND == 21) { REP64(asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
		if(KIND == 22) { REP64(asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %4\n v_cvt_f32_u32 %2, %4\n v_cvt_f32_u32 %3, %4" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(u0));) }
	}
	const uint64_t t1 = __builtin_readcyclecounter();
	if(threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
	if(seed == 12345.f) out[0] = (uint64_t)(r0.x + r1.y + r2.x + r3.y + s0 + s1 + s2 + s3) + u0 + u1 + u2 + u3;
}
template<int KIND> void run(const char* name, uint64_t* d, int wg_per_cu, int threads) {
	const int iters = 256, blocks = 256 * wg_per_cu;
	for(int rep = 0; rep < 2; rep++) {
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 1.5f, iters);
		hipDeviceSynchronize();
	}
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 1.5f, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	std::vector<uint64_t> h(blocks * threads / 64);
	hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
	double sum = 0; for(auto v : h) sum += (double)v;
	const double n_inst = (double)iters * 64 * 4, waves_per_simd = (double)wg_per_cu * threads / 64 / 4;
	// s_memtime counts at a constant 100 MHz; wall-clock per instruction and SIMD is the robust figure
	printf("%-44s %d waves/SIMD: %7.3f ns per wave-instruction and SIMD (launch %.3f ms, memtime ticks per wave %.0f)\n", name, (int)waves_per_simd, ms * 1e6 / (n_inst * waves_per_simd), ms, sum / h.size());
}
__global__ void k_round(uint32_t* o) {
	const float v[8] = {0.25f, 0.5f, 0.75f, 1.5f, 2.5f, 254.75f, 255.5f, 300.f};
	uint32_t r = 0;
	asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(v[threadIdx.x & 7]));
	o[threadIdx.x] = r;
}
int main() {
	uint64_t* d; hipMalloc(&d, 1 << 22);
	{
		hipLaunchKernelGGL(k_round, dim3(1), dim3(8), 0, 0, (uint32_t*)d);
		uint32_t h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
		printf("v_cvt_pk_u8_f32 of 0.25 0.5 0.75 1.5 2.5 254.75 255.5 300: %u %u %u %u %u %u %u %u\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
	}
	for(int occ = 0; occ < 2; occ++) {
		const int wg = occ ? 8 : 1;
#define R(K, N) run<K>(N, d, wg, 256);
		R(0, "v_mul_f32") R(5, "v_add_f32") R(1, "v_pk_mul_f32") R(2, "v_pk_mul_f32 op_sel (weight half)") R(3, "v_pk_add_f32") R(4, "v_pk_fma_f32")
		R(6, "v_cvt_u32_f32") R(9, "v_cvt_f32_ubyteN") R(7, "v_mad_u32_u24") R(8, "v_perm_b32") R(10, "v_add_lshl_u32") R(11, "v_pk_add_f32 dependent + s_nop") R(12, "v_add_f32 dependent")
		R(13, "v_trunc_f32") R(16, "v_floor_f32") R(14, "v_fma_f32") R(18, "v_max_f32") R(15, "v_cvt_pk_u8_f32") R(22, "v_cvt_f32_u32") R(17, "v_add_u32") R(20, "v_and_b32") R(21, "v_lshlrev_b32") R(19, "v_mov_b32")
	}
	return 0;
}
