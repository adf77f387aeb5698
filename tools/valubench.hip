// VALU issue-rate microbenchmark for gfx950: how many cycles one wave64 instruction of a given kind occupies a SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valubench tools/valubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ inline v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
#define REP8(x) x x x x x x x x
#define KERNEL(name, decl, body, fin)                                                     \
	__global__ __launch_bounds__(256) void name(uint32_t* out, int iters, uint32_t seed) { \
		decl;                                                                             \
		for(int i = 0; i < iters; i++) { REP8(body) }                                     \
		out[blockIdx.x * 256 + threadIdx.x] = fin;                                        \
	}

KERNEL(k_mul_f32, float a0 = seed + threadIdx.x; float a1 = a0 + 1; float a2 = a0 + 2; float a3 = a0 + 3; float m = 1.0001f,
       asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       (uint32_t)(a0 + a1 + a2 + a3))
KERNEL(k_add_u32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       a0 + a1 + a2 + a3)
KERNEL(k_perm, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = 0x03020100u,
       asm volatile("v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %1, %1, %2, %4\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       a0 + a1 + a2 + a3)
KERNEL(k_cvt_ubyte, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = 0,
       asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte2 %2, %2\n v_cvt_f32_ubyte3 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       a0 + a1 + a2 + a3)
KERNEL(k_cvt_u32_f32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = 0,
       asm volatile("v_cvt_u32_f32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_f32_u32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       a0 + a1 + a2 + a3)
KERNEL(k_pk_mul_f32, v2f a0 = mk2((float)seed, (float)threadIdx.x); v2f a1 = a0 + 1.f; v2f a2 = a0 + 2.f; v2f a3 = a0 + 3.f; v2f m = mk2(1.0001f, 1.0002f),
       asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       (uint32_t)(a0.x + a1.y + a2.x + a3.y))
KERNEL(k_mad_u32_u24, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = 3,
       asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       a0 + a1 + a2 + a3)
KERNEL(k_add3, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = 3,
       asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       a0 + a1 + a2 + a3)
KERNEL(k_mul_lo_u32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = 3,
       asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       a0 + a1 + a2 + a3)
KERNEL(k_dpp_add, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = 3,
       asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       a0 + a1 + a2 + a3)
KERNEL(k_pk_sub_i16, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = 3,
       asm volatile("v_pk_sub_i16 %0, %0, %4\n v_pk_sub_i16 %1, %1, %4\n v_pk_sub_i16 %2, %2, %4\n v_pk_sub_i16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       a0 + a1 + a2 + a3)
KERNEL(k_dot2c, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = 3,
       asm volatile("v_dot2c_i32_i16 %0, %1, %4\n v_dot2c_i32_i16 %1, %2, %4\n v_dot2c_i32_i16 %2, %3, %4\n v_dot2c_i32_i16 %3, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));,
       a0 + a1 + a2 + a3)
KERNEL(k_v_add_f32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_sub_u32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_and_b32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_or_b32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_lshrrev_b32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_lshlrev_b32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_min_u32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_max_i32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_mov_b32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_fma_f32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_fmac_f32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_fmac_f32 %0, %0, %4\n v_fmac_f32 %1, %1, %4\n v_fmac_f32 %2, %2, %4\n v_fmac_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_lshl_or_b32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_lshl_or_b32 %0, %0, 8, %4\n v_lshl_or_b32 %1, %1, 8, %4\n v_lshl_or_b32 %2, %2, 8, %4\n v_lshl_or_b32 %3, %3, 8, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_lshl_add_u32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_and_or_b32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_and_or_b32 %0, %0, %4, %0\n v_and_or_b32 %1, %1, %4, %1\n v_and_or_b32 %2, %2, %4, %2\n v_and_or_b32 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_or3_b32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_or3_b32 %0, %0, %4, %0\n v_or3_b32 %1, %1, %4, %1\n v_or3_b32 %2, %2, %4, %2\n v_or3_b32 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_bfe_u32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_alignbit_b32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_alignbit_b32 %0, %0, %4, 8\n v_alignbit_b32 %1, %1, %4, 8\n v_alignbit_b32 %2, %2, %4, 8\n v_alignbit_b32 %3, %3, %4, 8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_cndmask_b32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_med3_i32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_med3_i32 %0, %0, %4, %0\n v_med3_i32 %1, %1, %4, %1\n v_med3_i32 %2, %2, %4, %2\n v_med3_i32 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_mul_u32_u24, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_add_co_u32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %4\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_cvt_f32_i32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_floor_f32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_rcp_f32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_mul_f32_2src, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_mul_f32 %0, %1, %4\n v_mul_f32 %1, %2, %4\n v_mul_f32 %2, %3, %4\n v_mul_f32 %3, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_add_u32_2src, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_add_u32 %0, %1, %4\n v_add_u32 %1, %2, %4\n v_add_u32 %2, %3, %4\n v_add_u32 %3, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_perm_b32_1src, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_perm_b32 %0, %0, %0, %4\n v_perm_b32 %1, %1, %1, %4\n v_perm_b32 %2, %2, %2, %4\n v_perm_b32 %3, %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_xor_b32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_add_u32_e64, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_add_u32_e64 %0, %0, %4\n v_add_u32_e64 %1, %1, %4\n v_add_u32_e64 %2, %2, %4\n v_add_u32_e64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_sad_u8, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_sad_u8 %0, %0, %4, %0\n v_sad_u8 %1, %1, %4, %1\n v_sad_u8 %2, %2, %4, %2\n v_sad_u8 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
KERNEL(k_v_cmp_lt_u32, uint32_t a0 = seed + threadIdx.x; uint32_t a1 = a0 + 1; uint32_t a2 = a0 + 2; uint32_t a3 = a0 + 3; uint32_t m = seed,
       asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");,
       a0 + a1 + a2 + a3)
template<class K>
static void run(const char* name, K k, int wg_per_cu, int ninstr_per_iter) {
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount, blocks = cus * wg_per_cu, iters = 4000;
	uint32_t* out;
	hipMalloc(&out, (size_t)blocks * 256 * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	k<<<blocks, 256>>>(out, 100, 1);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	k<<<blocks, 256>>>(out, iters, 1);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	// wave-instructions per SIMD: each WG has 4 waves = one per SIMD -> wg_per_cu waves per SIMD
	const double inst_per_simd = (double)wg_per_cu * iters * ninstr_per_iter;
	printf("%-16s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles @2.4 GHz)\n", name, wg_per_cu, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
	hipFree(out);
}

int main() {
	for(int occ : {8}) {
		run("v_mul_f32", k_mul_f32, occ, 32);
		run("v_add_u32", k_add_u32, occ, 32);
		run("v_perm_b32", k_perm, occ, 32);
		run("v_cvt_f32_ubyte", k_cvt_ubyte, occ, 32);
		run("v_cvt_u32/f32", k_cvt_u32_f32, occ, 32);
		run("v_pk_mul_f32", k_pk_mul_f32, occ, 32);
		run("v_mad_u32_u24", k_mad_u32_u24, occ, 32);
		run("v_add3_u32", k_add3, occ, 32);
		run("v_mul_lo_u32", k_mul_lo_u32, occ, 32);
		run("v_add_u32_dpp", k_dpp_add, occ, 32);
		run("v_pk_sub_i16", k_pk_sub_i16, occ, 32);
		run("v_dot2c_i32_i16", k_dot2c, occ, 32);
		run("v_add_f32", k_v_add_f32, occ, 32);
		run("v_sub_u32", k_v_sub_u32, occ, 32);
		run("v_and_b32", k_v_and_b32, occ, 32);
		run("v_or_b32", k_v_or_b32, occ, 32);
		run("v_lshrrev_b32", k_v_lshrrev_b32, occ, 32);
		run("v_lshlrev_b32", k_v_lshlrev_b32, occ, 32);
		run("v_min_u32", k_v_min_u32, occ, 32);
		run("v_max_i32", k_v_max_i32, occ, 32);
		run("v_mov_b32", k_v_mov_b32, occ, 32);
		run("v_fma_f32", k_v_fma_f32, occ, 32);
		run("v_fmac_f32", k_v_fmac_f32, occ, 32);
		run("v_lshl_or_b32", k_v_lshl_or_b32, occ, 32);
		run("v_lshl_add_u32", k_v_lshl_add_u32, occ, 32);
		run("v_and_or_b32", k_v_and_or_b32, occ, 32);
		run("v_or3_b32", k_v_or3_b32, occ, 32);
		run("v_bfe_u32", k_v_bfe_u32, occ, 32);
		run("v_alignbit_b32", k_v_alignbit_b32, occ, 32);
		run("v_cndmask_b32", k_v_cndmask_b32, occ, 32);
		run("v_med3_i32", k_v_med3_i32, occ, 32);
		run("v_mul_u32_u24", k_v_mul_u32_u24, occ, 32);
		run("v_add_co_u32", k_v_add_co_u32, occ, 32);
		run("v_cvt_f32_i32", k_v_cvt_f32_i32, occ, 32);
		run("v_floor_f32", k_v_floor_f32, occ, 32);
		run("v_rcp_f32", k_v_rcp_f32, occ, 32);
		run("v_mul_f32_2src", k_v_mul_f32_2src, occ, 32);
		run("v_add_u32_2src", k_v_add_u32_2src, occ, 32);
		run("v_perm_b32_1src", k_v_perm_b32_1src, occ, 32);
		run("v_xor_b32", k_v_xor_b32, occ, 32);
		run("v_add_u32_e64", k_v_add_u32_e64, occ, 32);
		run("v_sad_u8", k_v_sad_u8, occ, 32);
		run("v_cmp_lt_u32", k_v_cmp_lt_u32, occ, 32);
	}
	return 0;
}
