/*
 * tests/shim_probe.c -- TEST HELPER for tests/test_ref_shim.py.
 *
 * oracle/ref_shim.c defines the OpenCL C builtins the reference's compiled kernels leave undefined.  Their arguments are clang vector
 * types (int2, float2, uint4) that travel in SSE registers, which ctypes cannot form; this file gives every builtin a door with plain
 * scalars and pointers.  The builtins are reached by the SAME Itanium-mangled names the compiled kernels call them by, so what is
 * probed here is exactly what the kernels get.  Built by the test into its tmp_path with the clang that built the shim (one ABI for
 * ext_vector_type) and linked against oracle/_ref/libvpref.so.
 */
#include <stdint.h>

typedef int int2 __attribute__((ext_vector_type(2)));
typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint4 __attribute__((ext_vector_type(4)));

typedef struct {
	int w, h, fmt; /* 0 U8 (CL_R, CL_UNSIGNED_INT8), 1 RGBA8 (CL_RGBA, CL_UNSIGNED_INT8), 2 F32 (CL_R, CL_FLOAT) */
	unsigned char* data;
} ProbeImage;

unsigned long b_get_global_id(unsigned int) __asm__("_Z13get_global_idj");
unsigned long b_get_global_size(unsigned int) __asm__("_Z15get_global_sizej");
void* __translate_sampler_initializer(int);
int b_get_image_width(ProbeImage*) __asm__("_Z15get_image_width14ocl_image2d_ro");
int b_get_image_height(ProbeImage*) __asm__("_Z16get_image_height14ocl_image2d_ro");
uint4 b_read_imageui_i(ProbeImage*, void*, int2) __asm__("_Z12read_imageui14ocl_image2d_ro11ocl_samplerDv2_i");
uint4 b_read_imageui_f(ProbeImage*, void*, float2) __asm__("_Z12read_imageui14ocl_image2d_ro11ocl_samplerDv2_f");
float4 b_read_imagef_i(ProbeImage*, void*, int2) __asm__("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_i");
void b_write_imageui(ProbeImage*, int2, uint4) __asm__("_Z13write_imageui14ocl_image2d_woDv2_iDv4_j");
void b_write_imagef(ProbeImage*, int2, float4) __asm__("_Z12write_imagef14ocl_image2d_woDv2_iDv4_f");
float4 b_convert_float4(uint4) __asm__("_Z14convert_float4Dv4_j");
float4 b_native_sqrt(float4) __asm__("_Z11native_sqrtDv4_f");
float b_min(float, float) __asm__("_Z3minff");
int b_atomic_inc(volatile int*) __asm__("_Z10atomic_incPU8CLglobalVi");
unsigned char b_convert_uchar_sat_f(float) __asm__("_Z17convert_uchar_satf");
unsigned char b_convert_uchar_sat_i(int) __asm__("_Z17convert_uchar_sati");
unsigned char b_convert_uchar_sat_u(unsigned int) __asm__("_Z17convert_uchar_satj");

unsigned long p_get_global_id(unsigned int d) { return b_get_global_id(d); }
unsigned long p_get_global_size(unsigned int d) { return b_get_global_size(d); }
int p_image_width(ProbeImage* im) { return b_get_image_width(im); }
int p_image_height(ProbeImage* im) { return b_get_image_height(im); }
void p_read_imageui_i(ProbeImage* im, int sampler, int x, int y, unsigned int* out) {
	const int2 c = {x, y};
	const uint4 r = b_read_imageui_i(im, __translate_sampler_initializer(sampler), c);
	out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void p_read_imageui_f(ProbeImage* im, int sampler, float u, float v, unsigned int* out) {
	const float2 c = {u, v};
	const uint4 r = b_read_imageui_f(im, __translate_sampler_initializer(sampler), c);
	out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
/* n reads of channel x in one call (the 10^4-coordinate comparison) */
void p_read_imageui_f_many(ProbeImage* im, int sampler, const float* uv, int n, unsigned int* out) {
	void* s = __translate_sampler_initializer(sampler);
	for(int i = 0; i < n; i++) {
		const float2 c = {uv[2 * i], uv[2 * i + 1]};
		out[i] = b_read_imageui_f(im, s, c).x;
	}
}
void p_read_imagef_i(ProbeImage* im, int sampler, int x, int y, float* out) {
	const int2 c = {x, y};
	const float4 r = b_read_imagef_i(im, __translate_sampler_initializer(sampler), c);
	out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void p_write_imageui(ProbeImage* im, int x, int y, const unsigned int* v) {
	const int2 c = {x, y};
	const uint4 q = {v[0], v[1], v[2], v[3]};
	b_write_imageui(im, c, q);
}
void p_write_imagef(ProbeImage* im, int x, int y, const float* v) {
	const int2 c = {x, y};
	const float4 q = {v[0], v[1], v[2], v[3]};
	b_write_imagef(im, c, q);
}
void p_convert_float4(const unsigned int* v, float* out) {
	const uint4 q = {v[0], v[1], v[2], v[3]};
	const float4 r = b_convert_float4(q);
	out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void p_native_sqrt(const float* v, float* out) {
	const float4 q = {v[0], v[1], v[2], v[3]};
	const float4 r = b_native_sqrt(q);
	out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
float p_min(float x, float y) { return b_min(x, y); }
int p_atomic_inc(int* p) { return b_atomic_inc(p); }
int p_convert_uchar_sat_f(float v) { return b_convert_uchar_sat_f(v); }
int p_convert_uchar_sat_i(int v) { return b_convert_uchar_sat_i(v); }
int p_convert_uchar_sat_u(unsigned int v) { return b_convert_uchar_sat_u(v); }
