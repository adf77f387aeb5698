/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The reference's kernel/<name>.cl files are compiled UNMODIFIED, from where they lie under
 * /root/reference, to x86-64 objects by oracle/Makefile (ROCm clang, -x cl -ffp-contract=off).
 * Those objects leave 14 OpenCL C builtins undefined (work-item ids, image reads/writes, sampler
 * translation, convert/min/sqrt/atomic_inc).  An OpenCL runtime normally supplies them; none exists in
 * this image, so this file defines them from the OpenCL 3.0 C specification (section 6.15.15 image
 * functions, section 8 image addressing and filtering).  It also holds the NDRange loops that call the compiled
 * kernel bodies (`__clang_ocl_kern_imp_<kernel>`), one call per work-item, raster order.
 *
 * What this file defines and the reference does not:
 *   - nearest filtering:  i = (int)floor(u), address clamp-to-edge = clamp(i, 0, size-1)   (spec 8.2)
 *   - integer image + CLK_FILTER_LINEAR (resampling.cl:50, undefined by the spec):
 *       mode 0 NEAREST       : as nearest
 *       mode 1 BILINEAR_RTZ  : spec 8.2 linear formula evaluated in binary32, one rounding per
 *                              operation, in exactly the order written in lin_u8() below, then
 *                              truncated toward zero.
 *     The mode is a process-global switch (ref_set_sampler_mode) recorded with every golden vector.
 *   - write_imageui to 8-bit channels saturates (spec 6.15.15.4); single-channel images return
 *     (x, 0, 0, 1) on read.
 *   - atomic_inc is an atomic fetch-and-add.  The per-kernel NDRange loops here are sequential, so blob order is raster there;
 *     ref_frame_mt (timing only) splits every NDRange into row bands over host threads, the way a CPU OpenCL runtime spreads
 *     work-groups over cores, and its blob order is whatever the threads produce (blobList.cl:87).
 *
 * Built with the same clang as the kernels so ext_vector_type arguments share one ABI.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int int2 __attribute__((ext_vector_type(2)));
typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint4 __attribute__((ext_vector_type(4)));

enum { REF_FMT_U8 = 0, REF_FMT_RGBA8 = 1, REF_FMT_F32 = 2 };

typedef struct {
	int w, h, fmt;
	unsigned char* data; /* tightly packed rows */
} RefImage;

/* clang's sampler bit encoding (opencl-c-base.h:443-459) */
#define S_ADDR_MASK 0x0E
#define S_ADDR_CLAMP_TO_EDGE 2
#define S_FILTER_LINEAR 0x20

static __thread long g_gid[2];
static __thread long g_gsize[2];
static int g_sampler_mode = 1; /* 0 NEAREST, 1 BILINEAR_RTZ */

void ref_set_sampler_mode(int mode) { g_sampler_mode = mode; }

/* ---- work-item functions ------------------------------------------------------------------------ */
unsigned long ocl_get_global_id(unsigned int d) __asm__("_Z13get_global_idj");
unsigned long ocl_get_global_id(unsigned int d) { return d < 2 ? (unsigned long)g_gid[d] : 0; }
unsigned long ocl_get_global_size(unsigned int d) __asm__("_Z15get_global_sizej");
unsigned long ocl_get_global_size(unsigned int d) { return d < 2 ? (unsigned long)g_gsize[d] : 1; }

void* __translate_sampler_initializer(int v) { return (void*)(intptr_t)v; }

int ocl_get_image_width(RefImage* im) __asm__("_Z15get_image_width14ocl_image2d_ro");
int ocl_get_image_width(RefImage* im) { return im->w; }
int ocl_get_image_height(RefImage* im) __asm__("_Z16get_image_height14ocl_image2d_ro");
int ocl_get_image_height(RefImage* im) { return im->h; }

/* ---- texel access -------------------------------------------------------------------------------- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline void addr(const RefImage* im, void* sampler, int* x, int* y) {
	if(((int)(intptr_t)sampler & S_ADDR_MASK) == S_ADDR_CLAMP_TO_EDGE) {
		*x = clampi(*x, 0, im->w - 1);
		*y = clampi(*y, 0, im->h - 1);
	}
}

static inline uint4 texel_ui(const RefImage* im, int x, int y) {
	uint4 r = {0, 0, 0, 1};
	if(im->fmt == REF_FMT_U8) {
		r.x = im->data[(size_t)y * im->w + x];
	} else if(im->fmt == REF_FMT_RGBA8) {
		const unsigned char* p = im->data + 4 * ((size_t)y * im->w + x);
		r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
	}
	return r;
}

uint4 ocl_read_imageui_i(RefImage* im, void* s, int2 c) __asm__("_Z12read_imageui14ocl_image2d_ro11ocl_samplerDv2_i");
uint4 ocl_read_imageui_i(RefImage* im, void* s, int2 c) {
	int x = c.x, y = c.y;
	addr(im, s, &x, &y);
	return texel_ui(im, x, y);
}

float4 ocl_read_imagef_i(RefImage* im, void* s, int2 c) __asm__("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_i");
float4 ocl_read_imagef_i(RefImage* im, void* s, int2 c) {
	int x = c.x, y = c.y;
	addr(im, s, &x, &y);
	float4 r = {0.f, 0.f, 0.f, 1.f};
	r.x = ((const float*)im->data)[(size_t)y * im->w + x];
	return r;
}

/* spec 8.2 linear filter on one 8-bit channel, binary32, fixed evaluation order, truncation */
static inline unsigned int lin_u8(const RefImage* im, float u, float v) {
	const float um = u - 0.5f, vm = v - 0.5f;
	const float fu = floorf(um), fv = floorf(vm);
	const float a = um - fu, b = vm - fv;
	const int i0 = clampi((int)fu, 0, im->w - 1), i1 = clampi((int)fu + 1, 0, im->w - 1);
	const int j0 = clampi((int)fv, 0, im->h - 1), j1 = clampi((int)fv + 1, 0, im->h - 1);
	const float t00 = (float)im->data[(size_t)j0 * im->w + i0];
	const float t10 = (float)im->data[(size_t)j0 * im->w + i1];
	const float t01 = (float)im->data[(size_t)j1 * im->w + i0];
	const float t11 = (float)im->data[(size_t)j1 * im->w + i1];
	const float oma = 1.0f - a, omb = 1.0f - b;
	const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
	float t = w00 * t00;
	t = t + w10 * t10;
	t = t + w01 * t01;
	t = t + w11 * t11;
	return (unsigned int)t;
}

uint4 ocl_read_imageui_f(RefImage* im, void* s, float2 c) __asm__("_Z12read_imageui14ocl_image2d_ro11ocl_samplerDv2_f");
uint4 ocl_read_imageui_f(RefImage* im, void* s, float2 c) {
	/* only ever called on U8 planes with the LINEAR|CLAMP_TO_EDGE sampler (resampling.cl:50) */
	if(((int)(intptr_t)s & S_FILTER_LINEAR) && g_sampler_mode == 1) {
		uint4 r = {lin_u8(im, c.x, c.y), 0, 0, 1};
		return r;
	}
	int x = (int)floorf(c.x), y = (int)floorf(c.y);
	x = clampi(x, 0, im->w - 1);
	y = clampi(y, 0, im->h - 1);
	return texel_ui(im, x, y);
}

static inline unsigned char sat8(unsigned int v) { return v > 255u ? 255 : (unsigned char)v; }

void ocl_write_imageui(RefImage* im, int2 c, uint4 v) __asm__("_Z13write_imageui14ocl_image2d_woDv2_iDv4_j");
void ocl_write_imageui(RefImage* im, int2 c, uint4 v) {
	if(c.x < 0 || c.y < 0 || c.x >= im->w || c.y >= im->h) return;
	if(im->fmt == REF_FMT_U8) {
		im->data[(size_t)c.y * im->w + c.x] = sat8(v.x);
	} else {
		unsigned char* p = im->data + 4 * ((size_t)c.y * im->w + c.x);
		p[0] = sat8(v.x); p[1] = sat8(v.y); p[2] = sat8(v.z); p[3] = sat8(v.w);
	}
}

void ocl_write_imagef(RefImage* im, int2 c, float4 v) __asm__("_Z12write_imagef14ocl_image2d_woDv2_iDv4_f");
void ocl_write_imagef(RefImage* im, int2 c, float4 v) {
	if(c.x < 0 || c.y < 0 || c.x >= im->w || c.y >= im->h) return;
	((float*)im->data)[(size_t)c.y * im->w + c.x] = v.x;
}

/* ---- math / conversion / atomics ---------------------------------------------------------------- */
float4 ocl_convert_float4(uint4 v) __asm__("_Z14convert_float4Dv4_j");
float4 ocl_convert_float4(uint4 v) {
	float4 r = {(float)v.x, (float)v.y, (float)v.z, (float)v.w};
	return r;
}
float4 ocl_native_sqrt(float4 v) __asm__("_Z11native_sqrtDv4_f");
float4 ocl_native_sqrt(float4 v) {
	float4 r = {sqrtf(v.x), sqrtf(v.y), sqrtf(v.z), sqrtf(v.w)};
	return r;
}
float ocl_min(float x, float y) __asm__("_Z3minff");
float ocl_min(float x, float y) { return y < x ? y : x; }
int ocl_atomic_inc(volatile int* p) __asm__("_Z10atomic_incPU8CLglobalVi");
int ocl_atomic_inc(volatile int* p) { return __atomic_fetch_add((int*)p, 1, __ATOMIC_RELAXED); }
/* convert_uchar_sat (OpenCL C 6.4.3): saturating; the float overload rounds toward zero, NaN -> 0 */
unsigned char ocl_convert_uchar_sat_u(unsigned int v) __asm__("_Z17convert_uchar_satj");
unsigned char ocl_convert_uchar_sat_u(unsigned int v) { return v > 255u ? 255 : (unsigned char)v; }
unsigned char ocl_convert_uchar_sat_i(int v) __asm__("_Z17convert_uchar_sati");
unsigned char ocl_convert_uchar_sat_i(int v) { return v < 0 ? 0 : (v > 255 ? 255 : (unsigned char)v); }
unsigned char ocl_convert_uchar_sat_f(float v) __asm__("_Z17convert_uchar_satf");
unsigned char ocl_convert_uchar_sat_f(float v) { return !(v > 0.0f) ? 0 : (v >= 255.0f ? 255 : (unsigned char)v); }

/* ---- compiled kernel bodies (symbols renamed per layout / file by the Makefile) ------------------ */
typedef struct __attribute__((packed)) {
	int shape[2];
	float f;
	float p[2];
	float d;
	float r[9];
	float c[3];
} RefCameraModel; /* resampling.cl:20-27 */

#define DECL_R2Q(L) void __clang_ocl_kern_imp_raw2quad_##L(const unsigned char*, RefImage*, RefImage*, RefImage*, RefImage*)
DECL_R2Q(RGGB); DECL_R2Q(GRBG); DECL_R2Q(BGR);
#define DECL_RES(L) void __clang_ocl_kern_imp_resampling_##L(RefImage*, RefImage*, RefImage*, RefImage*, RefImage*, RefCameraModel, float, float, float, float)
DECL_RES(RGGB); DECL_RES(GRBG); DECL_RES(BGR);
void __clang_ocl_kern_imp_gradient_dotproduct(RefImage*, RefImage*, int);
void __clang_ocl_kern_imp_sat_horizontal(RefImage*, RefImage*);
void __clang_ocl_kern_imp_sat_vertical(RefImage*, RefImage*);
void __clang_ocl_kern_imp_circle(RefImage*, RefImage*, int);
void __clang_ocl_kern_imp_matches(RefImage*, RefImage*, void*, volatile int*, float, float, int, int);
void __clang_ocl_kern_imp_matches_score(RefImage*, RefImage*, RefImage*, float, int);
void __clang_ocl_kern_imp_circularize(RefImage*, RefImage*, int, int);
#define DECL_Q2R(L) void __clang_ocl_kern_imp_quad2rgba_##L(RefImage*, RefImage*, RefImage*, RefImage*, RefImage*)
DECL_Q2R(RGGB); DECL_Q2R(GRBG); DECL_Q2R(BGR);
#define DECL_Q2N(L) void __clang_ocl_kern_imp_quad2nv12_##L(RefImage*, RefImage*, RefImage*, RefImage*, unsigned char*)
DECL_Q2N(RGGB); DECL_Q2N(GRBG); DECL_Q2N(BGR);
void __clang_ocl_kern_imp_rgb2nv12(RefImage*, unsigned char*);
void __clang_ocl_kern_imp_f2nv12(RefImage*, unsigned char*);

#define NDRANGE_2D(W, H, CALL) \
	do { g_gsize[0] = (W); g_gsize[1] = (H); \
		for(long y_ = 0; y_ < (H); y_++) for(long x_ = 0; x_ < (W); x_++) { g_gid[0] = x_; g_gid[1] = y_; CALL; } } while(0)
#define NDRANGE_1D(N, CALL) \
	do { g_gsize[0] = (N); g_gsize[1] = 1; g_gid[1] = 0; \
		for(long x_ = 0; x_ < (N); x_++) { g_gid[0] = x_; CALL; } } while(0)

static RefImage mk(int w, int h, int fmt, void* d) { RefImage i = {w, h, fmt, (unsigned char*)d}; return i; }

/* layout: 0 RGGB, 1 GRBG, 2 BGR.  Launch shapes follow Resources.cpp:142,159-163 and main.cpp:289. */
void ref_raw2quad(const unsigned char* raw, int layout, int qw, int qh, unsigned char* p0, unsigned char* p1, unsigned char* p2, unsigned char* p3) {
	RefImage c0 = mk(qw, qh, REF_FMT_U8, p0), c1 = mk(qw, qh, REF_FMT_U8, p1), c2 = mk(qw, qh, REF_FMT_U8, p2), c3 = mk(qw, qh, REF_FMT_U8, p3);
	if(layout == 0) NDRANGE_2D(qw, qh, __clang_ocl_kern_imp_raw2quad_RGGB(raw, &c0, &c1, &c2, &c3));
	else if(layout == 1) NDRANGE_2D(qw, qh, __clang_ocl_kern_imp_raw2quad_GRBG(raw, &c0, &c1, &c2, &c3));
	else NDRANGE_2D(qw, qh, __clang_ocl_kern_imp_raw2quad_BGR(raw, &c0, &c1, &c2, &c3));
}

void ref_resampling(unsigned char* p0, unsigned char* p1, unsigned char* p2, unsigned char* p3, int qw, int qh, int layout,
		const RefCameraModel* m, float maxRobotHeight, float fieldScale, float offX, float offY, unsigned char* flat, int fw, int fh) {
	RefImage c0 = mk(qw, qh, REF_FMT_U8, p0), c1 = mk(qw, qh, REF_FMT_U8, p1), c2 = mk(qw, qh, REF_FMT_U8, p2), c3 = mk(qw, qh, REF_FMT_U8, p3);
	RefImage out = mk(fw, fh, REF_FMT_RGBA8, flat);
	if(layout == 0) NDRANGE_2D(fw, fh, __clang_ocl_kern_imp_resampling_RGGB(&c0, &c1, &c2, &c3, &out, *m, maxRobotHeight, fieldScale, offX, offY));
	else if(layout == 1) NDRANGE_2D(fw, fh, __clang_ocl_kern_imp_resampling_GRBG(&c0, &c1, &c2, &c3, &out, *m, maxRobotHeight, fieldScale, offX, offY));
	else NDRANGE_2D(fw, fh, __clang_ocl_kern_imp_resampling_BGR(&c0, &c1, &c2, &c3, &out, *m, maxRobotHeight, fieldScale, offX, offY));
}

void ref_gradient_dotproduct(unsigned char* flat, int w, int h, int offset, float* out) {
	RefImage in = mk(w, h, REF_FMT_RGBA8, flat), o = mk(w, h, REF_FMT_F32, out);
	NDRANGE_2D(w, h, __clang_ocl_kern_imp_gradient_dotproduct(&in, &o, offset));
}

void ref_sat_horizontal(float* in, int w, int h, float* out) {
	RefImage i = mk(w, h, REF_FMT_F32, in), o = mk(w, h, REF_FMT_F32, out);
	NDRANGE_1D(h, __clang_ocl_kern_imp_sat_horizontal(&i, &o));
}

void ref_sat_vertical(float* in, int w, int h, float* out) {
	RefImage i = mk(w, h, REF_FMT_F32, in), o = mk(w, h, REF_FMT_F32, out);
	NDRANGE_1D(w, __clang_ocl_kern_imp_sat_vertical(&i, &o));
}

void ref_circle(float* sat, int w, int h, int radius, float* out) {
	RefImage i = mk(w, h, REF_FMT_F32, sat), o = mk(w, h, REF_FMT_F32, out);
	NDRANGE_2D(w, h, __clang_ocl_kern_imp_circle(&i, &o, radius));
}

/* matches: 22-byte packed records (blobList.cl:26-32); counter[3] must be zeroed by the caller (main.cpp:283-288).
 * idx_out (optional, capacity maxMatches) receives the raster pixel index of each appended record. */
void ref_matches(unsigned char* flat, float* circ, int w, int h, void* matches, int* counter, float circThreshold, float minScore, int radius, int maxMatches, unsigned int* idx_out) {
	RefImage img = mk(w, h, REF_FMT_RGBA8, flat), c = mk(w, h, REF_FMT_F32, circ);
	g_gsize[0] = w; g_gsize[1] = h;
	for(long y = 0; y < h; y++) for(long x = 0; x < w; x++) {
		g_gid[0] = x; g_gid[1] = y;
		const int before = counter[0];
		__clang_ocl_kern_imp_matches(&img, &c, matches, counter, circThreshold, minScore, radius, maxMatches);
		if(idx_out && counter[0] != before && before < maxMatches) idx_out[before] = (unsigned int)(y * w + x);
	}
}

void ref_blobscore(unsigned char* flat, float* circ, int w, int h, float* score, float circThreshold, int radius) {
	RefImage img = mk(w, h, REF_FMT_RGBA8, flat), c = mk(w, h, REF_FMT_F32, circ), s = mk(w, h, REF_FMT_F32, score);
	NDRANGE_2D(w, h, __clang_ocl_kern_imp_matches_score(&img, &c, &s, circThreshold, radius));
}

void ref_circularize(float* graddot, int w, int h, int minRadius, int maxRadius, float* out) {
	RefImage i = mk(w, h, REF_FMT_F32, graddot), o = mk(w, h, REF_FMT_F32, out);
	NDRANGE_2D(w, h, __clang_ocl_kern_imp_circularize(&i, &o, minRadius, maxRadius));
}

/* ---- debug / stream colour kernels (Resources.cpp:145-186).  The NV12 kernels let every pixel of a 2x2 block write the block's
 * one UV pair (quad2nv12.cl:55-57, rgba2nv12.cl:28-30): a data race on a real device; this sequential raster execution
 * resolves it as "the last pixel of the block in raster order wins". */
void ref_quad2rgba(unsigned char* p0, unsigned char* p1, unsigned char* p2, unsigned char* p3, int qw, int qh, int layout, unsigned char* rgba) {
	RefImage c0 = mk(qw, qh, REF_FMT_U8, p0), c1 = mk(qw, qh, REF_FMT_U8, p1), c2 = mk(qw, qh, REF_FMT_U8, p2), c3 = mk(qw, qh, REF_FMT_U8, p3);
	RefImage out = mk(qw, qh, REF_FMT_RGBA8, rgba);
	if(layout == 0) NDRANGE_2D(qw, qh, __clang_ocl_kern_imp_quad2rgba_RGGB(&c0, &c1, &c2, &c3, &out));
	else if(layout == 1) NDRANGE_2D(qw, qh, __clang_ocl_kern_imp_quad2rgba_GRBG(&c0, &c1, &c2, &c3, &out));
	else NDRANGE_2D(qw, qh, __clang_ocl_kern_imp_quad2rgba_BGR(&c0, &c1, &c2, &c3, &out));
}

void ref_quad2nv12(unsigned char* p0, unsigned char* p1, unsigned char* p2, unsigned char* p3, int qw, int qh, int layout, unsigned char* nv12) {
	RefImage c0 = mk(qw, qh, REF_FMT_U8, p0), c1 = mk(qw, qh, REF_FMT_U8, p1), c2 = mk(qw, qh, REF_FMT_U8, p2), c3 = mk(qw, qh, REF_FMT_U8, p3);
	if(layout == 0) NDRANGE_2D(qw, qh, __clang_ocl_kern_imp_quad2nv12_RGGB(&c0, &c1, &c2, &c3, nv12));
	else if(layout == 1) NDRANGE_2D(qw, qh, __clang_ocl_kern_imp_quad2nv12_GRBG(&c0, &c1, &c2, &c3, nv12));
	else NDRANGE_2D(qw, qh, __clang_ocl_kern_imp_quad2nv12_BGR(&c0, &c1, &c2, &c3, nv12));
}

void ref_rgba2nv12(unsigned char* rgba, int w, int h, unsigned char* nv12) {
	RefImage in = mk(w, h, REF_FMT_RGBA8, rgba);
	NDRANGE_2D(w, h, __clang_ocl_kern_imp_rgb2nv12(&in, nv12));
}

void ref_f2nv12(float* img, int w, int h, unsigned char* nv12) {
	RefImage in = mk(w, h, REF_FMT_F32, img);
	NDRANGE_2D(w, h, __clang_ocl_kern_imp_f2nv12(&in, nv12));
}


/* ---- whole frame on several host threads (bench.py cpu_baseline kind "reference") ---------------------------------------------
 * The seven launches of Resources.cpp:142,159-163 and main.cpp:289, each NDRange cut into contiguous bands (rows of a 2-D range,
 * work-items of a 1-D range), one band per thread, a join between launches (the reference chains them by events).  Every
 * work-item still runs the reference's own compiled kernel body. */
typedef struct {
	int stage, layout;
	long lo, hi, w, h;
	const unsigned char* raw;
	RefImage *c0, *c1, *c2, *c3, *flat, *f0, *f1;
	const RefCameraModel* m;
	float maxRobotHeight, fieldScale, offX, offY, circThreshold, minScore;
	int offset, radius, listRadius, maxMatches;
	void* matches;
	int* counter;
} RefJob;

static void* ref_job_run(void* arg) {
	const RefJob* j = (const RefJob*)arg;
	if(j->stage == 3 || j->stage == 4) { /* 1-D ranges: satHorizontal over rows, satVertical over columns */
		g_gsize[0] = j->stage == 3 ? j->h : j->w; g_gsize[1] = 1; g_gid[1] = 0;
		for(long i = j->lo; i < j->hi; i++) {
			g_gid[0] = i;
			if(j->stage == 3) __clang_ocl_kern_imp_sat_horizontal(j->f0, j->f1);
			else __clang_ocl_kern_imp_sat_vertical(j->f0, j->f1);
		}
		return NULL;
	}
	g_gsize[0] = j->w; g_gsize[1] = j->h;
	for(long y = j->lo; y < j->hi; y++) for(long x = 0; x < j->w; x++) {
		g_gid[0] = x; g_gid[1] = y;
		switch(j->stage) {
		case 0:
			if(j->layout == 0) __clang_ocl_kern_imp_raw2quad_RGGB(j->raw, j->c0, j->c1, j->c2, j->c3);
			else if(j->layout == 1) __clang_ocl_kern_imp_raw2quad_GRBG(j->raw, j->c0, j->c1, j->c2, j->c3);
			else __clang_ocl_kern_imp_raw2quad_BGR(j->raw, j->c0, j->c1, j->c2, j->c3);
			break;
		case 1:
			if(j->layout == 0) __clang_ocl_kern_imp_resampling_RGGB(j->c0, j->c1, j->c2, j->c3, j->flat, *j->m, j->maxRobotHeight, j->fieldScale, j->offX, j->offY);
			else if(j->layout == 1) __clang_ocl_kern_imp_resampling_GRBG(j->c0, j->c1, j->c2, j->c3, j->flat, *j->m, j->maxRobotHeight, j->fieldScale, j->offX, j->offY);
			else __clang_ocl_kern_imp_resampling_BGR(j->c0, j->c1, j->c2, j->c3, j->flat, *j->m, j->maxRobotHeight, j->fieldScale, j->offX, j->offY);
			break;
		case 2: __clang_ocl_kern_imp_gradient_dotproduct(j->flat, j->f0, j->offset); break;
		case 5: __clang_ocl_kern_imp_circle(j->f0, j->f1, j->radius); break;
		case 6: __clang_ocl_kern_imp_matches(j->flat, j->f0, j->matches, j->counter, j->circThreshold, j->minScore, j->listRadius, j->maxMatches); break;
		}
	}
	return NULL;
}

static void ref_launch(RefJob base, long n, int threads) {
	if(threads < 1) threads = 1;
	if(threads > 256) threads = 256;
	pthread_t tid[256];
	RefJob jobs[256];
	int started = 0;
	for(int t = 0; t < threads; t++) {
		jobs[t] = base;
		jobs[t].lo = n * t / threads;
		jobs[t].hi = n * (t + 1) / threads;
		if(jobs[t].lo == jobs[t].hi) continue;
		if(t == threads - 1 || pthread_create(&tid[started], NULL, ref_job_run, &jobs[t]) != 0) ref_job_run(&jobs[t]);
		else started++;
	}
	for(int t = 0; t < started; t++) pthread_join(tid[t], NULL);
}

/* scratch: caller-provided, 4*qw*qh bytes of planes + 4*fw*fh bytes flat + 3 * 4*fw*fh bytes of float images.  counter[3] is zeroed
 * here (main.cpp:283-288).  Returns 0, or -1 when a buffer is missing. */
int ref_frame_mt(const unsigned char* raw, int layout, int qw, int qh, const RefCameraModel* m, float maxRobotHeight, float fieldScale, float offX, float offY,
		int fw, int fh, int offset, int circleRadius, int listRadius, float circThreshold, float minScore, int maxMatches, int threads,
		unsigned char* planes, unsigned char* flat, float* fa, float* fb, float* fc, void* matches, int* counter) {
	if(!raw || !m || !planes || !flat || !fa || !fb || !fc || !matches || !counter) return -1;
	const size_t pn = (size_t)qw * qh;
	RefImage c0 = mk(qw, qh, REF_FMT_U8, planes), c1 = mk(qw, qh, REF_FMT_U8, planes + pn), c2 = mk(qw, qh, REF_FMT_U8, planes + 2 * pn), c3 = mk(qw, qh, REF_FMT_U8, planes + 3 * pn);
	RefImage fl = mk(fw, fh, REF_FMT_RGBA8, flat), ia = mk(fw, fh, REF_FMT_F32, fa), ib = mk(fw, fh, REF_FMT_F32, fb), ic = mk(fw, fh, REF_FMT_F32, fc);
	RefJob j;
	memset(&j, 0, sizeof(j));
	j.layout = layout; j.raw = raw; j.c0 = &c0; j.c1 = &c1; j.c2 = &c2; j.c3 = &c3; j.flat = &fl; j.m = m;
	j.maxRobotHeight = maxRobotHeight; j.fieldScale = fieldScale; j.offX = offX; j.offY = offY;
	j.offset = offset; j.radius = circleRadius; j.listRadius = listRadius; j.circThreshold = circThreshold; j.minScore = minScore; j.maxMatches = maxMatches;
	j.matches = matches; j.counter = counter;
	counter[0] = counter[1] = counter[2] = 0;
	j.stage = 0; j.w = qw; j.h = qh; ref_launch(j, qh, threads);                       /* raw2quad        (Resources.cpp:142) */
	j.stage = 1; j.w = fw; j.h = fh; ref_launch(j, fh, threads);                       /* resampling      (:159) */
	j.stage = 2; j.f0 = &ia; ref_launch(j, fh, threads);                               /* gradientDot     (:160) -> fa */
	j.stage = 3; j.f0 = &ia; j.f1 = &ib; ref_launch(j, fh, threads);                   /* satHorizontal   (:161) fa -> fb, one work-item per row */
	j.stage = 4; j.f0 = &ib; j.f1 = &ic; ref_launch(j, fw, threads);                   /* satVertical     (:162) fb -> fc, one work-item per column */
	j.stage = 5; j.f0 = &ic; j.f1 = &ia; ref_launch(j, fh, threads);                   /* circle          (:163) fc -> fa */
	j.stage = 6; j.f0 = &ia; ref_launch(j, fh, threads);                               /* matches         (main.cpp:289) */
	return 0;
}
