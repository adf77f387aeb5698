/*
 * oracle/vp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11) of the reference's blob-detection kernel chain.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product (libvp_hip.so and the
 * C++ host facade) never links, imports or executes anything under oracle/.
 *
 * Pinning: every function here is checked bit-for-bit against oracle/_ref (the reference's own kernel .cl files,
 * compiled unmodified, see oracle/Makefile and ref_shim.c) by tests/test_oracle_vs_ref.py in the build
 * container, and against the committed vectors in tests/golden/ everywhere else.  The reference holds no
 * golden vectors or unit tests of its own for this path.  The behaviour of an integer image
 * read through CLK_FILTER_LINEAR (resampling.cl:50) is undefined by OpenCL and pinned by no reference
 * test: for that sampler mode parity is "unpinned" and frozen by the definition in ref_shim.c / lin_u8.
 *
 * All arithmetic is IEEE binary32 with one rounding per source-level operation (build with
 * -ffp-contract=off, no fast-math), in the evaluation order of the cited reference lines.
 */
#ifndef VP_ORACLE_H
#define VP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VPO_RGGB = 0, VPO_GRBG = 1, VPO_BGR = 2 };
enum { VPO_SAMPLER_NEAREST = 0, VPO_SAMPLER_BILINEAR_RTZ = 1 };

/* resampling.cl:20-27 == Perspective.h:22-29 (72 bytes, packed) */
typedef struct __attribute__((packed)) {
	int32_t shape[2];
	float f;
	float p[2];
	float d;
	float r[9];
	float c[3];
} VpoCameraModel;

/* blobList.cl:26-32 == main.cpp:33-41 (22 bytes, packed, unaligned floats) */
typedef struct __attribute__((packed)) {
	float x, y;
	uint8_t color[3];
	uint8_t center[3];
	float circ;
	float score;
} VpoMatch;

/* raw2quad.cl:21-39.  Bayer: qw/qh are 2x2-cell counts; BGR: plane 3 is left untouched. */
void vpo_raw2quad(const uint8_t* raw, int layout, int qw, int qh, uint8_t* p0, uint8_t* p1, uint8_t* p2, uint8_t* p3);

/* resampling.cl:29-100 (field2image :29-47).  flat is RGBA8 fw x fh, tightly packed. */
void vpo_resample(const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3, int qw, int qh, int layout,
		const VpoCameraModel* m, float maxRobotHeight, float fieldScale, float offX, float offY, int sampler_mode,
		uint8_t* flat, int fw, int fh, int threads);

/* gradientDot.cl:22-30 */
void vpo_graddot(const uint8_t* flat, int w, int h, int offset, float* out, int threads);

/* satHorizontal.cl:22-32 / satVertical.cl:22-32: sequential binary32 running sums */
void vpo_sat_horizontal(const float* in, int w, int h, float* out, int threads);
void vpo_sat_vertical(const float* in, int w, int h, float* out, int threads);

/* Exact-integer SAT (design decision H2): int32 with wrap-around.  Returns max |true SAT| (int64) so the
 * caller can assert the 2^24 exactness guard under which the binary32 reference SAT equals this one. */
int64_t vpo_sat_i32(const float* graddot, int w, int h, int32_t* sat);

/* satBlobCenter.cl:34-42 on the binary32 SAT */
void vpo_blobcenter(const float* sat, int w, int h, int radius, float* out, int threads);
/* same taps and evaluation order on the int32 SAT: box sums in wrapping int32, converted once */
void vpo_blobcenter_i32(const int32_t* sat, int w, int h, int radius, float* out, int threads);

/* blobList.cl:36-103; counter[3] is zeroed here (main.cpp:283-288).  Records come out in raster order;
 * idx[i] (optional) is the pixel index y*w+x of record i.  counter[0] may exceed maxMatches. */
void vpo_bloblist(const uint8_t* flat, const float* circ, int w, int h, float circThreshold, float minScore, int radius,
		int maxMatches, VpoMatch* out, uint32_t* idx, int32_t counter[3]);

/* blobScore.cl:23-67 (unlaunched in the reference) */
void vpo_blobscore(const uint8_t* flat, const float* circ, int w, int h, float circThreshold, int radius, float* score, int threads);
/* blobCenter.cl:29-63 (never compiled by the reference) */
void vpo_circularize(const float* graddot, int w, int h, int minRadius, int maxRadius, float* out, int threads);

/* Debug / stream colour kernels (Resources.cpp:145-186).  NV12 buffers are w*h luma bytes followed by
 * ceil(h/2) rows of w interleaved U,V bytes (vpo_nv12_bytes); w must be even.  The reference lets all four pixels of a 2x2
 * block write the block's UV pair (a race): here the last pixel of the block in raster order wins. */
size_t vpo_nv12_bytes(int w, int h);
/* quad2rgba.cl:24-53 */
void vpo_quad2rgba(const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3, int qw, int qh, int layout, int sampler_mode, uint8_t* rgba);
/* quad2nv12.cl:24-58 */
void vpo_quad2nv12(const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3, int qw, int qh, int layout, int sampler_mode, uint8_t* nv12);
/* rgba2nv12.cl:22-31 */
void vpo_rgba2nv12(const uint8_t* rgba, int w, int h, uint8_t* nv12);
/* f2nv12.cl:22-26 */
void vpo_f2nv12(const float* img, int w, int h, uint8_t* nv12);
/* Perspective.cpp:127-129 */
void vpo_flat2field(float x, float y, float fieldScale, float extX, float extY, float* fx, float* fy);

/* Whole chain raw -> blob list for the timed CPU baseline (planes/flat/graddot/sat/blobcenter are
 * caller-provided scratch of the right sizes).  sat_mode: 0 = binary32 sequential (reference structure),
 * 1 = int32.  Returns counter[0]. */
int vpo_frame(const uint8_t* raw, int layout, int qw, int qh, const VpoCameraModel* m, float maxRobotHeight,
		float fieldScale, float offX, float offY, int sampler_mode, int fw, int fh, int o, int r, int R,
		float circThreshold, float minScore, int maxMatches, int sat_mode, int threads,
		uint8_t* planes, uint8_t* flat, float* graddot, float* tmp, float* sat, float* blobcenter,
		VpoMatch* out, uint32_t* idx, int32_t counter[3]);

#ifdef __cplusplus
}
#endif
#endif
