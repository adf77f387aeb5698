/*
 * oracle/vp_oracle.c -- TEST INFRASTRUCTURE ONLY.  See vp_oracle.h for the contract and the pinning.
 *
 * Each function restates one reference kernel from its source text; citations are file:line under
 * /root/reference.  Rows (or columns) may be sharded over pthreads: no stage has a cross-row
 * floating-point reduction except the two SAT passes, which are sharded along the independent axis, so the
 * threaded result is bit-identical to the 1-thread result.
 */
#define _GNU_SOURCE
#include "vp_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- row-sharded helper ---------------- */
typedef void (*span_fn)(void* ctx, int begin, int end);
typedef struct { span_fn fn; void* ctx; int begin, end; } span_job;

static void* span_tramp(void* p) {
	span_job* j = (span_job*)p;
	j->fn(j->ctx, j->begin, j->end);
	return NULL;
}

static void parallel_span(span_fn fn, void* ctx, int n, int threads) {
	if(threads <= 1 || n < 2 * threads) {
		fn(ctx, 0, n);
		return;
	}
	if(threads > 256) threads = 256;
	pthread_t tid[256];
	span_job job[256];
	const int chunk = (n + threads - 1) / threads;
	int used = 0;
	for(int t = 0; t < threads; t++) {
		int b = t * chunk, e = b + chunk > n ? n : b + chunk;
		if(b >= e) break;
		job[t] = (span_job){fn, ctx, b, e};
		pthread_create(&tid[t], NULL, span_tramp, &job[t]);
		used++;
	}
	for(int t = 0; t < used; t++) pthread_join(tid[t], NULL);
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---------------------------------------------------------------- raw2quad -------------------------- */
void vpo_raw2quad(const uint8_t* raw, int layout, int qw, int qh, uint8_t* p0, uint8_t* p1, uint8_t* p2, uint8_t* p3) {
	if(layout == VPO_BGR) { /* raw2quad.cl:24-29 */
		for(int y = 0; y < qh; y++)
			for(int x = 0; x < qw; x++) {
				const size_t ip = 3 * ((size_t)x + (size_t)y * qw);
				const size_t o = (size_t)y * qw + x;
				p0[o] = raw[ip];
				p1[o] = raw[ip + 1];
				p2[o] = raw[ip + 2];
			}
		(void)p3;
		return;
	}
	/* raw2quad.cl:30-38: 2x2 cell -> ch0 TL, ch1 TR, ch2 BL, ch3 BR */
	const size_t row = 2 * (size_t)qw;
	for(int y = 0; y < qh; y++)
		for(int x = 0; x < qw; x++) {
			const size_t ip = 2 * (size_t)x + 2 * (size_t)y * row;
			const size_t o = (size_t)y * qw + x;
			p0[o] = raw[ip];
			p1[o] = raw[ip + 1];
			p2[o] = raw[ip + row];
			p3[o] = raw[ip + 1 + row];
		}
}

/* ---------------------------------------------------------------- resampling ------------------------ */
/* resampling.cl:29-47 */
static inline void field2image(const VpoCameraModel* m, float fx, float fy, float fz, float* ox, float* oy) {
	fx = fx - m->c[0];
	fy = fy - m->c[1];
	fz = fz - m->c[2];
	const float cx = m->r[0] * fx + m->r[1] * fy + m->r[2] * fz;
	const float cy = m->r[3] * fx + m->r[4] * fy + m->r[5] * fz;
	const float cz = m->r[6] * fx + m->r[7] * fy + m->r[8] * fz;
	const float x2 = cx / cz, y2 = cy / cz;
	float ux = x2, uy = y2;
	for(int i = 0; i < 8; i++) {
		const float rx = ux * ux, ry = uy * uy;
		const float dr = 1.0f + m->d * (rx + ry);
		ux = x2 / dr;
		uy = y2 / dr;
	}
	*ox = m->f * ux + m->p[0];
	*oy = m->f * uy + m->p[1];
}

/* read_imageui(plane, LINEAR|CLAMP_TO_EDGE, float2).x on a U8 plane; see vp_oracle.h on sampler_mode */
static inline uint32_t sample_u8(const uint8_t* p, int w, int h, float u, float v, int mode) {
	if(mode == VPO_SAMPLER_NEAREST) {
		const int x = clampi((int)floorf(u), 0, w - 1), y = clampi((int)floorf(v), 0, h - 1);
		return p[(size_t)y * w + x];
	}
	const float um = u - 0.5f, vm = v - 0.5f;
	const float fu = floorf(um), fv = floorf(vm);
	const float a = um - fu, b = vm - fv;
	const int i0 = clampi((int)fu, 0, w - 1), i1 = clampi((int)fu + 1, 0, w - 1);
	const int j0 = clampi((int)fv, 0, h - 1), j1 = clampi((int)fv + 1, 0, h - 1);
	const float t00 = (float)p[(size_t)j0 * w + i0], t10 = (float)p[(size_t)j0 * w + i1];
	const float t01 = (float)p[(size_t)j1 * w + i0], t11 = (float)p[(size_t)j1 * w + i1];
	const float oma = 1.0f - a, omb = 1.0f - b;
	const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
	float t = w00 * t00;
	t = t + w10 * t10;
	t = t + w01 * t01;
	t = t + w11 * t11;
	return (uint32_t)t;
}

typedef struct {
	const uint8_t* p[4];
	int qw, qh, layout;
	const VpoCameraModel* m;
	float height, fs, ox, oy;
	int mode;
	uint8_t* flat;
	int fw, fh;
} resample_ctx;

static void resample_rows(void* vc, int y0, int y1) {
	const resample_ctx* c = (const resample_ctx*)vc;
	for(int gy = y0; gy < y1; gy++)
		for(int gx = 0; gx < c->fw; gx++) {
			float px, py; /* resampling.cl:53 */
			field2image(c->m, (float)gx * c->fs + c->ox, (float)gy * c->fs + c->oy, c->height, &px, &py);
			uint32_t r, g, b;
			if(c->layout == VPO_BGR) { /* :56-61 */
				r = sample_u8(c->p[2], c->qw, c->qh, px, py, c->mode);
				g = sample_u8(c->p[1], c->qw, c->qh, px, py, c->mode);
				b = sample_u8(c->p[0], c->qw, c->qh, px, py, c->mode);
			} else if(c->layout == VPO_RGGB) { /* :64-71 */
				r = sample_u8(c->p[0], c->qw, c->qh, px + 0.25f, py + 0.25f, c->mode);
				g = sample_u8(c->p[1], c->qw, c->qh, px - 0.25f, py + 0.25f, c->mode) / 2 + sample_u8(c->p[2], c->qw, c->qh, px + 0.25f, py - 0.25f, c->mode) / 2;
				b = sample_u8(c->p[3], c->qw, c->qh, px - 0.25f, py - 0.25f, c->mode);
			} else { /* GRBG :74-79 */
				r = sample_u8(c->p[1], c->qw, c->qh, px - 0.25f, py + 0.25f, c->mode);
				g = sample_u8(c->p[0], c->qw, c->qh, px + 0.25f, py + 0.25f, c->mode) / 2 + sample_u8(c->p[3], c->qw, c->qh, px - 0.25f, py - 0.25f, c->mode) / 2;
				b = sample_u8(c->p[2], c->qw, c->qh, px + 0.25f, py - 0.25f, c->mode);
			}
			/* dRGB :86-91, uint arithmetic, then write_imageui saturation (never triggers: max 255) */
			uint8_t* o = c->flat + 4 * ((size_t)gy * c->fw + gx);
			const uint32_t dr = (2 * r - g - b + 510) / 4, dg = (2 * g - b - r + 510) / 4, db = (2 * b - r - g + 510) / 4;
			o[0] = dr > 255 ? 255 : (uint8_t)dr;
			o[1] = dg > 255 ? 255 : (uint8_t)dg;
			o[2] = db > 255 ? 255 : (uint8_t)db;
			o[3] = 255;
		}
}

void vpo_resample(const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3, int qw, int qh, int layout,
		const VpoCameraModel* m, float maxRobotHeight, float fieldScale, float offX, float offY, int sampler_mode,
		uint8_t* flat, int fw, int fh, int threads) {
	resample_ctx c = {{p0, p1, p2, p3}, qw, qh, layout, m, maxRobotHeight, fieldScale, offX, offY, sampler_mode, flat, fw, fh};
	parallel_span(resample_rows, &c, fh, threads);
}

/* ---------------------------------------------------------------- gradientDot ----------------------- */
typedef struct { const uint8_t* flat; int w, h, o; float* out; } graddot_ctx;

static void graddot_rows(void* vc, int y0, int y1) {
	const graddot_ctx* c = (const graddot_ctx*)vc;
	const int w = c->w, h = c->h, o = c->o;
	for(int y = y0; y < y1; y++) {
		const uint8_t* rowc = c->flat + 4 * (size_t)y * w;
		const uint8_t* rowp = c->flat + 4 * (size_t)clampi(y + o, 0, h - 1) * w;
		const uint8_t* rown = c->flat + 4 * (size_t)clampi(y - o, 0, h - 1) * w;
		for(int x = 0; x < w; x++) { /* gradientDot.cl:25-29, clamp-to-edge :20 */
			const uint8_t* xp = rowc + 4 * clampi(x + o, 0, w - 1);
			const uint8_t* xn = rowc + 4 * clampi(x - o, 0, w - 1);
			const uint8_t* yp = rowp + 4 * x;
			const uint8_t* yn = rown + 4 * x;
			float acc[3];
			for(int k = 0; k < 3; k++) {
				const float gx = (float)xp[k] - (float)xn[k];
				const float gy = (float)yp[k] - (float)yn[k];
				acc[k] = gx * gy;
			}
			c->out[(size_t)y * w + x] = acc[0] + acc[1] + acc[2];
		}
	}
}

void vpo_graddot(const uint8_t* flat, int w, int h, int offset, float* out, int threads) {
	graddot_ctx c = {flat, w, h, offset, out};
	parallel_span(graddot_rows, &c, h, threads);
}

/* ---------------------------------------------------------------- SAT ------------------------------- */
typedef struct { const float* in; int w, h; float* out; } sat_ctx;

static void sat_h_rows(void* vc, int y0, int y1) { /* satHorizontal.cl:26-31 */
	const sat_ctx* c = (const sat_ctx*)vc;
	for(int y = y0; y < y1; y++) {
		float sum = 0.f;
		const float* i = c->in + (size_t)y * c->w;
		float* o = c->out + (size_t)y * c->w;
		for(int x = 0; x < c->w; x++) {
			sum += i[x];
			o[x] = sum;
		}
	}
}

static void sat_v_cols(void* vc, int x0, int x1) { /* satVertical.cl:26-31; each column keeps its own running sum */
	const sat_ctx* c = (const sat_ctx*)vc;
	if(x1 <= x0) return;
	float* sum = (float*)calloc((size_t)(x1 - x0), sizeof(float));
	for(int y = 0; y < c->h; y++) {
		const float* i = c->in + (size_t)y * c->w;
		float* o = c->out + (size_t)y * c->w;
		for(int x = x0; x < x1; x++) {
			sum[x - x0] += i[x];
			o[x] = sum[x - x0];
		}
	}
	free(sum);
}

void vpo_sat_horizontal(const float* in, int w, int h, float* out, int threads) {
	sat_ctx c = {in, w, h, out};
	parallel_span(sat_h_rows, &c, h, threads);
}

void vpo_sat_vertical(const float* in, int w, int h, float* out, int threads) {
	sat_ctx c = {in, w, h, out};
	parallel_span(sat_v_cols, &c, w, threads);
}

int64_t vpo_sat_i32(const float* graddot, int w, int h, int32_t* sat) {
	int64_t* col = (int64_t*)calloc((size_t)w, sizeof(int64_t));
	int64_t maxabs = 0;
	for(int y = 0; y < h; y++) {
		int64_t rowsum = 0;
		for(int x = 0; x < w; x++) {
			rowsum += (int64_t)graddot[(size_t)y * w + x]; /* exact integers, |v| <= 3*255^2 */
			col[x] += rowsum;
			const int64_t v = col[x];
			const int64_t a = v < 0 ? -v : v;
			if(a > maxabs) maxabs = a;
			sat[(size_t)y * w + x] = (int32_t)(uint32_t)(uint64_t)v;
		}
	}
	free(col);
	return maxabs;
}

/* ---------------------------------------------------------------- satBlobCenter --------------------- */
typedef struct { const float* sat; const int32_t* sati; int w, h, r; float* out; } circle_ctx;

static inline float satf(const circle_ctx* c, int x, int y, int dx, int dy) { /* satBlobCenter.cl:22-27 */
	return c->sat[(size_t)clampi(y + dy, 0, c->h - 1) * c->w + clampi(x + dx, 0, c->w - 1)];
}
static inline float fminf_cl(float x, float y) { return y < x ? y : x; } /* OpenCL min(): y < x ? y : x */

static void circle_rows(void* vc, int y0, int y1) {
	const circle_ctx* c = (const circle_ctx*)vc;
	const int r = c->r;
	const float div = (float)(r * r); /* :41 int product converted to float */
	for(int y = y0; y < y1; y++)
		for(int x = 0; x < c->w; x++) { /* :37-41, evaluation order ((A-B)-C)+D */
			const float pp = satf(c, x, y, r, r) - satf(c, x, y, r, 1) - satf(c, x, y, 1, r) + satf(c, x, y, 1, 1);
			const float pn = satf(c, x, y, r, -r) - satf(c, x, y, r, -1) - satf(c, x, y, 1, -r) + satf(c, x, y, 1, -1);
			const float np = satf(c, x, y, -r, r) - satf(c, x, y, -r, 1) - satf(c, x, y, -1, r) + satf(c, x, y, -1, 1);
			const float nn = satf(c, x, y, -r, -r) - satf(c, x, y, -r, -1) - satf(c, x, y, -1, -r) + satf(c, x, y, -1, -1);
			c->out[(size_t)y * c->w + x] = fminf_cl(fminf_cl(pp, nn), fminf_cl(pn, np)) / div;
		}
}

void vpo_blobcenter(const float* sat, int w, int h, int radius, float* out, int threads) {
	circle_ctx c = {sat, NULL, w, h, radius, out};
	parallel_span(circle_rows, &c, h, threads);
}

static inline uint32_t sati(const circle_ctx* c, int x, int y, int dx, int dy) {
	return (uint32_t)c->sati[(size_t)clampi(y + dy, 0, c->h - 1) * c->w + clampi(x + dx, 0, c->w - 1)];
}

static void circle_i32_rows(void* vc, int y0, int y1) {
	const circle_ctx* c = (const circle_ctx*)vc;
	const int r = c->r;
	const float div = (float)(r * r);
	for(int y = y0; y < y1; y++)
		for(int x = 0; x < c->w; x++) {
			/* wrapping uint32 box sums; the true box sum (|.| <= (r-1)^2 * 195075) always fits int32 */
			const float pp = (float)(int32_t)(sati(c, x, y, r, r) - sati(c, x, y, r, 1) - sati(c, x, y, 1, r) + sati(c, x, y, 1, 1));
			const float pn = (float)(int32_t)(sati(c, x, y, r, -r) - sati(c, x, y, r, -1) - sati(c, x, y, 1, -r) + sati(c, x, y, 1, -1));
			const float np = (float)(int32_t)(sati(c, x, y, -r, r) - sati(c, x, y, -r, 1) - sati(c, x, y, -1, r) + sati(c, x, y, -1, 1));
			const float nn = (float)(int32_t)(sati(c, x, y, -r, -r) - sati(c, x, y, -r, -1) - sati(c, x, y, -1, -r) + sati(c, x, y, -1, -1));
			c->out[(size_t)y * c->w + x] = fminf_cl(fminf_cl(pp, nn), fminf_cl(pn, np)) / div;
		}
}

void vpo_blobcenter_i32(const int32_t* sat, int w, int h, int radius, float* out, int threads) {
	circle_ctx c = {NULL, sat, w, h, radius, out};
	parallel_span(circle_i32_rows, &c, h, threads);
}

/* ---------------------------------------------------------------- blobList / blobScore -------------- */
static inline float circ_at(const float* circ, int w, int h, int x, int y) {
	return circ[(size_t)clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)];
}

/* blobList.cl:57-78 / blobScore.cl:41-66: disc statistics and score at one pixel */
static inline float disc_score(const uint8_t* flat, int w, int h, int px, int py, int radius, float circScore, uint32_t s1out[4], int* nout) {
	int n = 0;
	uint32_t s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
	const int sq = radius * radius;
	for(int y = -radius; y <= radius; y++)
		for(int x = -radius; x <= radius; x++)
			if(x * x + y * y <= sq) {
				const uint8_t* v = flat + 4 * ((size_t)clampi(py + y, 0, h - 1) * w + clampi(px + x, 0, w - 1));
				for(int k = 0; k < 4; k++) {
					s1[k] += v[k];
					s2[k] += (uint32_t)v[k] * v[k];
				}
				n++;
			}
	const float nf = (float)n;
	float sd[3];
	for(int k = 0; k < 3; k++) {
		const float a = (float)s1[k];
		const float q = a * a / nf;
		sd[k] = sqrtf(((float)s2[k] - q) / nf);
	}
	memcpy(s1out, s1, sizeof(s1));
	*nout = n;
	return circScore / (sd[0] + sd[1] + sd[2]);
}

void vpo_bloblist(const uint8_t* flat, const float* circ, int w, int h, float circThreshold, float minScore, int radius,
		int maxMatches, VpoMatch* out, uint32_t* idx, int32_t counter[3]) {
	counter[0] = counter[1] = counter[2] = 0;
	for(int y = 0; y < h; y++)
		for(int x = 0; x < w; x++) {
			const float c = circ[(size_t)y * w + x];
			if(c < circThreshold) continue; /* :39 (NaN passes) */
			const float nx = circ_at(circ, w, h, x - 1, y), px = circ_at(circ, w, h, x + 1, y);
			const float ny = circ_at(circ, w, h, x, y - 1), py = circ_at(circ, w, h, x, y + 1);
			if(nx > c || px > c || ny > c || py > c) { /* :47-55 strict > */
				counter[2]++;
				continue;
			}
			uint32_t s1[4];
			int n;
			const float score = disc_score(flat, w, h, x, y, radius, c, s1, &n);
			if(score < minScore) { /* :79-82 */
				counter[1]++;
				continue;
			}
			const int i = counter[0]++; /* :87 */
			if(i >= maxMatches) continue;
			VpoMatch m; /* :93-102 */
			m.x = (float)x + 0.5f * (nx - px) / (nx - 2 * c + px);
			m.y = (float)y + 0.5f * (ny - py) / (ny - 2 * c + py);
			const uint8_t* ctr = flat + 4 * ((size_t)y * w + x);
			for(int k = 0; k < 3; k++) {
				m.color[k] = (uint8_t)(s1[k] / (uint32_t)n);
				m.center[k] = ctr[k];
			}
			m.circ = c;
			m.score = score;
			memcpy(&out[i], &m, sizeof(m));
			if(idx) idx[i] = (uint32_t)((size_t)y * w + x);
		}
}

typedef struct { const uint8_t* flat; const float* circ; int w, h; float thr; int radius; float* score; } score_ctx;

static void score_rows(void* vc, int y0, int y1) {
	const score_ctx* s = (const score_ctx*)vc;
	const int w = s->w, h = s->h;
	for(int y = y0; y < y1; y++)
		for(int x = 0; x < w; x++) {
			const float c = s->circ[(size_t)y * w + x];
			float v = -INFINITY; /* blobScore.cl:26-29,37-38 */
			if(!(c < s->thr) && !(circ_at(s->circ, w, h, x - 1, y) > c || circ_at(s->circ, w, h, x + 1, y) > c || circ_at(s->circ, w, h, x, y - 1) > c || circ_at(s->circ, w, h, x, y + 1) > c)) {
				uint32_t s1[4];
				int n;
				v = disc_score(s->flat, w, h, x, y, s->radius, c, s1, &n);
			}
			s->score[(size_t)y * w + x] = v;
		}
}

void vpo_blobscore(const uint8_t* flat, const float* circ, int w, int h, float circThreshold, int radius, float* score, int threads) {
	score_ctx s = {flat, circ, w, h, circThreshold, radius, score};
	parallel_span(score_rows, &s, h, threads);
}

/* ---------------------------------------------------------------- blobCenter (legacy, non-SAT) ------- */
typedef struct { const float* g; int w, h, minr, maxr; float* out; } circz_ctx;

static void circz_rows(void* vc, int y0, int y1) { /* blobCenter.cl:29-63 */
	const circz_ctx* c = (const circz_ctx*)vc;
	const int w = c->w, h = c->h, R = c->maxr;
	const float sq = ((float)R + 0.5f) * ((float)R + 0.5f);
	for(int py = y0; py < y1; py++)
		for(int px = 0; px < w; px++) {
			int n = 0;
			float pp = 0.f, pn = 0.f, np = 0.f, nn = 0.f;
			for(int y = 1; y <= R; y++)
				for(int x = 1; x <= R; x++)
					if((float)(x * x + y * y) <= sq) { /* int compared against float :48 */
						np += c->g[(size_t)clampi(py + y, 0, h - 1) * w + clampi(px - x, 0, w - 1)];
						pp += c->g[(size_t)clampi(py + y, 0, h - 1) * w + clampi(px + x, 0, w - 1)];
						nn += c->g[(size_t)clampi(py - y, 0, h - 1) * w + clampi(px - x, 0, w - 1)];
						pn += c->g[(size_t)clampi(py - y, 0, h - 1) * w + clampi(px + x, 0, w - 1)];
						n++;
					}
			const float nf = (float)n;
			pp /= nf; nn /= nf; pn /= nf; np /= nf;
			c->out[(size_t)py * w + px] = fminf_cl(fminf_cl(pp, nn), fminf_cl(-pn, -np));
		}
}

void vpo_circularize(const float* graddot, int w, int h, int minRadius, int maxRadius, float* out, int threads) {
	circz_ctx c = {graddot, w, h, minRadius, maxRadius, out};
	parallel_span(circz_rows, &c, h, threads);
}

/* ---------------------------------------------------------------- debug / stream colour kernels ------- */
/* The colour of one quad-resolution pixel: quad2rgba.cl:27-50 == quad2nv12.cl:27-50.  BGR reads texel (x, y) itself
 * (read_imageui with int2 coordinates); Bayer samples the four planes at (x +- 0.25, y +- 0.25) like resampling.cl:64-79,
 * without the dRGB transform. */
static inline void quad_colour(const uint8_t* const p[4], int qw, int qh, int layout, int mode, int x, int y, uint32_t* r, uint32_t* g, uint32_t* b) {
	const float fx = (float)x, fy = (float)y;
	if(layout == VPO_BGR) {
		const size_t o = (size_t)y * qw + x;
		*r = p[2][o];
		*g = p[1][o];
		*b = p[0][o];
	} else if(layout == VPO_RGGB) {
		*r = sample_u8(p[0], qw, qh, fx + 0.25f, fy + 0.25f, mode);
		*g = sample_u8(p[1], qw, qh, fx - 0.25f, fy + 0.25f, mode) / 2 + sample_u8(p[2], qw, qh, fx + 0.25f, fy - 0.25f, mode) / 2;
		*b = sample_u8(p[3], qw, qh, fx - 0.25f, fy - 0.25f, mode);
	} else {
		*r = sample_u8(p[1], qw, qh, fx - 0.25f, fy + 0.25f, mode);
		*g = sample_u8(p[0], qw, qh, fx + 0.25f, fy + 0.25f, mode) / 2 + sample_u8(p[3], qw, qh, fx - 0.25f, fy - 0.25f, mode) / 2;
		*b = sample_u8(p[2], qw, qh, fx + 0.25f, fy - 0.25f, mode);
	}
}
static inline uint8_t sat_u8_u(uint32_t v) { return v > 255u ? 255 : (uint8_t)v; }
static inline uint8_t sat_u8_i(int v) { return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v); }
/* BT.601 limited range, integer (quad2nv12.cl:52,56-57 == rgba2nv12.cl:26,29-30): uint arithmetic for Y, int arithmetic with
 * C division (truncation toward zero) for U and V */
static inline uint8_t nv12_y(uint32_t r, uint32_t g, uint32_t b) { return sat_u8_u((66 * r + 129 * g + 25 * b) / 256 + 16); }
static inline uint8_t nv12_u(uint32_t r, uint32_t g, uint32_t b) { return sat_u8_i((-38 * (int)r + -74 * (int)g + 112 * (int)b) / 256 + 128); }
static inline uint8_t nv12_v(uint32_t r, uint32_t g, uint32_t b) { return sat_u8_i((112 * (int)r + -94 * (int)g + -18 * (int)b) / 256 + 128); }

size_t vpo_nv12_bytes(int w, int h) { return (size_t)w * h + (size_t)w * ((h + 1) / 2); }

/* quad2rgba.cl:24-53 */
void vpo_quad2rgba(const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3, int qw, int qh, int layout, int sampler_mode, uint8_t* rgba) {
	const uint8_t* const p[4] = {p0, p1, p2, p3};
	for(int y = 0; y < qh; y++)
		for(int x = 0; x < qw; x++) {
			uint32_t r, g, b;
			quad_colour(p, qw, qh, layout, sampler_mode, x, y, &r, &g, &b);
			uint8_t* o = rgba + 4 * ((size_t)y * qw + x);
			o[0] = sat_u8_u(r);
			o[1] = sat_u8_u(g);
			o[2] = sat_u8_u(b);
			o[3] = 255;
		}
}

/* quad2nv12.cl:24-58.  Every pixel writes its Y; every pixel also writes the UV pair of its 2x2 block (:54-57), so on a
 * device the four writers race.  Raster-sequential execution: the last pixel of the block in raster order wins. */
void vpo_quad2nv12(const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3, int qw, int qh, int layout, int sampler_mode, uint8_t* nv12) {
	const uint8_t* const p[4] = {p0, p1, p2, p3};
	for(int y = 0; y < qh; y++)
		for(int x = 0; x < qw; x++) {
			uint32_t r, g, b;
			quad_colour(p, qw, qh, layout, sampler_mode, x, y, &r, &g, &b);
			nv12[(size_t)x + (size_t)y * qw] = nv12_y(r, g, b);
			const size_t uv = (size_t)qw * qh + (size_t)(x / 2) * 2 + (size_t)(y / 2) * qw;
			nv12[uv] = nv12_u(r, g, b);
			nv12[uv + 1] = nv12_v(r, g, b);
		}
}

/* rgba2nv12.cl:22-31 (kernel name rgb2nv12) */
void vpo_rgba2nv12(const uint8_t* rgba, int w, int h, uint8_t* nv12) {
	for(int y = 0; y < h; y++)
		for(int x = 0; x < w; x++) {
			const uint8_t* v = rgba + 4 * ((size_t)y * w + x);
			nv12[(size_t)x + (size_t)y * w] = nv12_y(v[0], v[1], v[2]);
			const size_t uv = (size_t)w * h + (size_t)(x / 2) * 2 + (size_t)(y / 2) * w;
			nv12[uv] = nv12_u(v[0], v[1], v[2]);
			nv12[uv + 1] = nv12_v(v[0], v[1], v[2]);
		}
}

/* f2nv12.cl:22-26: Y = convert_uchar_sat(value + 127) (round toward zero, NaN -> 0), chroma 127 */
void vpo_f2nv12(const float* img, int w, int h, uint8_t* nv12) {
	for(int y = 0; y < h; y++)
		for(int x = 0; x < w; x++) {
			const float v = img[(size_t)y * w + x] + 127;
			nv12[(size_t)x + (size_t)y * w] = !(v > 0.0f) ? 0 : (v >= 255.0f ? 255 : (uint8_t)v);
			nv12[(size_t)w * h + (size_t)x + (size_t)(y / 2) * w] = 127;
		}
}

/* ---------------------------------------------------------------- host readback ---------------------- */
void vpo_flat2field(float x, float y, float fieldScale, float extX, float extY, float* fx, float* fy) {
	*fx = x * fieldScale + extX; /* Perspective.cpp:127-129 */
	*fy = y * fieldScale + extY;
}

/* ---------------------------------------------------------------- whole chain ------------------------ */
int vpo_frame(const uint8_t* raw, int layout, int qw, int qh, const VpoCameraModel* m, float maxRobotHeight,
		float fieldScale, float offX, float offY, int sampler_mode, int fw, int fh, int o, int r, int R,
		float circThreshold, float minScore, int maxMatches, int sat_mode, int threads,
		uint8_t* planes, uint8_t* flat, float* graddot, float* tmp, float* sat, float* blobcenter,
		VpoMatch* out, uint32_t* idx, int32_t counter[3]) {
	const size_t q = (size_t)qw * qh;
	uint8_t* p0 = planes, *p1 = planes + q, *p2 = planes + 2 * q, *p3 = planes + 3 * q;
	vpo_raw2quad(raw, layout, qw, qh, p0, p1, p2, p3);                                           /* Resources.cpp:138-143 */
	vpo_resample(p0, p1, p2, p3, qw, qh, layout, m, maxRobotHeight, fieldScale, offX, offY, sampler_mode, flat, fw, fh, threads); /* :159 */
	vpo_graddot(flat, fw, fh, o, graddot, threads);                                             /* :160 */
	if(sat_mode == 0) {
		vpo_sat_horizontal(graddot, fw, fh, tmp, threads);                                      /* :161 */
		vpo_sat_vertical(tmp, fw, fh, sat, threads);                                            /* :162 */
		vpo_blobcenter(sat, fw, fh, r, blobcenter, threads);                                    /* :163 */
	} else {
		vpo_sat_i32(graddot, fw, fh, (int32_t*)sat);
		vpo_blobcenter_i32((const int32_t*)sat, fw, fh, r, blobcenter, threads);
	}
	vpo_bloblist(flat, blobcenter, fw, fh, circThreshold, minScore, R, maxMatches, out, idx, counter); /* main.cpp:289 */
	return counter[0];
}
