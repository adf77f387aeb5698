// Synthetic camera frames for parity tests and the benchmark.
//
// Not derived from the reference (which ships no recorded frames): a flat scene at robot height
// (green carpet, 5-disc robot patterns at pattern.h:47-53 offsets, orange balls, white field lines) is
// rendered through the INVERSE of the pinhole + 1-coefficient radial camera model that resampling.cl:29-47
// applies, into a Bayer mosaic (or BGR8), plus Gaussian sensor noise.  Deterministic for a given seed.
//
// Coordinate convention (what resampling.cl:64-79 implies): quad-plane coordinate q of sensor pixel s is
// q = (s + 0.5) / 2 on both axes for Bayer layouts, q = s + 0.5 for BGR.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "vp_hip.h"

namespace {

struct Disc {
	double x, y, r;
	uint8_t rgb[3];
};

struct Scene {
	std::vector<Disc> discs;
	double fieldLen, fieldWid, lineHalf, circleR;
	// uniform grid over the discs
	double gx0, gy0, cell;
	int gw, gh;
	std::vector<std::vector<int>> grid;
};

const uint8_t CARPET[3] = {40, 110, 50}, YELLOW[3] = {230, 200, 30}, BLUE[3] = {30, 60, 220}, GREEN[3] = {40, 230, 80},
		PINK[3] = {240, 50, 150}, ORANGE[3] = {240, 120, 20}, WHITE[3] = {245, 245, 245};

// butterfly pattern offsets, mm (pattern.h:47-53) and green/pink codes (pattern.h:19-36)
const double PATTERN_POS[5][2] = {{0, 0}, {35, 54.772}, {-54.772, 35}, {-54.772, -35}, {35, -54.772}};
const int PATTERNS[16] = {0b0100, 0b1100, 0b1101, 0b0101, 0b0010, 0b1010, 0b1011, 0b0011, 0b1111, 0b0000, 0b0110, 0b1001, 0b1110, 0b1000, 0b0111, 0b0001};

void buildScene(Scene& s, uint32_t seed, int nRobots, int nBalls, double fieldLen, double fieldWid) {
	std::mt19937 rng(seed);
	auto uni = [&](double lo, double hi) { return lo + (hi - lo) * (rng() / 4294967296.0); };
	s.fieldLen = fieldLen;
	s.fieldWid = fieldWid;
	s.lineHalf = 5.0;
	s.circleR = std::min(fieldLen, fieldWid) * 0.07;
	std::vector<std::pair<double, double>> centres;
	auto place = [&](double margin, double minDist, double& x, double& y) {
		for(int tries = 0; tries < 200; tries++) {
			x = uni(-fieldLen / 2 + margin, fieldLen / 2 - margin);
			y = uni(-fieldWid / 2 + margin, fieldWid / 2 - margin);
			bool ok = true;
			for(auto& c : centres)
				if(std::hypot(c.first - x, c.second - y) < minDist) ok = false;
			if(ok) break;
		}
		centres.emplace_back(x, y);
	};
	for(int i = 0; i < nRobots; i++) {
		double x, y;
		place(120.0, 260.0, x, y);
		const double a = uni(0, 2 * M_PI), ca = std::cos(a), sa = std::sin(a);
		const int id = i % 16;
		for(int b = 0; b < 5; b++) {
			Disc d;
			d.x = x + ca * PATTERN_POS[b][0] - sa * PATTERN_POS[b][1];
			d.y = y + sa * PATTERN_POS[b][0] + ca * PATTERN_POS[b][1];
			d.r = b == 0 ? 25.0 : 20.0; // CENTER_BLOB_RADIUS / SIDE_BLOB_RADIUS (pattern.h:55-56)
			const uint8_t* c = b == 0 ? ((i & 1) ? BLUE : YELLOW) : (((PATTERNS[id] >> (4 - b)) & 1) ? GREEN : PINK);
			std::memcpy(d.rgb, c, 3);
			s.discs.push_back(d);
		}
	}
	for(int i = 0; i < nBalls; i++) {
		Disc d;
		place(60.0, 200.0, d.x, d.y);
		d.r = 21.5;
		std::memcpy(d.rgb, ORANGE, 3);
		s.discs.push_back(d);
	}
	s.cell = 100.0;
	s.gx0 = -fieldLen / 2 - 200;
	s.gy0 = -fieldWid / 2 - 200;
	s.gw = (int)std::ceil((fieldLen + 400) / s.cell);
	s.gh = (int)std::ceil((fieldWid + 400) / s.cell);
	s.grid.assign((size_t)s.gw * s.gh, {});
	for(int i = 0; i < (int)s.discs.size(); i++) {
		const Disc& d = s.discs[i];
		const int x0 = (int)std::floor((d.x - d.r - 10 - s.gx0) / s.cell), x1 = (int)std::floor((d.x + d.r + 10 - s.gx0) / s.cell);
		const int y0 = (int)std::floor((d.y - d.r - 10 - s.gy0) / s.cell), y1 = (int)std::floor((d.y + d.r + 10 - s.gy0) / s.cell);
		for(int y = std::max(0, y0); y <= std::min(s.gh - 1, y1); y++)
			for(int x = std::max(0, x0); x <= std::min(s.gw - 1, x1); x++)
				s.grid[(size_t)y * s.gw + x].push_back(i);
	}
}

inline double cover(double signedInside, double px) { // analytic edge anti-aliasing
	return std::clamp(0.5 + signedInside / px, 0.0, 1.0);
}

void shade(const Scene& s, double X, double Y, double pxmm, double rgb[3]) {
	for(int k = 0; k < 3; k++) rgb[k] = CARPET[k];
	// white lines: touch lines, goal lines, halfway line, centre circle
	const double hl = s.fieldLen / 2, hw = s.fieldWid / 2;
	double lineCov = 0.0;
	if(std::fabs(Y) <= hw + s.lineHalf) {
		lineCov = std::max(lineCov, cover(s.lineHalf - std::fabs(std::fabs(X) - hl), pxmm));
		lineCov = std::max(lineCov, cover(s.lineHalf - std::fabs(X), pxmm));
	}
	if(std::fabs(X) <= hl + s.lineHalf) lineCov = std::max(lineCov, cover(s.lineHalf - std::fabs(std::fabs(Y) - hw), pxmm));
	lineCov = std::max(lineCov, cover(s.lineHalf - std::fabs(std::hypot(X, Y) - s.circleR), pxmm));
	for(int k = 0; k < 3; k++) rgb[k] += lineCov * (WHITE[k] - rgb[k]);

	const int cx = (int)std::floor((X - s.gx0) / s.cell), cy = (int)std::floor((Y - s.gy0) / s.cell);
	if(cx < 0 || cy < 0 || cx >= s.gw || cy >= s.gh) return;
	for(int i : s.grid[(size_t)cy * s.gw + cx]) {
		const Disc& d = s.discs[i];
		const double c = cover(d.r - std::hypot(X - d.x, Y - d.y), pxmm);
		if(c > 0.0)
			for(int k = 0; k < 3; k++) rgb[k] += c * (d.rgb[k] - rgb[k]);
	}
}

// inverse of resampling.cl:29-47 (closed form: CameraModel.cpp:137-172), double precision
inline bool image2field(const VpCameraModel& m, double qx, double qy, double height, double& X, double& Y) {
	double nx = (qx - m.p[0]) / m.f, ny = (qy - m.p[1]) / m.f;
	const double k = 1.0 + m.d * (nx * nx + ny * ny);
	nx *= k;
	ny *= k;
	// ray = R^T * (nx, ny, 1): r is the row-major field->image rotation
	const double rx = m.r[0] * nx + m.r[3] * ny + m.r[6];
	const double ry = m.r[1] * nx + m.r[4] * ny + m.r[7];
	const double rz = m.r[2] * nx + m.r[5] * ny + m.r[8];
	if(rz >= 0) return false;
	const double t = (height - m.c[2]) / rz;
	X = m.c[0] + t * rx;
	Y = m.c[1] + t * ry;
	return true;
}

} // namespace

// See include/vp_hip.h (synthetic input section).
extern "C" int vp_synth_frame(const VpCameraModel* model, int layout, int qw, int qh, float planeHeight, float fieldLen, float fieldWid,
		uint32_t seed, int nRobots, int nBalls, float noiseSigma, uint8_t* raw) {
	if(!model || !raw || qw <= 0 || qh <= 0 || layout < 0 || layout > 2) return VP_ERR_INVALID;
	Scene scene;
	buildScene(scene, seed, nRobots, nBalls, fieldLen, fieldWid);
	const bool bgr = layout == VP_LAYOUT_BGR;
	const int sw = bgr ? qw : 2 * qw, sh = bgr ? qh : 2 * qh;
	const double sub = bgr ? 1.0 : 0.5;

	// mm per sensor pixel at the image centre (for edge anti-aliasing)
	double X0, Y0, X1, Y1, pxmm = 5.0;
	if(image2field(*model, model->p[0], model->p[1], planeHeight, X0, Y0) && image2field(*model, model->p[0] + sub, model->p[1], planeHeight, X1, Y1))
		pxmm = std::max(1e-3, std::hypot(X1 - X0, Y1 - Y0));

	const unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
	std::vector<std::thread> pool;
	for(unsigned t = 0; t < nthreads; t++)
		pool.emplace_back([&, t]() {
			for(int sy = (int)t; sy < sh; sy += (int)nthreads) {
				// one generator per row => identical output for any thread count
				std::mt19937 rng(seed * 2654435761u + 0x9E3779B9u * (uint32_t)(sy + 1));
				auto gauss = [&]() { // Box-Muller on raw 32-bit draws
					const double u1 = (rng() + 1.0) / 4294967297.0, u2 = rng() / 4294967296.0;
					return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
				};
				for(int sx = 0; sx < sw; sx++) {
					double rgb[3] = {(double)CARPET[0], (double)CARPET[1], (double)CARPET[2]};
					double X, Y;
					if(image2field(*model, (sx + 0.5) * sub, (sy + 0.5) * sub, planeHeight, X, Y)) shade(scene, X, Y, pxmm, rgb);
					auto put = [&](double v) {
						v += noiseSigma * gauss();
						return (uint8_t)std::clamp((int)std::lrint(v), 0, 255);
					};
					if(bgr) {
						uint8_t* o = raw + 3 * ((size_t)sy * sw + sx);
						o[0] = put(rgb[2]);
						o[1] = put(rgb[1]);
						o[2] = put(rgb[0]);
					} else {
						const int ex = sx & 1, ey = sy & 1;
						int ch; // 0 R, 1 G, 2 B
						if(layout == VP_LAYOUT_RGGB) ch = ey == 0 ? (ex == 0 ? 0 : 1) : (ex == 0 ? 1 : 2);
						else ch = ey == 0 ? (ex == 0 ? 1 : 0) : (ex == 0 ? 2 : 1); // GRBG
						raw[(size_t)sy * sw + sx] = put(rgb[ch]);
					}
				}
			}
		});
	for(auto& th : pool) th.join();
	return VP_OK;
}

// Ground truth of a synthetic frame: the discs buildScene() places for this seed, in field coordinates on the plane the scene is
// rendered on.  out = n x {x, y, radius, r, g, b}; returns the number of discs (5 per robot, then the balls), at most cap are written.
extern "C" int vp_synth_truth(uint32_t seed, int nRobots, int nBalls, float fieldLen, float fieldWid, float* out, int cap) {
	if(nRobots < 0 || nBalls < 0 || (cap > 0 && !out)) return VP_ERR_INVALID;
	Scene scene;
	buildScene(scene, seed, nRobots, nBalls, fieldLen, fieldWid);
	const int n = (int)scene.discs.size();
	for(int i = 0; i < std::min(n, cap); i++) {
		const Disc& d = scene.discs[i];
		out[6 * i] = (float)d.x;
		out[6 * i + 1] = (float)d.y;
		out[6 * i + 2] = (float)d.r;
		out[6 * i + 3] = d.rgb[0];
		out[6 * i + 4] = d.rgb[1];
		out[6 * i + 5] = d.rgb[2];
	}
	return n;
}

// Nadir camera over the field centre: q = (w 0, x -1, y 0, z 0) (CameraModel.h:50) => R = diag(1, -1, -1);
// focal length such that `visibleLen` mm of the plane z = planeHeight spans the quad-plane width (distortion ignored).
extern "C" void vp_synth_camera(int qw, int qh, float camHeight, float planeHeight, float visibleLen, float distortion, VpCameraModel* out) {
	std::memset(out, 0, sizeof(*out));
	out->shape[0] = qw;
	out->shape[1] = qh;
	out->f = (float)(qw * 0.5 * (camHeight - planeHeight) / (visibleLen * 0.5));
	out->p[0] = qw * 0.5f;
	out->p[1] = qh * 0.5f;
	out->d = distortion;
	out->r[0] = 1.f;
	out->r[4] = -1.f;
	out->r[8] = -1.f;
	out->c[2] = camHeight;
}
