// See CameraModel.h.  Restates src/CameraModel.cpp and src/Perspective.cpp without Eigen / protobuf.
#include "CameraModel.h"

#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <limits>

namespace vp {

namespace {

inline Vec3f cross(const Vec3f& a, const Vec3f& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// unit-quaternion vector rotation in the form Eigen's Quaternion::_transformVector uses:
// uv = 2 * (q.vec x v);  v + w*uv + q.vec x uv
inline Vec3f rotate(const Quatf& q, const Vec3f& v) {
	const Vec3f qv{q.x, q.y, q.z};
	Vec3f uv = cross(qv, v);
	uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
	const Vec3f c = cross(qv, uv);
	return {v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z};
}

inline Quatf inverse(const Quatf& q) { // conjugate / squared norm
	const float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
	if(n2 > 0.f) return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
	return {0.f, 0.f, 0.f, 0.f};
}

inline Mat3f toRotationMatrix(const Quatf& q) {
	const float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
	const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
	const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
	const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
	Mat3f r;
	r.m[0] = 1.f - (tyy + tzz); r.m[1] = txy - twz;         r.m[2] = txz + twy;
	r.m[3] = txy + twz;         r.m[4] = 1.f - (txx + tzz); r.m[5] = tyz - twx;
	r.m[6] = txz - twy;         r.m[7] = tyz + twx;         r.m[8] = 1.f - (txx + tyy);
	return r;
}

inline Vec3f mul(const Mat3f& m, const Vec3f& v) {
	return {m.m[0] * v.x + m.m[1] * v.y + m.m[2] * v.z, m.m[3] * v.x + m.m[4] * v.y + m.m[5] * v.z, m.m[6] * v.x + m.m[7] * v.y + m.m[8] * v.z};
}

} // namespace

float goalBoundaryWidth(const FieldSize& field) {
	return field.has_boundary_width_goal_line ? (float)field.boundary_width_goal_line : (float)field.boundary_width;
}

// CameraModel.cpp:24-60: tile the field over camAmount cameras, column-major camera ids
void visibleFieldExtentEstimation(const int camId, const int camAmount, const FieldSize& field, const bool withBoundary, Vec2f& min, Vec2f& max) {
	const float fieldSize[2] = {(float)field.field_length, (float)field.field_width};
	int size[2] = {1, 1};
	for(int i = camAmount; i > 1; i /= 2) {
		if(fieldSize[0] / (float)size[0] >= fieldSize[1] / (float)size[1]) size[0] *= 2;
		else size[1] *= 2;
	}
	int pos[2] = {0, 0};
	for(int i = camId % camAmount; i > 0; i--) {
		pos[1]++;
		if(pos[1] == size[1]) {
			pos[1] = 0;
			pos[0]++;
		}
	}
	const float ext[2] = {fieldSize[0] / (float)size[0], fieldSize[1] / (float)size[1]};
	min = {ext[0] * (float)pos[0] - fieldSize[0] / 2, ext[1] * (float)pos[1] - fieldSize[1] / 2};
	max = {min.x + ext[0], min.y + ext[1]};
	if(withBoundary) {
		if(pos[0] == 0) min.x -= goalBoundaryWidth(field);
		if(pos[1] == 0) min.y -= (float)field.boundary_width;
		if(pos[0] == size[0] - 1) max.x += goalBoundaryWidth(field);
		if(pos[1] == size[1] - 1) max.y += (float)field.boundary_width;
	}
}

CameraModel::CameraModel(): focalLength(1224.0f), principalPoint{612.f, 512.f}, size{1224, 1024} { // CameraModel.cpp:63-65
	updateDerived();
}

CameraModel::CameraModel(const Vec2i& size_, int camId, int camAmount, const float cameraHeight, const FieldSize& field): // :67-84
		principalPoint{(float)size_.x / 2, (float)size_.y / 2}, size(size_) {
	Vec2f min, max;
	visibleFieldExtentEstimation(camId, camAmount, field, true, min, max);
	pos.x = min.x / 2 + max.x / 2;
	pos.y = min.y / 2 + max.y / 2;
	if(cameraHeight != 0.f) pos.z = cameraHeight;
	// whole extent visible: long image side over long extent side
	const float orderedSize[2] = {(float)std::max(size.x, size.y), (float)std::min(size.x, size.y)};
	const float ex = max.x - min.x, ey = max.y - min.y;
	const float orderedExtent[2] = {std::max(ex, ey), std::min(ex, ey)};
	const float a = (orderedSize[0] - principalPoint.x) * pos.z / orderedExtent[0];
	const float b = (orderedSize[1] - principalPoint.y) * pos.z / orderedExtent[1];
	focalLength = std::min(a, b) * 2;
	updateDerived();
}

CameraModel::CameraModel(const CameraCalibration& calib): // :86-94; Quaternionf(w=q3, x=q0, y=q1, z=q2)
		focalLength(calib.focal_length), principalPoint{calib.principal_point_x, calib.principal_point_y}, distortionK2(calib.distortion),
		size{(int)calib.pixel_image_width, (int)calib.pixel_image_height} {
	f2iOrientation = {calib.q3, calib.q0, calib.q1, calib.q2};
	pos = rotate(inverse(f2iOrientation), Vec3f{-calib.tx, -calib.ty, -calib.tz});
	updateDerived();
}

CameraCalibration CameraModel::getProto(int camId) const { // :96-117
	CameraCalibration c;
	c.camera_id = (uint32_t)camId;
	c.focal_length = focalLength;
	c.principal_point_x = principalPoint.x;
	c.principal_point_y = principalPoint.y;
	c.distortion = distortionK2;
	c.q0 = f2iOrientation.x;
	c.q1 = f2iOrientation.y;
	c.q2 = f2iOrientation.z;
	c.q3 = f2iOrientation.w;
	const Vec3f iPos = rotate(f2iOrientation, Vec3f{-pos.x, -pos.y, -pos.z});
	c.tx = iPos.x;
	c.ty = iPos.y;
	c.tz = iPos.z;
	c.has_derived = true;
	c.derived_camera_world_tx = pos.x;
	c.derived_camera_world_ty = pos.y;
	c.derived_camera_world_tz = pos.z;
	c.pixel_image_width = (uint32_t)size.x;
	c.pixel_image_height = (uint32_t)size.y;
	return c;
}

void CameraModel::updateDerived() { // :119-123
	const float n = std::sqrt(f2iOrientation.w * f2iOrientation.w + f2iOrientation.x * f2iOrientation.x + f2iOrientation.y * f2iOrientation.y + f2iOrientation.z * f2iOrientation.z);
	if(n > 0.f) f2iOrientation = {f2iOrientation.w / n, f2iOrientation.x / n, f2iOrientation.y / n, f2iOrientation.z / n};
	i2fOrientation = toRotationMatrix(inverse(f2iOrientation));
	f2iRotation = toRotationMatrix(f2iOrientation);
	f2iTranslation = mul(f2iRotation, Vec3f{-pos.x, -pos.y, -pos.z});
}

void CameraModel::ensureSize(const Vec2i& newSize) { // :125-136
	if(size == newSize) return;
	const float factor = (float)newSize.x / (float)size.x;
	if((float)size.y * factor != (float)newSize.y) fprintf(stderr, "[WARN] ensureSize with diverging aspect ratios\n");
	size = newSize;
	focalLength *= factor;
	principalPoint = {principalPoint.x * factor, principalPoint.y * factor};
}

Vec2f CameraModel::normalizeUndistort(const Vec2f& p) const { // :138-142
	Vec2f n{(p.x - principalPoint.x) / focalLength, (p.y - principalPoint.y) / focalLength};
	const float k = 1.0f + distortionK2 * (n.x * n.x + n.y * n.y);
	return {n.x * k, n.y * k};
}

Vec2f CameraModel::undistort(const Vec2f& p) const { // :144-146
	const Vec2f n = normalizeUndistort(p);
	return {n.x * focalLength + principalPoint.x, n.y * focalLength + principalPoint.y};
}

Vec2f CameraModel::field2image(const Vec3f& p) const { // :148-159 -- NB 10 iterations on the host, 8 on the device
	const Vec3f r = mul(f2iRotation, p);
	const Vec3f camRay{r.x + f2iTranslation.x, r.y + f2iTranslation.y, r.z + f2iTranslation.z};
	const Vec2f original{camRay.x / camRay.z, camRay.y / camRay.z};
	Vec2f n = original;
	for(int i = 0; i < 10; i++) {
		const float d = 1 + distortionK2 * (n.x * n.x + n.y * n.y);
		n = {original.x / d, original.y / d};
	}
	return {focalLength * n.x + principalPoint.x, focalLength * n.y + principalPoint.y};
}

Vec3f CameraModel::image2field(const Vec2f& p, const float height) const { // :161-174
	const Vec2f n = normalizeUndistort(p);
	Vec3f camRay = mul(i2fOrientation, Vec3f{n.x, n.y, 1.0f});
	if(camRay.z >= 0) return {NAN, NAN, NAN}; // camera below the carpet
	const float t = (-pos.z + height) / camRay.z;
	return {camRay.x * t + pos.x, camRay.y * t + pos.y, height};
}

void Perspective::setGeometry(const FieldSize& f, const CameraCalibration* calibs, int nCalibs) {
	pendingField = f;
	nPendingCalibs = std::min(nCalibs, 8);
	for(int i = 0; i < nPendingCalibs; i++) pendingCalibs[i] = calibs[i];
	pendingVersion++;
}

namespace {
inline void updateExtent(Vec4f& e, const Vec3f& p) { // Perspective.cpp:23-33 (NaN never updates)
	if(p.x < e[0]) e[0] = p.x;
	if(p.x > e[1]) e[1] = p.x;
	if(p.y < e[2]) e[2] = p.y;
	if(p.y > e[3]) e[3] = p.y;
}
} // namespace

void Perspective::geometryCheck(const int width, const int height, const double maxBotHeight, const float resamplingFactor) {
	const Vec2i size{width, height};
	if(pendingVersion == geometryVersion && model.size == size) return; // Perspective.cpp:37-38

	bool calibFound = false;
	for(int i = 0; i < nPendingCalibs; i++)
		if(pendingCalibs[i].camera_id == camId) { // :41-56 (the re-broadcast of a completed calib is network I/O: out of scope)
			calibFound = true;
			model = CameraModel(pendingCalibs[i]);
			break;
		}
	if(!calibFound) {
		if(nPendingCalibs == 0) geometryVersion = 0; // :59-63
		return;
	}

	model.ensureSize(size);
	geometryVersion = pendingVersion;
	field = pendingField;
	minBlobRadius = std::min({CENTER_BLOB_RADIUS, SIDE_BLOB_RADIUS, field.ball_radius}); // :69-70
	maxBlobRadius = std::max({CENTER_BLOB_RADIUS, SIDE_BLOB_RADIUS, field.ball_radius});

	// :72-91 mean in-field pixel footprint, sequential binary32 accumulation
	const float h = (float)maxBotHeight;
	const float limX = (float)field.field_length / 2.f + goalBoundaryWidth(field), limY = (float)field.field_width / 2.f + (float)field.boundary_width;
	float fieldScaleSum = 0;
	int n = 0;
	for(int y = 0; y < height - 1; y++)
		for(int x = 0; x < width - 1; x++) {
			const Vec3f p = model.image2field({(float)x, (float)y}, h);
			if(std::fabs(p.x) < limX && std::fabs(p.y) < limY) {
				const Vec3f px = model.image2field({(float)(x + 1), (float)y}, h), py = model.image2field({(float)x, (float)(y + 1)}, h);
				const float dx = std::sqrt((px.x - p.x) * (px.x - p.x) + (px.y - p.y) * (px.y - p.y));
				const float dy = std::sqrt((py.x - p.x) * (py.x - p.x) + (py.y - p.y) * (py.y - p.y));
				fieldScaleSum += dx + dy;
				n += 2;
			}
		}
	fieldScale = fieldScaleSum / (float)n * resamplingFactor;

	// :94-113 image border projected onto the field plane, clamped to field + boundary + tolerance
	const Vec3f center = model.image2field({0.0f, 0.0f}, h);
	visibleFieldExtent[0] = visibleFieldExtent[1] = center.x;
	visibleFieldExtent[2] = visibleFieldExtent[3] = center.y;
	for(int x = 0; x < width; x++) {
		updateExtent(visibleFieldExtent, model.image2field({(float)x, 0.0f}, h));
		updateExtent(visibleFieldExtent, model.image2field({(float)x, (float)height - 1.0f}, h));
	}
	for(int y = 0; y < height; y++) {
		updateExtent(visibleFieldExtent, model.image2field({0.0f, (float)y}, h));
		updateExtent(visibleFieldExtent, model.image2field({(float)width - 1.0f, (float)y}, h));
	}
	const float halfLength = (float)field.field_length / 2.0f + goalBoundaryWidth(field) + geometryTolerance;
	const float halfWidth = (float)field.field_width / 2.0f + (float)field.boundary_width + geometryTolerance;
	visibleFieldExtent[0] = std::max(visibleFieldExtent[0], -halfLength);
	visibleFieldExtent[1] = std::min(visibleFieldExtent[1], halfLength);
	visibleFieldExtent[2] = std::max(visibleFieldExtent[2], -halfWidth);
	visibleFieldExtent[3] = std::min(visibleFieldExtent[3], halfWidth);

	// :115-122 flat size, forced even
	reprojectedFieldSize.x = (int)std::rint((visibleFieldExtent[1] - visibleFieldExtent[0]) / fieldScale);
	reprojectedFieldSize.y = (int)std::rint((visibleFieldExtent[3] - visibleFieldExtent[2]) / fieldScale);
	if(reprojectedFieldSize.x % 2) reprojectedFieldSize.x++;
	if(reprojectedFieldSize.y % 2) reprojectedFieldSize.y++;
}

Vec2f Perspective::flat2field(const Vec2f& p) const { return {p.x * fieldScale + visibleFieldExtent[0], p.y * fieldScale + visibleFieldExtent[2]}; }
Vec2f Perspective::field2flat(const Vec2f& p) const { return {(p.x - visibleFieldExtent[0]) / fieldScale, (p.y - visibleFieldExtent[2]) / fieldScale}; }

VpCameraModel Perspective::getCLCameraModel() const { // Perspective.cpp:136-150
	VpCameraModel m;
	m.shape[0] = model.size.x;
	m.shape[1] = model.size.y;
	m.f = model.focalLength;
	m.p[0] = model.principalPoint.x;
	m.p[1] = model.principalPoint.y;
	m.d = model.distortionK2;
	for(int i = 0; i < 9; i++) m.r[i] = model.f2iRotation.m[i];
	m.c[0] = model.pos.x;
	m.c[1] = model.pos.y;
	m.c[2] = model.pos.z;
	return m;
}

} // namespace vp

// ---- C entry points for tests (ctypes) ------------------------------------------------------------------------------
extern "C" {

// Runs Perspective::geometryCheck for one calibration and returns the kernel parameters.
// out_f: {fieldScale, extent xmin, xmax, ymin, ymax, minBlobRadius, maxBlobRadius}; out_i: {fw, fh, gradOffset, circleRadius, listRadius, geometryVersion}
int vp_geometry_check(const float* calib /* focal, ppx, ppy, dist, q0..q3, tx, ty, tz */, int img_w, int img_h, int calib_w, int calib_h,
		const int* field /* length, width, boundary, boundary_goal_line(-1 = unset) */, float ball_radius, float geometry_tolerance,
		double max_bot_height, float resampling_factor, float* out_f, int* out_i, VpCameraModel* out_model) {
	vp::FieldSize f;
	f.field_length = field[0];
	f.field_width = field[1];
	f.boundary_width = field[2];
	f.has_boundary_width_goal_line = field[3] >= 0;
	f.boundary_width_goal_line = field[3] >= 0 ? field[3] : 0;
	f.ball_radius = ball_radius;
	vp::CameraCalibration c;
	c.camera_id = 0;
	c.focal_length = calib[0];
	c.principal_point_x = calib[1];
	c.principal_point_y = calib[2];
	c.distortion = calib[3];
	c.q0 = calib[4]; c.q1 = calib[5]; c.q2 = calib[6]; c.q3 = calib[7];
	c.tx = calib[8]; c.ty = calib[9]; c.tz = calib[10];
	c.pixel_image_width = (uint32_t)calib_w;
	c.pixel_image_height = (uint32_t)calib_h;
	vp::Perspective p(0, geometry_tolerance);
	p.setGeometry(f, &c, 1);
	p.geometryCheck(img_w, img_h, max_bot_height, resampling_factor);
	out_f[0] = p.fieldScale;
	for(int i = 0; i < 4; i++) out_f[1 + i] = p.visibleFieldExtent[i];
	out_f[5] = p.minBlobRadius;
	out_f[6] = p.maxBlobRadius;
	out_i[0] = p.reprojectedFieldSize.x;
	out_i[1] = p.reprojectedFieldSize.y;
	out_i[2] = p.gradOffset();
	out_i[3] = p.circleRadius();
	out_i[4] = p.listRadius();
	out_i[5] = p.geometryVersion;
	if(out_model) *out_model = p.getCLCameraModel();
	return 0;
}

} // extern "C"
