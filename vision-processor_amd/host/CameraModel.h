// Eigen-free restatement of the reference's geometry -> kernel-parameter producer
// (src/CameraModel.{h,cpp}, src/Perspective.{h,cpp}, src/pattern.h:55-56).  Same member names and call surface;
// Eigen vectors become small PODs and the protobuf messages become the plain structs below.
//
// Arithmetic is binary32 in the order written in the cited reference lines.  The reference host code cannot be
// built in this image (no Eigen/protobuf), so these values are NOT pinned against it: parity for this producer
// is "unpinned"; the kernels downstream consume whatever scalars they are given.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <vector>

#include "vp_hip.h"

namespace vp {

struct Vec2f { float x = 0.f, y = 0.f; };
struct Vec2i { int x = 0, y = 0; bool operator==(const Vec2i& o) const { return x == o.x && y == o.y; } int& operator[](int i) { return i ? y : x; } int operator[](int i) const { return i ? y : x; } }; // [] as Eigen::Vector2i (Resources.cpp:152: reprojectedFieldSize[0])
struct Vec3f { float x = 0.f, y = 0.f, z = 0.f; };
struct Vec4f { float v[4] = {0, 0, 0, 0}; float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };
struct Quatf { float w = 0.f, x = -1.f, y = 0.f, z = 0.f; };
struct Mat3f { float m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; float operator()(int r, int c) const { return m[3 * r + c]; } };

// SSL_GeometryCameraCalibration (proto/ssl_vision_geometry.proto) -- only the fields CameraModel reads/writes
struct CameraCalibration {
	uint32_t camera_id = 0;
	float focal_length = 0.f, principal_point_x = 0.f, principal_point_y = 0.f, distortion = 0.f;
	float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
	float tx = 0.f, ty = 0.f, tz = 0.f;
	bool has_derived = false;
	float derived_camera_world_tx = 0.f, derived_camera_world_ty = 0.f, derived_camera_world_tz = 0.f;
	uint32_t pixel_image_width = 0, pixel_image_height = 0;
};

// SSL_FieldLineSegment / SSL_FieldCircularArc (proto/ssl_vision_geometry.proto:14-41) -- the fields colorupdate.cpp:21-55 reads
struct FieldLine { Vec2f p1, p2; };
struct FieldArc { Vec2f center; float radius = 0.f, a1 = 0.f, a2 = 0.f; };

// SSL_GeometryFieldSize -- the fields Perspective/CameraModel and the hypothesis stage read
struct FieldSize {
	int field_length = 9000, field_width = 6000, boundary_width = 300;
	bool has_boundary_width_goal_line = false;
	int boundary_width_goal_line = 0;
	float ball_radius = 21.5f, max_robot_radius = 90.f;
	int line_thickness = 0; // optional int32 without a default (:69-70)
	std::vector<FieldLine> field_lines;
	std::vector<FieldArc> field_arcs;
};

const float CENTER_BLOB_RADIUS = 25.0f; // pattern.h:55
const float SIDE_BLOB_RADIUS = 20.0f;   // pattern.h:56

float goalBoundaryWidth(const FieldSize& field); // CameraModel.cpp:20-22
void visibleFieldExtentEstimation(int camId, int camAmount, const FieldSize& field, bool withBoundary, Vec2f& min, Vec2f& max); // :24-60

class CameraModel { // CameraModel.h:27-57
public:
	CameraModel();
	CameraModel(const Vec2i& size, int camId, int camAmount, float cameraHeight, const FieldSize& field);
	explicit CameraModel(const CameraCalibration& calib);
	CameraCalibration getProto(int camId) const;

	void updateDerived();
	void ensureSize(const Vec2i& newSize);

	Vec2f normalizeUndistort(const Vec2f& p) const;
	Vec2f undistort(const Vec2f& p) const;
	Vec2f field2image(const Vec3f& p) const;
	Vec3f image2field(const Vec2f& p, float height) const;

	float focalLength;
	Vec2f principalPoint;
	float distortionK2 = 0.0f;
	Vec3f pos{0.0f, 0.0f, 5000.0f};
	Quatf f2iOrientation; // (w 0, x -1, y 0, z 0): nadir
	Vec2i size;

	Mat3f f2iRotation;   // linear part of f2iTransformation
	Vec3f f2iTranslation; // f2iRotation * (-pos)
	Mat3f i2fOrientation;
};

class Perspective { // Perspective.h:32-58, fed by plain structs instead of a VisionSocket
public:
	Perspective(int camId, float geometryTolerance): camId((unsigned)camId), geometryTolerance(geometryTolerance) {}

	// The socket's geometry is replaced by setGeometry(): bumps the version like VisionSocket::getGeometryVersion.
	void setGeometry(const FieldSize& field, const CameraCalibration* calibs, int nCalibs);
	void geometryCheck(int width, int height, double maxBotHeight, float resamplingFactor); // Perspective.cpp:35-125

	Vec2f flat2field(const Vec2f& pos) const; // :127-129
	Vec2f field2flat(const Vec2f& pos) const; // :131-133
	VpCameraModel getCLCameraModel() const;   // :136-150

	// kernel scalars derived at the three call sites (Resources.cpp:160,163; main.cpp:289)
	int gradOffset() const { return (int)ceilf(maxBlobRadius / fieldScale) / 3; }
	int circleRadius() const { return (int)ceilf(minBlobRadius / fieldScale); }
	int listRadius() const { return (int)floorf(minBlobRadius / fieldScale); }

	FieldSize field;
	CameraModel model;
	Vec4f visibleFieldExtent; // xmin, xmax, ymin, ymax
	float fieldScale = 5.f;
	Vec2i reprojectedFieldSize;
	float minBlobRadius = 20.0f;
	float maxBlobRadius = 25.0f;
	int geometryVersion = 0;

private:
	const unsigned camId;
	const float geometryTolerance;
	FieldSize pendingField;
	CameraCalibration pendingCalibs[8];
	int nPendingCalibs = 0;
	int pendingVersion = 0;
};

} // namespace vp
