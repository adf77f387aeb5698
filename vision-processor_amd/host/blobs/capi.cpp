// Flat C binding of the blob -> detection stage (include/vp_host.h).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <string>
#include <thread>

#include "detection.h"
#include "vp_host.h"

struct VphDetector {
	vp::BlobResources r;
	vp::DetectionFrame frame;
	std::unique_ptr<vp::VisionSocket> socket;
	std::unique_ptr<vp::GCSocket> gcSocket;
	int socketGeometryApplied = 0; // version of the socket's geometry the perspective was last derived from
};

namespace {

vp::Vec3i v3(const int32_t* c) { return {c[0], c[1], c[2]}; }

vp::FieldSize fieldFrom(const int32_t field[5], float ball_radius, float max_robot_radius, const float* lines, int n_lines, const float* arcs, int n_arcs) {
	vp::FieldSize f;
	f.field_length = field[0];
	f.field_width = field[1];
	f.boundary_width = field[2];
	f.has_boundary_width_goal_line = field[3] >= 0;
	f.boundary_width_goal_line = field[3] >= 0 ? field[3] : 0;
	f.line_thickness = field[4];
	f.ball_radius = ball_radius;
	f.max_robot_radius = max_robot_radius;
	for(int i = 0; i < n_lines; i++) f.field_lines.push_back({{lines[4 * i], lines[4 * i + 1]}, {lines[4 * i + 2], lines[4 * i + 3]}});
	for(int i = 0; i < n_arcs; i++) f.field_arcs.push_back({{arcs[5 * i], arcs[5 * i + 1]}, arcs[5 * i + 2], arcs[5 * i + 3], arcs[5 * i + 4]});
	return f;
}

vp::CameraCalibration calibFrom(const float calib[11], int cam_id, int w, int h) {
	vp::CameraCalibration c;
	c.camera_id = (uint32_t)cam_id;
	c.focal_length = calib[0];
	c.principal_point_x = calib[1];
	c.principal_point_y = calib[2];
	c.distortion = calib[3];
	c.q0 = calib[4];
	c.q1 = calib[5];
	c.q2 = calib[6];
	c.q3 = calib[7];
	c.tx = calib[8];
	c.ty = calib[9];
	c.tz = calib[10];
	c.pixel_image_width = (uint32_t)w;
	c.pixel_image_height = (uint32_t)h;
	return c;
}

std::vector<vp::Match> matchesFromCl(const vp::Perspective& p, const VpMatch* in, int n) { // main.cpp:303-311
	std::vector<vp::Match> ms;
	ms.reserve((size_t)n);
	for(int i = 0; i < n; i++) {
		const VpMatch& m = in[i];
		ms.push_back({p.flat2field({m.x, m.y}), {m.color[0], m.color[1], m.color[2]}, {m.center[0], m.center[1], m.center[2]}, m.circ, m.score});
	}
	return ms;
}
void put3(int32_t* o, const vp::Vec3i& c) {
	o[0] = c.x;
	o[1] = c.y;
	o[2] = c.z;
}

} // namespace

extern "C" {

void vph_config_defaults(VphConfig* cfg) {
	const vp::BlobResources d;
	cfg->cam_id = d.camId;
	cfg->min_tracking_radius = d.minTrackingRadius;
	cfg->max_bot_acceleration = d.maxBotAcceleration;
	cfg->min_score = d.minScore;
	cfg->min_cam_edge_distance = d.minCamEdgeDistance;
	cfg->min_confidence = d.minConfidence;
	cfg->clipping_tolerance = d.clippingTolerance;
	cfg->geometry_tolerance = d.geometryTolerance;
	cfg->reference_force = d.referenceForce;
	cfg->history_force = d.historyForce;
	const vp::Vec3i refs[6] = {d.orangeReference, d.fieldReference, d.yellowReference, d.blueReference, d.greenReference, d.pinkReference};
	for(int i = 0; i < 6; i++) put3(cfg->reference_colors[i], refs[i]);
	cfg->max_bot_height = d.gcSocket->maxBotHeight;
	cfg->default_bot_height = d.gcSocket->defaultBotHeight;
	cfg->yellow_bot_height = d.gcSocket->yellowBotHeight;
	cfg->blue_bot_height = d.gcSocket->blueBotHeight;
}

VphDetector* vph_detector_create(const VphConfig* cfg) {
	if(!cfg) return nullptr;
	VphDetector* d = new VphDetector();
	vp::BlobResources& r = d->r;
	r.camId = cfg->cam_id;
	r.minTrackingRadius = cfg->min_tracking_radius;
	r.maxBotAcceleration = cfg->max_bot_acceleration;
	r.minScore = cfg->min_score;
	r.minCamEdgeDistance = cfg->min_cam_edge_distance;
	r.minConfidence = cfg->min_confidence;
	r.clippingTolerance = cfg->clipping_tolerance;
	r.geometryTolerance = cfg->geometry_tolerance;
	r.referenceForce = cfg->reference_force;
	r.historyForce = cfg->history_force;
	r.orange = r.orangeReference = v3(cfg->reference_colors[0]); // Resources.cpp:89-95
	r.field = r.fieldReference = v3(cfg->reference_colors[1]);
	r.yellow = r.yellowReference = v3(cfg->reference_colors[2]);
	r.blue = r.blueReference = v3(cfg->reference_colors[3]);
	r.green = r.greenReference = v3(cfg->reference_colors[4]);
	r.pink = r.pinkReference = v3(cfg->reference_colors[5]);
	r.fieldLineColor = r.fieldReference;
	r.gcSocket->maxBotHeight = cfg->max_bot_height;
	r.gcSocket->defaultBotHeight = cfg->default_bot_height;
	r.gcSocket->yellowBotHeight = cfg->yellow_bot_height;
	r.gcSocket->blueBotHeight = cfg->blue_bot_height;
	r.perspective = std::make_shared<vp::Perspective>(r.camId, r.geometryTolerance);
	return d;
}

void vph_detector_destroy(VphDetector* d) { delete d; }

int vph_detector_set_geometry(VphDetector* d, const float calib[11], int calib_w, int calib_h, int img_w, int img_h, float resampling_factor, const int32_t field[5],
		float ball_radius, float max_robot_radius, const float* lines, int n_lines, const float* arcs, int n_arcs) {
	if(!d || !calib || !field || img_w <= 1 || img_h <= 1 || calib_w <= 0 || calib_h <= 0 || n_lines < 0 || n_arcs < 0 || (n_lines && !lines) || (n_arcs && !arcs)) return VP_ERR_INVALID;
	const vp::FieldSize f = fieldFrom(field, ball_radius, max_robot_radius, lines, n_lines, arcs, n_arcs);
	const vp::CameraCalibration c = calibFrom(calib, d->r.camId, calib_w, calib_h);
	d->r.perspective->setGeometry(f, &c, 1);
	d->r.perspective->geometryCheck(img_w, img_h, d->r.gcSocket->maxBotHeight, resampling_factor);
	d->r.ballRadius = ball_radius; // udpsocket.cpp:123-124
	return d->r.perspective->geometryVersion ? VP_OK : VP_ERR_INVALID;
}

int vph_detector_set_geometry_direct(VphDetector* d, const float calib[11], int img_w, int img_h, float field_scale, float off_x, float off_y, int fw, int fh,
		const int32_t field[5], float ball_radius, float max_robot_radius, const float* lines, int n_lines, const float* arcs, int n_arcs) {
	if(!d || !calib || !field || img_w <= 1 || img_h <= 1 || !(field_scale > 0.f) || fw <= 0 || fh <= 0 || n_lines < 0 || n_arcs < 0 || (n_lines && !lines) || (n_arcs && !arcs))
		return VP_ERR_INVALID;
	vp::Perspective& p = *d->r.perspective;
	p.field = fieldFrom(field, ball_radius, max_robot_radius, lines, n_lines, arcs, n_arcs);
	p.model = vp::CameraModel(calibFrom(calib, d->r.camId, img_w, img_h));
	p.minBlobRadius = std::min({vp::CENTER_BLOB_RADIUS, vp::SIDE_BLOB_RADIUS, ball_radius}); // Perspective.cpp:69-70
	p.maxBlobRadius = std::max({vp::CENTER_BLOB_RADIUS, vp::SIDE_BLOB_RADIUS, ball_radius});
	p.fieldScale = field_scale;
	p.visibleFieldExtent[0] = off_x;
	p.visibleFieldExtent[1] = off_x + field_scale * (float)fw;
	p.visibleFieldExtent[2] = off_y;
	p.visibleFieldExtent[3] = off_y + field_scale * (float)fh;
	p.reprojectedFieldSize = {fw, fh};
	p.geometryVersion = 1;
	d->r.ballRadius = ball_radius;
	return VP_OK;
}

int vph_matches_from_cl(const VphDetector* d, const VpMatch* in, int n, VphMatch* out) {
	if(!d || n < 0 || (n && (!in || !out))) return VP_ERR_INVALID;
	for(int i = 0; i < n; i++) {
		const VpMatch& m = in[i];
		const vp::Vec2f pos = d->r.perspective->flat2field({m.x, m.y});
		out[i] = {pos.x, pos.y, {m.color[0], m.color[1], m.color[2]}, {m.center[0], m.center[1], m.center[2]}, m.circ, m.score};
	}
	return VP_OK;
}

int vph_detector_set_tracked(VphDetector* d, uint32_t cam, const VphTracked* objs, int n) {
	if(!d || n < 0 || (n && !objs)) return VP_ERR_INVALID;
	std::vector<vp::TrackingState>& v = d->r.trackedObjects[cam];
	v.clear();
	for(int i = 0; i < n; i++) {
		const VphTracked& t = objs[i];
		v.push_back({t.id, t.timestamp, t.x, t.y, t.z, t.w, t.vx, t.vy, t.vz, t.vw, t.confidence, t.age});
	}
	return VP_OK;
}

int vph_detector_get_tracked(const VphDetector* d, uint32_t cam, VphTracked* out, int cap) {
	if(!d) return VP_ERR_INVALID;
	const auto it = d->r.trackedObjects.find(cam);
	if(it == d->r.trackedObjects.end()) return 0;
	const int n = (int)it->second.size();
	for(int i = 0; i < std::min(n, cap); i++) {
		const vp::TrackingState& t = it->second[i];
		out[i] = {t.id, t.timestamp, t.x, t.y, t.z, t.w, t.vx, t.vy, t.vz, t.vw, t.confidence, t.age};
	}
	return n;
}

int vph_detector_set_offsets(VphDetector* d, const float* offsets, int n) {
	if(!d || n < 0 || (n && !offsets)) return VP_ERR_INVALID;
	d->r.receivedOffsets.assign(offsets, offsets + n);
	return VP_OK;
}

int vph_detector_colors(VphDetector* d, int32_t colors[7][3], int set) {
	if(!d || !colors) return VP_ERR_INVALID;
	vp::Vec3i* slots[7] = {&d->r.orange, &d->r.field, &d->r.yellow, &d->r.blue, &d->r.green, &d->r.pink, &d->r.fieldLineColor};
	for(int i = 0; i < 7; i++) {
		if(set) *slots[i] = v3(colors[i]);
		else put3(colors[i], *slots[i]);
	}
	return VP_OK;
}

int vph_detect(VphDetector* d, const VphMatch* matches, int n, uint32_t frame_id, double start_time, double img_timestamp) {
	if(!d || n < 0 || (n && !matches)) return VP_ERR_INVALID;
	if(!d->r.perspective->geometryVersion) return VP_ERR_INVALID; // main.cpp:277: no geometry, no detection
	std::vector<vp::Match> ms;
	ms.reserve((size_t)n);
	for(int i = 0; i < n; i++) {
		const VphMatch& m = matches[i];
		ms.push_back({{m.x, m.y}, v3(m.color), v3(m.center), m.circ, m.score});
	}
	d->frame = vp::detectFrame(d->r, ms, frame_id, start_time, img_timestamp);
	return VP_OK;
}

int vph_frame_header(const VphDetector* d, VphFrameHeader* out) {
	if(!d || !out) return VP_ERR_INVALID;
	const vp::DetectionFrame& f = d->frame;
	*out = {f.frame_number, f.camera_id, f.t_capture, f.t_sent, f.t_capture_camera, f.has_t_capture_camera ? 1 : 0, (int32_t)f.balls.size(), (int32_t)f.robots_yellow.size(),
			(int32_t)f.robots_blue.size(), (int32_t)f.t_offsets.size()};
	return VP_OK;
}

int vph_frame_set_t_sent(VphDetector* d, double t_sent) {
	if(!d) return VP_ERR_INVALID;
	d->frame.t_sent = t_sent;
	return VP_OK;
}

int vph_frame_balls(const VphDetector* d, VphBall* out, int cap) {
	if(!d) return VP_ERR_INVALID;
	const int n = (int)d->frame.balls.size();
	for(int i = 0; i < std::min(n, cap); i++) {
		const vp::DetectionBall& b = d->frame.balls[i];
		out[i] = {b.confidence, b.x, b.y, b.pixel_x, b.pixel_y};
	}
	return n;
}

int vph_frame_robots(const VphDetector* d, int blue, VphRobot* out, int cap) {
	if(!d) return VP_ERR_INVALID;
	const std::vector<vp::DetectionRobot>& v = blue ? d->frame.robots_blue : d->frame.robots_yellow;
	const int n = (int)v.size();
	for(int i = 0; i < std::min(n, cap); i++) out[i] = {v[i].confidence, v[i].robot_id, v[i].x, v[i].y, v[i].orientation, v[i].pixel_x, v[i].pixel_y, v[i].height};
	return n;
}

int vph_frame_offsets(const VphDetector* d, float* out, int cap) {
	if(!d) return VP_ERR_INVALID;
	const int n = (int)d->frame.t_offsets.size();
	for(int i = 0; i < std::min(n, cap); i++) out[i] = d->frame.t_offsets[i];
	return n;
}

int vph_track(VphDetector* d) {
	if(!d) return VP_ERR_INVALID;
	vp::detectionTracking(d->r, d->frame);
	return VP_OK;
}

int vph_encode(const VphDetector* d, uint8_t* buf, int cap) {
	if(!d) return VP_ERR_INVALID;
	const std::vector<uint8_t> bytes = vp::encodeWrapperPacket(d->frame);
	if((int)bytes.size() > cap || !buf) return -(int)bytes.size();
	std::memcpy(buf, bytes.data(), bytes.size());
	return (int)bytes.size();
}

int vph_decode(VphDetector* d, const uint8_t* buf, int len, int* source) {
	if(!d || len < 0 || (len && !buf)) return VP_ERR_INVALID;
	vp::DetectionFrame f;
	bool has = false;
	int src = 0;
	if(!vp::decodeWrapperPacket(buf, (size_t)len, f, has, src)) return VP_ERR_INVALID;
	if(source) *source = src;
	if(has) d->frame = f;
	return has ? 1 : 0;
}

} // extern "C"

// One step of n feeds on host threads.  in(i, &amount) yields feed i's records and how many the host loop reads (main.cpp:301); out(i, bytes) stores its packet.
template<typename In, typename Out>
static int detectBatch(VphDetector* const* ds, int n, uint32_t frame_id, double start_time, double img_timestamp, double t_sent, int threads, In in, Out out) {
	for(int i = 0; i < n; i++)
		if(!ds[i] || !ds[i]->r.perspective->geometryVersion) return VP_ERR_INVALID;
	std::atomic<int> next{0};
	auto work = [&]() {
		for(int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
			VphDetector& d = *ds[i];
			int amount = 0;
			const VpMatch* recs = in(i, &amount);
			std::vector<vp::Match> ms = matchesFromCl(*d.r.perspective, recs, amount);
			d.frame = vp::detectFrame(d.r, ms, frame_id, start_time, img_timestamp);
			d.frame.t_sent = t_sent;
			const std::vector<uint8_t> bytes = vp::encodeWrapperPacket(d.frame);
			vp::detectionTracking(d.r, d.frame); // the feed's own packet comes back through the multicast group (udpsocket.cpp:146-154)
			out(i, bytes);
		}
	};
	const int nt = std::max(1, std::min(threads, n));
	std::vector<std::thread> pool;
	for(int t = 1; t < nt; t++) pool.emplace_back(work);
	work();
	for(std::thread& t : pool) t.join();
	return VP_OK;
}

extern "C" {

int vph_detect_batch(VphDetector* const* ds, int n, const VpMatch* cl_matches, const int32_t* counters, int max_matches, uint32_t frame_id, double start_time,
		double img_timestamp, double t_sent, int threads, uint8_t* packets, int packet_cap, int32_t* lengths) {
	if(n < 0 || max_matches < 0 || packet_cap < 0 || (n && (!ds || !cl_matches || !counters || !packets || !lengths))) return VP_ERR_INVALID;
	return detectBatch(ds, n, frame_id, start_time, img_timestamp, t_sent, threads,
			[&](int i, int* amount) {
				*amount = std::max(0, std::min(max_matches, counters[3 * i])); // main.cpp:301
				return cl_matches + (size_t)i * max_matches;
			},
			[&](int i, const std::vector<uint8_t>& bytes) {
				if((int)bytes.size() > packet_cap) {
					lengths[i] = -(int32_t)bytes.size();
					return;
				}
				std::memcpy(packets + (size_t)i * packet_cap, bytes.data(), bytes.size());
				lengths[i] = (int32_t)bytes.size();
			});
}

int vph_detect_batch_slots(VphDetector* const* ds, int n, const uint8_t* slots, size_t slot_bytes, int max_matches, uint32_t frame_id, double start_time, double img_timestamp,
		double t_sent, int threads, uint8_t* packet_block, int packet_slot_bytes) {
	if(n < 0 || max_matches < 0 || packet_slot_bytes <= 4 || slot_bytes < 16 || (n && (!ds || !slots || !packet_block))) return VP_ERR_INVALID;
	for(int i = 0; i < n; i++) { // a slot too small for its frame would silently drop blobs: refuse the step
		int32_t head[4];
		std::memcpy(head, slots + (size_t)i * slot_bytes, sizeof(head));
		if(head[3] < std::max(0, std::min(max_matches, head[0])) || 16 + sizeof(VpMatch) * (size_t)std::max(0, head[3]) > slot_bytes) return VP_ERR_INVALID;
	}
	return detectBatch(ds, n, frame_id, start_time, img_timestamp, t_sent, threads,
			[&](int i, int* amount) {
				const uint8_t* slot = slots + (size_t)i * slot_bytes;
				int32_t head[4];
				std::memcpy(head, slot, sizeof(head));
				*amount = std::max(0, std::min(max_matches, head[0])); // main.cpp:301
				return reinterpret_cast<const VpMatch*>(slot + 16);
			},
			[&](int i, const std::vector<uint8_t>& bytes) {
				uint8_t* dst = packet_block + (size_t)i * packet_slot_bytes;
				const int32_t len = (int)bytes.size() + 4 > packet_slot_bytes ? -(int32_t)bytes.size() : (int32_t)bytes.size();
				std::memcpy(dst, &len, 4);
				if(len > 0) std::memcpy(dst + 4, bytes.data(), bytes.size());
			});
}

static int applyGeometry(VphDetector* d, const vp::FieldSize& field, const std::vector<vp::CameraCalibration>& calibs, int img_w, int img_h, float resampling_factor, float* out_f,
		int32_t* out_i, VpCameraModel* out_model);

int vph_apply_geometry_packet(VphDetector* d, const uint8_t* buf, int len, int img_w, int img_h, float resampling_factor, float* out_f, int32_t* out_i, VpCameraModel* out_model) {
	if(!d || len < 0 || (len && !buf) || img_w <= 1 || img_h <= 1) return VP_ERR_INVALID;
	vp::FieldSize field;
	std::vector<vp::CameraCalibration> calibs;
	bool has = false;
	if(!vp::decodeGeometryPacket(buf, (size_t)len, field, calibs, has)) return VP_ERR_INVALID;
	if(!has) return 0;
	return applyGeometry(d, field, calibs, img_w, img_h, resampling_factor, out_f, out_i, out_model);
}

int vph_udp_apply_geometry(VphDetector* d, int img_w, int img_h, float resampling_factor, int timeout_ms, float* out_f, int32_t* out_i, VpCameraModel* out_model) {
	if(!d || !d->socket || img_w <= 1 || img_h <= 1) return VP_ERR_INVALID;
	if(!d->socket->waitForGeometry(d->socketGeometryApplied, timeout_ms)) return 0;
	vp::FieldSize field;
	std::vector<vp::CameraCalibration> calibs;
	d->socketGeometryApplied = d->socket->getGeometryVersion();
	d->socket->getGeometry(field, calibs);
	return applyGeometry(d, field, calibs, img_w, img_h, resampling_factor, out_f, out_i, out_model);
}

static int applyGeometry(VphDetector* d, const vp::FieldSize& field, const std::vector<vp::CameraCalibration>& calibs, int img_w, int img_h, float resampling_factor, float* out_f,
		int32_t* out_i, VpCameraModel* out_model) {
	vp::Perspective& p = *d->r.perspective;
	const int before = p.geometryVersion;
	p.setGeometry(field, calibs.data(), (int)calibs.size());
	p.geometryCheck(img_w, img_h, d->r.gcSocket->maxBotHeight, resampling_factor);
	bool mine = false;
	for(const vp::CameraCalibration& c : calibs) mine |= c.camera_id == (uint32_t)d->r.camId;
	if(!mine || !p.geometryVersion || p.geometryVersion == before) return 0;
	if(field.ball_radius > 0.f) d->r.ballRadius = field.ball_radius; // udpsocket.cpp:123-124 (has_ball_radius)
	if(out_f) {
		out_f[0] = p.fieldScale;
		for(int i = 0; i < 4; i++) out_f[1 + i] = p.visibleFieldExtent[i];
		out_f[5] = p.minBlobRadius;
		out_f[6] = p.maxBlobRadius;
	}
	if(out_i) {
		out_i[0] = p.reprojectedFieldSize.x;
		out_i[1] = p.reprojectedFieldSize.y;
		out_i[2] = p.gradOffset();
		out_i[3] = p.circleRadius();
		out_i[4] = p.listRadius();
		out_i[5] = p.geometryVersion;
	}
	if(out_model) *out_model = p.getCLCameraModel();
	return 1;
}

int vph_udp_open(VphDetector* d, const char* ip, int port) {
	if(!d || !ip || port <= 0 || port > 65535) return VP_ERR_INVALID;
	d->socket = std::make_unique<vp::VisionSocket>(ip, (uint16_t)port, d->r.camId, (float)d->r.gcSocket->defaultBotHeight);
	if(!d->socket->ok()) {
		d->socket.reset();
		return VP_ERR_INVALID;
	}
	return VP_OK;
}

int vph_udp_send(VphDetector* d) {
	if(!d || !d->socket) return VP_ERR_INVALID;
	d->socket->send(d->frame);
	return VP_OK;
}

int vph_udp_sync(VphDetector* d, int at_least, int timeout_ms) {
	if(!d || !d->socket) return VP_ERR_INVALID;
	if(!d->socket->waitForPackets(at_least, timeout_ms)) return -1;
	d->r.trackedObjects = d->socket->getTrackedObjects();
	return d->socket->receivedPackets();
}

int vph_udp_update_time(VphDetector* d, double* now) {
	if(!d || !d->socket) return VP_ERR_INVALID;
	d->r.receivedOffsets = d->socket->getReceivedOffsets(); // main.cpp:361-362
	d->socket->updateTime();                                 // main.cpp:371
	if(now) *now = d->socket->getRealTime();
	return VP_OK;
}

int vph_gc_open(VphDetector* d, const char* ip, int port, const char* const* names, const double* heights_mm, int n) {
	if(!d || !ip || port <= 0 || port > 65535 || n <= 0 || !names || !heights_mm) return VP_ERR_INVALID;
	std::map<std::string, double> table;
	for(int i = 0; i < n; i++) {
		if(!names[i]) return VP_ERR_INVALID;
		table[names[i]] = heights_mm[i];
	}
	d->gcSocket = std::make_unique<vp::GCSocket>(ip, (uint16_t)port, table);
	if(!d->gcSocket->ok()) {
		d->gcSocket.reset();
		return VP_ERR_INVALID;
	}
	d->gcSocket->heights(d->r.gcSocket->maxBotHeight, d->r.gcSocket->defaultBotHeight, d->r.gcSocket->yellowBotHeight, d->r.gcSocket->blueBotHeight);
	return VP_OK;
}

int vph_gc_sync(VphDetector* d, int at_least, int timeout_ms, double heights_out[4]) {
	if(!d || !d->gcSocket) return VP_ERR_INVALID;
	const bool ok = d->gcSocket->waitForPackets(at_least, timeout_ms);
	d->gcSocket->heights(d->r.gcSocket->maxBotHeight, d->r.gcSocket->defaultBotHeight, d->r.gcSocket->yellowBotHeight, d->r.gcSocket->blueBotHeight);
	if(heights_out) {
		heights_out[0] = d->r.gcSocket->maxBotHeight;
		heights_out[1] = d->r.gcSocket->defaultBotHeight;
		heights_out[2] = d->r.gcSocket->yellowBotHeight;
		heights_out[3] = d->r.gcSocket->blueBotHeight;
	}
	return ok ? d->gcSocket->receivedPackets() : -1;
}

int vph_gc_close(VphDetector* d) {
	if(!d) return VP_ERR_INVALID;
	d->gcSocket.reset();
	return VP_OK;
}

int vph_udp_close(VphDetector* d) {
	if(!d) return VP_ERR_INVALID;
	d->socket.reset();
	return VP_OK;
}

float vph_atan2_fast(float y, float x) { return vp::atan2_fast(y, x); }

int vph_kmeans(const int32_t contrast[3], const int32_t* values, int n, int32_t c1[3], int32_t c2[3]) {
	std::vector<vp::Vec3i> vs;
	for(int i = 0; i < n; i++) vs.push_back(v3(values + 3 * i));
	vp::Vec3i a = v3(c1), b = v3(c2);
	const bool ok = vp::kMeans(v3(contrast), vs, a, b);
	put3(c1, a);
	put3(c2, b);
	return ok ? 1 : 0;
}

int vph_kdtree_range(const float* pts, int n, float px, float py, float radius, int32_t* out_idx, int cap) {
	if(n <= 0) return 0;
	std::vector<vp::Match> ms((size_t)n);
	for(int i = 0; i < n; i++) ms[i].pos = {pts[2 * i], pts[2 * i + 1]};
	vp::KDTree tree(&ms[0]);
	for(int i = 1; i < n; i++) tree.insert(&ms[i]);
	std::vector<vp::Match*> found;
	tree.rangeSearch(found, {px, py}, radius);
	for(int i = 0; i < std::min((int)found.size(), cap); i++) out_idx[i] = (int32_t)(found[i] - &ms[0]);
	return (int)found.size();
}

} // extern "C"
