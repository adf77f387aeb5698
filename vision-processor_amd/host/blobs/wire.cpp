// proto2 wire format of SSL_WrapperPacket { detection = 1; source = 3 } (proto/ssl_vision_wrapper.proto:15-19,
// proto/ssl_vision_detection.proto:3-35), SURVEY.md 8(f)-4.  Written against the protobuf encoding specification (varint,
// 64-bit, length-delimited and 32-bit wire types; tag = field << 3 | type); no protobuf library is linked.
//
// Encoder contract: the bytes protobuf's own serializer produces for a message carrying exactly the fields the reference sets
// (main.cpp:345-362, hypothesis.cpp:69-80,140-153) -- known fields in field-number order, `repeated float t_offsets` unpacked
// (proto2 default).  tests/test_wire_format.py compares against google.protobuf (python) built from the same schema.
#include <cstring>
#include <string>

#include "detection.h"

namespace vp {

namespace {

enum WireType { VARINT = 0, I64 = 1, LEN = 2, I32 = 5 };
const int SSL_SOURCE_VISION_PROCESSOR = 3; // ssl_vision_wrapper.proto:9

struct Writer {
	std::vector<uint8_t> out;
	void varint(uint64_t v) {
		while(v >= 0x80) {
			out.push_back((uint8_t)(v | 0x80));
			v >>= 7;
		}
		out.push_back((uint8_t)v);
	}
	void tag(int field, WireType t) { varint(((uint64_t)field << 3) | (uint64_t)t); }
	void f32(int field, float v) {
		tag(field, I32);
		uint8_t b[4];
		std::memcpy(b, &v, 4); // little-endian host (x86-64), as the wire format requires
		out.insert(out.end(), b, b + 4);
	}
	void f64(int field, double v) {
		tag(field, I64);
		uint8_t b[8];
		std::memcpy(b, &v, 8);
		out.insert(out.end(), b, b + 8);
	}
	void u32(int field, uint32_t v) {
		tag(field, VARINT);
		varint(v);
	}
	void bytes(int field, const std::vector<uint8_t>& payload) {
		tag(field, LEN);
		varint(payload.size());
		out.insert(out.end(), payload.begin(), payload.end());
	}
};

std::vector<uint8_t> encodeBall(const DetectionBall& b) { // SSL_DetectionBall: area (2) and z (5) are never set
	Writer w;
	w.f32(1, b.confidence);
	w.f32(3, b.x);
	w.f32(4, b.y);
	w.f32(6, b.pixel_x);
	w.f32(7, b.pixel_y);
	return w.out;
}

std::vector<uint8_t> encodeRobot(const DetectionRobot& r) { // SSL_DetectionRobot
	Writer w;
	w.f32(1, r.confidence);
	w.u32(2, r.robot_id);
	w.f32(3, r.x);
	w.f32(4, r.y);
	w.f32(5, r.orientation);
	w.f32(6, r.pixel_x);
	w.f32(7, r.pixel_y);
	w.f32(8, r.height);
	return w.out;
}

std::vector<uint8_t> encodeFrame(const DetectionFrame& d) { // SSL_DetectionFrame, ascending field numbers
	Writer w;
	w.u32(1, d.frame_number);
	w.f64(2, d.t_capture);
	w.f64(3, d.t_sent);
	w.u32(4, d.camera_id);
	for(const DetectionBall& b : d.balls) w.bytes(5, encodeBall(b));
	for(const DetectionRobot& r : d.robots_yellow) w.bytes(6, encodeRobot(r));
	for(const DetectionRobot& r : d.robots_blue) w.bytes(7, encodeRobot(r));
	if(d.has_t_capture_camera) w.f64(8, d.t_capture_camera);
	for(const float o : d.t_offsets) w.f32(9, o);
	return w.out;
}

struct Reader {
	const uint8_t* p;
	const uint8_t* end;
	bool ok = true;
	bool done() const { return p >= end; }
	uint64_t varint() {
		uint64_t v = 0;
		for(int shift = 0; shift < 64; shift += 7) {
			if(p >= end) break;
			const uint8_t b = *p++;
			v |= (uint64_t)(b & 0x7F) << shift;
			if(!(b & 0x80)) return v;
		}
		ok = false;
		return 0;
	}
	float f32() {
		float v = 0;
		if(end - p < 4) {
			ok = false;
			return v;
		}
		std::memcpy(&v, p, 4);
		p += 4;
		return v;
	}
	double f64() {
		double v = 0;
		if(end - p < 8) {
			ok = false;
			return v;
		}
		std::memcpy(&v, p, 8);
		p += 8;
		return v;
	}
	Reader sub() {
		const uint64_t n = varint();
		if(!ok || n > (uint64_t)(end - p)) {
			ok = false;
			return {p, p};
		}
		Reader r{p, p + n};
		p += n;
		return r;
	}
	void skip(int type) { // unknown fields are legal and ignored
		if(type == VARINT) varint();
		else if(type == I64) f64();
		else if(type == I32) f32();
		else if(type == LEN) sub();
		else ok = false; // groups are not used by these schemas
	}
};

bool decodeBall(Reader r, DetectionBall& b) {
	b = {};
	while(r.ok && !r.done()) {
		const uint64_t t = r.varint();
		const int field = (int)(t >> 3), type = (int)(t & 7);
		if(type == I32 && field == 1) b.confidence = r.f32();
		else if(type == I32 && field == 3) b.x = r.f32();
		else if(type == I32 && field == 4) b.y = r.f32();
		else if(type == I32 && field == 6) b.pixel_x = r.f32();
		else if(type == I32 && field == 7) b.pixel_y = r.f32();
		else r.skip(type);
	}
	return r.ok;
}

bool decodeRobot(Reader r, DetectionRobot& b) {
	b = {};
	while(r.ok && !r.done()) {
		const uint64_t t = r.varint();
		const int field = (int)(t >> 3), type = (int)(t & 7);
		if(type == I32 && field == 1) b.confidence = r.f32();
		else if(type == VARINT && field == 2) b.robot_id = (uint32_t)r.varint();
		else if(type == I32 && field == 3) b.x = r.f32();
		else if(type == I32 && field == 4) b.y = r.f32();
		else if(type == I32 && field == 5) b.orientation = r.f32();
		else if(type == I32 && field == 6) b.pixel_x = r.f32();
		else if(type == I32 && field == 7) b.pixel_y = r.f32();
		else if(type == I32 && field == 8) b.height = r.f32();
		else r.skip(type);
	}
	return r.ok;
}

bool decodeFrame(Reader r, DetectionFrame& d) {
	d = DetectionFrame();
	while(r.ok && !r.done()) {
		const uint64_t t = r.varint();
		const int field = (int)(t >> 3), type = (int)(t & 7);
		if(type == VARINT && field == 1) d.frame_number = (uint32_t)r.varint();
		else if(type == I64 && field == 2) d.t_capture = r.f64();
		else if(type == I64 && field == 3) d.t_sent = r.f64();
		else if(type == VARINT && field == 4) d.camera_id = (uint32_t)r.varint();
		else if(type == LEN && field == 5) {
			DetectionBall b;
			if(!decodeBall(r.sub(), b)) return false;
			d.balls.push_back(b);
		} else if(type == LEN && (field == 6 || field == 7)) {
			DetectionRobot b;
			if(!decodeRobot(r.sub(), b)) return false;
			(field == 6 ? d.robots_yellow : d.robots_blue).push_back(b);
		} else if(type == I64 && field == 8) {
			d.has_t_capture_camera = true;
			d.t_capture_camera = r.f64();
		} else if(type == I32 && field == 9) d.t_offsets.push_back(r.f32());
		else if(type == LEN && field == 9) { // a packed encoding is accepted on input, as protobuf parsers do
			Reader s = r.sub();
			while(s.ok && !s.done()) d.t_offsets.push_back(s.f32());
			if(!s.ok) return false;
		} else r.skip(type);
	}
	return r.ok;
}

// ---- SSL_GeometryData (proto/ssl_vision_geometry.proto:3-99): the fields Perspective / CameraModel / the hypothesis stage read ----
bool decodeVector2f(Reader r, Vec2f& v) {
	v = {};
	while(r.ok && !r.done()) {
		const uint64_t t = r.varint();
		const int field = (int)(t >> 3), type = (int)(t & 7);
		if(type == I32 && field == 1) v.x = r.f32();
		else if(type == I32 && field == 2) v.y = r.f32();
		else r.skip(type);
	}
	return r.ok;
}

bool decodeFieldSize(Reader r, FieldSize& f) {
	f = FieldSize();
	f.ball_radius = 0.f; // proto2 optionals without a default read as zero when absent (Perspective.cpp:69-70 takes them as they come)
	f.max_robot_radius = 0.f;
	while(r.ok && !r.done()) {
		const uint64_t t = r.varint();
		const int field = (int)(t >> 3), type = (int)(t & 7);
		if(type == VARINT && field == 1) f.field_length = (int32_t)r.varint();
		else if(type == VARINT && field == 2) f.field_width = (int32_t)r.varint();
		else if(type == VARINT && field == 5) f.boundary_width = (int32_t)r.varint();
		else if(type == VARINT && field == 16) {
			f.has_boundary_width_goal_line = true;
			f.boundary_width_goal_line = (int32_t)r.varint();
		} else if(type == VARINT && field == 11) f.line_thickness = (int32_t)r.varint();
		else if(type == I32 && field == 14) f.ball_radius = r.f32();
		else if(type == I32 && field == 15) f.max_robot_radius = r.f32();
		else if(type == LEN && field == 6) { // SSL_FieldLineSegment: p1 = 2, p2 = 3
			Reader m = r.sub();
			FieldLine line;
			while(m.ok && !m.done()) {
				const uint64_t mt = m.varint();
				const int mf = (int)(mt >> 3), my = (int)(mt & 7);
				if(my == LEN && mf == 2) {
					if(!decodeVector2f(m.sub(), line.p1)) return false;
				} else if(my == LEN && mf == 3) {
					if(!decodeVector2f(m.sub(), line.p2)) return false;
				} else m.skip(my);
			}
			if(!m.ok) return false;
			f.field_lines.push_back(line);
		} else if(type == LEN && field == 7) { // SSL_FieldCircularArc: center = 2, radius = 3, a1 = 4, a2 = 5
			Reader m = r.sub();
			FieldArc arc;
			while(m.ok && !m.done()) {
				const uint64_t mt = m.varint();
				const int mf = (int)(mt >> 3), my = (int)(mt & 7);
				if(my == LEN && mf == 2) {
					if(!decodeVector2f(m.sub(), arc.center)) return false;
				} else if(my == I32 && mf == 3) arc.radius = m.f32();
				else if(my == I32 && mf == 4) arc.a1 = m.f32();
				else if(my == I32 && mf == 5) arc.a2 = m.f32();
				else m.skip(my);
			}
			if(!m.ok) return false;
			f.field_arcs.push_back(arc);
		} else r.skip(type); // goal sizes, penalty area, centre circle radius ...: not read on this path
	}
	return r.ok;
}

bool decodeCalibration(Reader r, CameraCalibration& c) {
	c = CameraCalibration();
	int derived = 0;
	while(r.ok && !r.done()) {
		const uint64_t t = r.varint();
		const int field = (int)(t >> 3), type = (int)(t & 7);
		if(type == VARINT && field == 1) c.camera_id = (uint32_t)r.varint();
		else if(type == VARINT && field == 16) c.pixel_image_width = (uint32_t)r.varint();
		else if(type == VARINT && field == 17) c.pixel_image_height = (uint32_t)r.varint();
		else if(type == I32 && field >= 2 && field <= 15) {
			const float v = r.f32();
			float* slots[14] = {&c.focal_length, &c.principal_point_x, &c.principal_point_y, &c.distortion, &c.q0, &c.q1, &c.q2, &c.q3, &c.tx, &c.ty, &c.tz,
					&c.derived_camera_world_tx, &c.derived_camera_world_ty, &c.derived_camera_world_tz};
			*slots[field - 2] = v;
			if(field >= 13) derived |= 1 << (field - 13);
		} else r.skip(type);
	}
	c.has_derived = derived == 7; // Perspective.cpp:45: all three derived_camera_world_* present
	return r.ok;
}

} // namespace

bool decodeGeometryPacket(const uint8_t* data, size_t size, FieldSize& field, std::vector<CameraCalibration>& calibs, bool& hasGeometry) {
	Reader r{data, data + size};
	hasGeometry = false;
	calibs.clear();
	while(r.ok && !r.done()) {
		const uint64_t t = r.varint();
		const int fieldNo = (int)(t >> 3), type = (int)(t & 7);
		if(type == LEN && fieldNo == 2) { // SSL_WrapperPacket.geometry
			Reader g = r.sub();
			calibs.clear();
			while(g.ok && !g.done()) {
				const uint64_t gt = g.varint();
				const int gf = (int)(gt >> 3), gy = (int)(gt & 7);
				if(gy == LEN && gf == 1) {
					if(!decodeFieldSize(g.sub(), field)) return false;
				} else if(gy == LEN && gf == 2) {
					CameraCalibration c;
					if(!decodeCalibration(g.sub(), c)) return false;
					calibs.push_back(c);
				} else g.skip(gy); // models
			}
			if(!g.ok) return false;
			hasGeometry = true;
		} else r.skip(type);
	}
	return r.ok;
}

// Referee (proto/ssl_gc_referee_message.proto:119-157): the two team names, all GCSocket::parse reads (udpsocket.cpp:317-330)
bool decodeRefereeTeams(const uint8_t* data, size_t size, std::string& yellow, std::string& blue) {
	Reader r{data, data + size};
	yellow.clear();
	blue.clear();
	while(r.ok && !r.done()) {
		const uint64_t t = r.varint();
		const int field = (int)(t >> 3), type = (int)(t & 7);
		if(type == LEN && (field == 7 || field == 8)) { // TeamInfo yellow = 7, blue = 8; name = 1
			Reader team = r.sub();
			std::string& name = field == 7 ? yellow : blue;
			while(team.ok && !team.done()) {
				const uint64_t tt = team.varint();
				const int tf = (int)(tt >> 3), ty = (int)(tt & 7);
				if(ty == LEN && tf == 1) {
					Reader n = team.sub();
					if(!n.ok) return false;
					name.assign(reinterpret_cast<const char*>(n.p), (size_t)(n.end - n.p));
				} else team.skip(ty);
			}
			if(!team.ok) return false;
		} else r.skip(type);
	}
	return r.ok;
}

std::vector<uint8_t> encodeWrapperPacket(const DetectionFrame& detection) {
	Writer w;
	w.bytes(1, encodeFrame(detection));
	w.u32(3, SSL_SOURCE_VISION_PROCESSOR);
	return w.out;
}

bool decodeWrapperPacket(const uint8_t* data, size_t size, DetectionFrame& detection, bool& hasDetection, int& source) {
	Reader r{data, data + size};
	hasDetection = false;
	source = 0; // SSL_SOURCE_UNKNOWN
	while(r.ok && !r.done()) {
		const uint64_t t = r.varint();
		const int field = (int)(t >> 3), type = (int)(t & 7);
		if(type == LEN && field == 1) {
			// a repeated occurrence of a singular message field merges; the producers of this schema never do that, so the last one wins
			if(!decodeFrame(r.sub(), detection)) return false;
			hasDetection = true;
		} else if(type == VARINT && field == 3) source = (int)r.varint();
		else r.skip(type); // geometry (2) and anything newer
	}
	return r.ok;
}

} // namespace vp
