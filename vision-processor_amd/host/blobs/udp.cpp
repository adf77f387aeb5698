// The datagram socket of src/udpsocket.cpp:26-113 and the detection half of VisionSocket (:146-163, 204-256), POSIX only.
// Networking is I/O around the path, kept to what carries the wire format of wire.cpp between camera processes.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#include "detection.h"

namespace vp {

struct UDPSocket::Impl {
	std::thread receiver;
	std::atomic<bool> closing{false};
	bool started = false;
};

UDPSocket::UDPSocket(const std::string& ip, uint16_t port): impl(new Impl()) {
	static_assert(sizeof(sockaddr_in) <= sizeof(addr_), "addr_ too small");
	sockaddr_in* in = reinterpret_cast<sockaddr_in*>(addr_);
	in->sin_family = AF_INET;
	in->sin_port = htons(port);
	if(!inet_aton(ip.c_str(), &in->sin_addr)) {
		fprintf(stderr, "[WARN] vision socket: '%s' is not an IPv4 address\n", ip.c_str());
		return;
	}
	socket_ = socket(AF_INET, SOCK_DGRAM, 0);
	if(socket_ < 0) {
		fprintf(stderr, "[WARN] vision socket: socket() failed\n");
		return;
	}
	const int yes = 1, ttl = 32;
	if(setsockopt(socket_, SOL_SOCKET, SO_REUSEADDR, &yes, sizeof(yes)) < 0) fprintf(stderr, "[WARN] vision socket: SO_REUSEADDR refused\n");
	if(setsockopt(socket_, SOL_SOCKET, SO_BROADCAST, &yes, sizeof(yes)) < 0) fprintf(stderr, "[WARN] vision socket: SO_BROADCAST refused\n");
	if(setsockopt(socket_, IPPROTO_IP, IP_MULTICAST_TTL, &ttl, sizeof(ttl)) < 0) fprintf(stderr, "[WARN] vision socket: multicast TTL refused\n");
	if(bind(socket_, reinterpret_cast<const sockaddr*>(in), sizeof(*in))) fprintf(stderr, "[WARN] vision socket: bind(%s:%u) failed (another receiver owns the port?)\n", ip.c_str(), port);
	if(IN_MULTICAST(ntohl(in->sin_addr.s_addr))) {
		ip_mreq mreq = {};
		mreq.imr_multiaddr = in->sin_addr;
		mreq.imr_interface.s_addr = htonl(INADDR_ANY);
		if(setsockopt(socket_, IPPROTO_IP, IP_ADD_MEMBERSHIP, &mreq, sizeof(mreq)) < 0) fprintf(stderr, "[WARN] vision socket: joining group %s failed\n", ip.c_str());
	}
}

void UDPSocket::start() {
	if(socket_ < 0 || impl->started) return;
	impl->started = true;
	impl->receiver = std::thread(&UDPSocket::run, this);
}

void UDPSocket::stop() {
	if(!impl->started) return;
	impl->closing = true;
	shutdown(socket_, SHUT_RD);
	// a datagram to ourselves wakes a receiver that shutdown() did not (unconnected UDP sockets on some kernels)
	sendto(socket_, "", 0, 0, reinterpret_cast<const sockaddr*>(addr_), sizeof(sockaddr_in));
	impl->receiver.join();
	impl->started = false;
}

UDPSocket::~UDPSocket() {
	stop();
	if(socket_ >= 0) ::close(socket_);
	delete impl;
}

bool UDPSocket::send(const uint8_t* data, size_t size) {
	if(socket_ < 0) return false;
	if(sendto(socket_, data, size, 0, reinterpret_cast<const sockaddr*>(addr_), sizeof(sockaddr_in)) < 0) {
		fprintf(stderr, "[WARN] vision socket: sendto failed: %s\n", strerror(errno));
		return false;
	}
	return true;
}

void UDPSocket::run() {
	std::vector<uint8_t> msgbuf(65535);
	while(true) {
		const ssize_t bytesRead = read(socket_, msgbuf.data(), msgbuf.size());
		if(impl->closing) return;
		if(bytesRead < 0) {
			fprintf(stderr, "[WARN] vision socket: read failed: %s\n", strerror(errno));
			return;
		}
		parse(msgbuf.data(), (int)bytesRead);
	}
}

struct GCSocket::State {
	std::mutex m;
	std::condition_variable cv;
	std::map<std::string, double> botHeights;
	double maxBotHeight = 0, defaultBotHeight = 0, yellowBotHeight = 0, blueBotHeight = 0;
	int received = 0;
};

GCSocket::GCSocket(const std::string& ip, uint16_t port, const std::map<std::string, double>& botHeights): UDPSocket(ip, port), state(new State()) {
	state->botHeights = botHeights;
	for(const auto& entry : botHeights) { // udpsocket.cpp:304-314
		if(entry.second > state->maxBotHeight) state->maxBotHeight = entry.second;
		state->defaultBotHeight += entry.second;
	}
	if(!botHeights.empty()) state->defaultBotHeight /= (double)botHeights.size();
	state->yellowBotHeight = state->blueBotHeight = state->defaultBotHeight;
	start();
}

GCSocket::~GCSocket() {
	stop();
	delete state;
}

void GCSocket::parse(const uint8_t* data, int length) {
	std::string yellow, blue;
	if(!decodeRefereeTeams(data, (size_t)length, yellow, blue)) return;
	std::lock_guard<std::mutex> lock(state->m);
	const auto y = state->botHeights.find(yellow), b = state->botHeights.find(blue);
	if(y != state->botHeights.end() && y->second != state->yellowBotHeight) state->yellowBotHeight = y->second;
	if(b != state->botHeights.end() && b->second != state->blueBotHeight) state->blueBotHeight = b->second;
	state->received++;
	state->cv.notify_all();
}

void GCSocket::heights(double& maxBotHeight, double& defaultBotHeight, double& yellowBotHeight, double& blueBotHeight) {
	std::lock_guard<std::mutex> lock(state->m);
	maxBotHeight = state->maxBotHeight;
	defaultBotHeight = state->defaultBotHeight;
	yellowBotHeight = state->yellowBotHeight;
	blueBotHeight = state->blueBotHeight;
}

int GCSocket::receivedPackets() {
	std::lock_guard<std::mutex> lock(state->m);
	return state->received;
}

bool GCSocket::waitForPackets(int atLeast, int timeoutMs) {
	std::unique_lock<std::mutex> lock(state->m);
	return state->cv.wait_for(lock, std::chrono::milliseconds(timeoutMs), [&] { return state->received >= atLeast; });
}

namespace {
bool sameBits(float a, float b) { return std::memcmp(&a, &b, sizeof(float)) == 0; }
bool sameGeometry(const FieldSize& fa, const std::vector<CameraCalibration>& ca, const FieldSize& fb, const std::vector<CameraCalibration>& cb) {
	if(fa.field_length != fb.field_length || fa.field_width != fb.field_width || fa.boundary_width != fb.boundary_width ||
			fa.has_boundary_width_goal_line != fb.has_boundary_width_goal_line || fa.boundary_width_goal_line != fb.boundary_width_goal_line || fa.line_thickness != fb.line_thickness ||
			!sameBits(fa.ball_radius, fb.ball_radius) || !sameBits(fa.max_robot_radius, fb.max_robot_radius) || fa.field_lines.size() != fb.field_lines.size() ||
			fa.field_arcs.size() != fb.field_arcs.size() || ca.size() != cb.size())
		return false;
	for(size_t i = 0; i < fa.field_lines.size(); i++)
		if(std::memcmp(&fa.field_lines[i], &fb.field_lines[i], sizeof(FieldLine)) != 0) return false;
	for(size_t i = 0; i < fa.field_arcs.size(); i++)
		if(std::memcmp(&fa.field_arcs[i], &fb.field_arcs[i], sizeof(FieldArc)) != 0) return false;
	for(size_t i = 0; i < ca.size(); i++) {
		const CameraCalibration &a = ca[i], &b = cb[i];
		const float av[14] = {a.focal_length, a.principal_point_x, a.principal_point_y, a.distortion, a.q0, a.q1, a.q2, a.q3, a.tx, a.ty, a.tz, a.derived_camera_world_tx,
				a.derived_camera_world_ty, a.derived_camera_world_tz};
		const float bv[14] = {b.focal_length, b.principal_point_x, b.principal_point_y, b.distortion, b.q0, b.q1, b.q2, b.q3, b.tx, b.ty, b.tz, b.derived_camera_world_tx,
				b.derived_camera_world_ty, b.derived_camera_world_tz};
		if(a.camera_id != b.camera_id || a.has_derived != b.has_derived || a.pixel_image_width != b.pixel_image_width || a.pixel_image_height != b.pixel_image_height ||
				std::memcmp(av, bv, sizeof(av)) != 0)
			return false;
	}
	return true;
}
} // namespace

struct VisionSocket::State {
	std::mutex m;
	std::condition_variable cv;
	BlobResources tracker; // only trackedObjects and ballRadius are used (detectionTracking's working set)
	int received = 0;
	// the last geometry heard on the group (udpsocket.cpp:156-162) and how often it changed (getGeometryVersion, :113-130)
	FieldSize field;
	std::vector<CameraCalibration> calibs;
	int geometryVersion = 0;
	// clock offsets against the other cameras (udpsocket.h:109-112): [sender id]
	std::vector<float> sentOffsets;     // local.t_sent - other.time, as the other camera reported it back in its t_offsets
	std::vector<float> receivedOffsets; // other.t_sent - local.time
	double realTimeOffset = 0.0;        // cameradriver.cpp:24: what updateTime() steers
};

VisionSocket::VisionSocket(const std::string& ip, uint16_t port, int camId, float defaultBotHeight): UDPSocket(ip, port), state(new State()), camId(camId) {
	state->tracker.gcSocket->defaultBotHeight = defaultBotHeight;
	start();
}

VisionSocket::~VisionSocket() {
	stop();
	delete state;
}

void VisionSocket::send(const DetectionFrame& detection) {
	const std::vector<uint8_t> bytes = encodeWrapperPacket(detection);
	UDPSocket::send(bytes.data(), bytes.size());
}

void VisionSocket::parse(const uint8_t* data, int length) {
	DetectionFrame frame;
	bool hasDetection = false, hasGeometry = false;
	int source = 0;
	if(!decodeWrapperPacket(data, (size_t)length, frame, hasDetection, source)) return; // ParseFromArray failure: dropped (udpsocket.cpp:147-148)
	FieldSize field;
	std::vector<CameraCalibration> calibs;
	if(!decodeGeometryPacket(data, (size_t)length, field, calibs, hasGeometry)) return;
	std::lock_guard<std::mutex> lock(state->m);
	if(hasDetection) {
		{ // timeSynchronization (udpsocket.cpp:284-302)
			const double localTime = getRealTimeLocked();
			const unsigned senderId = frame.camera_id;
			if(senderId < 4096) { // a camera id is a small index; a garbage id must not size the vectors
				while(state->receivedOffsets.size() <= senderId) {
					state->receivedOffsets.push_back(0.0f);
					state->sentOffsets.push_back(0.0f);
				}
				state->receivedOffsets[senderId] = (float)(frame.t_sent - localTime);
				if(frame.t_offsets.size() > (size_t)camId) state->sentOffsets[senderId] = frame.t_offsets[(size_t)camId];
			}
		}
		detectionTracking(state->tracker, frame);
		state->received++;
	}
	if(hasGeometry && !sameGeometry(field, calibs, state->field, state->calibs)) { // MessageDifferencer::Equals on the fields this path reads (:157)
		state->field = field;
		state->calibs = calibs;
		state->geometryVersion++;
		if(field.ball_radius > 0.f) state->tracker.ballRadius = field.ball_radius; // :123-124
	}
	state->cv.notify_all();
}

double VisionSocket::getRealTimeLocked() const { // cameradriver.cpp:25-27
	return (double)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::system_clock::now().time_since_epoch()).count() / 1e6 + state->realTimeOffset;
}

double VisionSocket::getRealTime() {
	std::lock_guard<std::mutex> lock(state->m);
	return getRealTimeLocked();
}

std::vector<float> VisionSocket::getReceivedOffsets() { // udpsocket.cpp:139-144
	std::lock_guard<std::mutex> lock(state->m);
	return state->receivedOffsets;
}

double VisionSocket::updateTime() { // udpsocket.cpp:259-282: half the mean asymmetry of the offsets, backward jumps beyond 10 ms refused
	std::lock_guard<std::mutex> lock(state->m);
	double offset = 0.0;
	const int cams = (int)state->receivedOffsets.size();
	for(int cam = 0; cam < cams; cam++) {
		if(cam == camId) continue; // not against itself
		offset += state->receivedOffsets[cam] - state->sentOffsets[cam];
	}
	if(cams == 0) return state->realTimeOffset;
	offset /= 2 * cams;
	if(offset < -0.010) {
		fprintf(stderr, "[WARN] vision socket: backward clock step of %f s refused\n", offset);
		return state->realTimeOffset;
	}
	state->realTimeOffset += offset;
	return state->realTimeOffset;
}

int VisionSocket::getGeometryVersion() {
	std::lock_guard<std::mutex> lock(state->m);
	return state->geometryVersion;
}

void VisionSocket::getGeometry(FieldSize& field, std::vector<CameraCalibration>& calibs) {
	std::lock_guard<std::mutex> lock(state->m);
	field = state->field;
	calibs = state->calibs;
}

bool VisionSocket::waitForGeometry(int newerThan, int timeoutMs) {
	std::unique_lock<std::mutex> lock(state->m);
	return state->cv.wait_for(lock, std::chrono::milliseconds(timeoutMs), [&] { return state->geometryVersion > newerThan; });
}

std::map<unsigned int, std::vector<TrackingState>> VisionSocket::getTrackedObjects() {
	std::lock_guard<std::mutex> lock(state->m);
	return state->tracker.trackedObjects;
}

int VisionSocket::receivedPackets() {
	std::lock_guard<std::mutex> lock(state->m);
	return state->received;
}

bool VisionSocket::waitForPackets(int atLeast, int timeoutMs) {
	std::unique_lock<std::mutex> lock(state->m);
	return state->cv.wait_for(lock, std::chrono::milliseconds(timeoutMs), [&] { return state->received >= atLeast; });
}

} // namespace vp
