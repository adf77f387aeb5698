// Blob list -> SSL_DetectionFrame: the CPU stage that follows the GPU hot path in the reference.
//
// Restates src/blobs/{match.h,kdtree.*,kmeans.*,hypothesis.*,colorupdate.*}, the hypothesis generators and filters of
// src/main.cpp:43-241, the per-frame sequence main.cpp:319-359, and the detection tracker of src/udpsocket.cpp:165-256,
// Eigen-free and protobuf-free: Eigen vectors are the PODs of ../CameraModel.h, the protobuf messages are plain structs,
// the polymorphic unique_ptr lists are value vectors (same element order, same erase semantics).
//
// PARITY UNPINNED: the reference host code cannot be built in this image (Eigen, protobuf, yaml-cpp absent) and ships no
// test or fixture for this stage; the reference is also compiled with -ffast-math -march=native (CMakeLists.txt:28), so its
// own float results are compiler-dependent.  This restatement is strict binary32, one rounding per source operation
// (-ffp-contract=off), evaluation order as written in the cited lines.  tests/ hold it to an independent numpy/libm
// restatement (oracle/detect_oracle.py), which pins the two against each other and nothing more.
#pragma once

#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../CameraModel.h"

namespace vp {

struct Vec3i {
	int x = 0, y = 0, z = 0;
	bool operator==(const Vec3i& o) const { return x == o.x && y == o.y && z == o.z; }
	bool operator!=(const Vec3i& o) const { return !(*this == o); }
};
inline int sqDiff(const Vec3i& a, const Vec3i& b) { // (a - b).squaredNorm()
	const int dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
	return dx * dx + dy * dy + dz * dz;
}

struct Match { // blobs/match.h:22-30 -- pos already in field coordinates (Perspective::flat2field, main.cpp:306)
	Vec2f pos;
	Vec3i color;
	Vec3i center;
	float circ = 0.f;
	float score = 0.f;
};

// blobs/kdtree.{h,cpp}: unbalanced 2-d tree in insertion order, split axis alternating per level.  Stored as an index
// arena instead of linked unique_ptr nodes; insert() and rangeSearch() visit nodes in the reference's order (node, left
// subtree, right subtree), which fixes the order of the result vector.
class KDTree {
public:
	KDTree() = default;
	explicit KDTree(Match* root) { nodes.push_back({root, -1, -1, 0}); }
	void insert(Match* m);
	void rangeSearch(std::vector<Match*>& values, const Vec2f& point, float radius) const;
	int getSize() const { return (int)nodes.size(); }

private:
	struct Node {
		Match* data;
		int left, right, dim;
	};
	std::vector<Node> nodes;
	void search(int n, std::vector<Match*>& values, const Vec2f& point, float radius) const;
};

bool kMeans(const Vec3i& contrast, const Vec3i* values, size_t n, Vec3i& c1, Vec3i& c2);      // blobs/kmeans.cpp:19-90
inline bool kMeans(const Vec3i& contrast, const std::vector<Vec3i>& values, Vec3i& c1, Vec3i& c2) { return kMeans(contrast, values.data(), values.size(), c1, c2); }
float atan2_fast(float y, float x);                                                            // blobs/hypothesis.cpp:21-58

// pattern.h:18-61
extern const int patterns[16];
extern const int patternLUT[16];
extern const float patternAnglesb2b[25];
extern const Vec2f patternPos[5];
const float MIN_ROBOT_RADIUS = 85.0f;
const float MIN_ROBOT_FRONT_DISTANCE = 55.0f;
const float MIN_ROBOT_OPENING_ANGLE = 0.86708f;

struct TrackingState { // udpsocket.h:57-64
	int id = 0; // -1: ball, 0-15: yellow bot, 16-31: blue bot
	double timestamp = 0.0;
	float x = 0, y = 0, z = 0, w = 0;
	float vx = 0, vy = 0, vz = 0, vw = 0;
	float confidence = 0;
	int age = 0;
};

// proto/ssl_vision_detection.proto as plain structs (only the fields the reference sets: main.cpp:345-362, hypothesis.cpp:69-80,140-153)
struct DetectionBall {
	float confidence, x, y, pixel_x, pixel_y;
};
struct DetectionRobot {
	float confidence;
	uint32_t robot_id;
	float x, y, orientation, pixel_x, pixel_y, height;
};
struct DetectionFrame {
	uint32_t frame_number = 0;
	double t_capture = 0.0, t_sent = 0.0;
	bool has_t_capture_camera = false;
	double t_capture_camera = 0.0;
	std::vector<float> t_offsets;
	uint32_t camera_id = 0;
	std::vector<DetectionBall> balls;
	std::vector<DetectionRobot> robots_yellow, robots_blue;
};

struct BotHeights { // udpsocket.h:121-128, defaults of the shipped robot-heights.yml
	double maxBotHeight = 180.0;
	double defaultBotHeight = 143.0;
	double yellowBotHeight = 143.0;
	double blueBotHeight = 143.0;
};

// The members of the reference's Resources / GCSocket / VisionSocket that this stage reads and writes (Resources.h:43-73,
// udpsocket.h:106,122-128), defaults as in Resources.cpp:84-85,189-209 and the shipped robot-heights.yml.
struct BlobResources {
	int camId = 0;
	double minTrackingRadius = 20.0;
	double maxBotAcceleration = 6500.0; // mm/s^2 (1000 * tracking.max_bot_acceleration)
	double minScore = 5.0;
	double minCamEdgeDistance = 170.0;
	float minConfidence = 0.2f;
	float clippingTolerance = 10.0f;
	float geometryTolerance = 10.0f;
	float referenceForce = 0.1f;
	float historyForce = 0.7f;
	Vec3i orangeReference{192, 128, 64}, fieldReference{128, 128, 128}, yellowReference{255, 128, 0}, blueReference{0, 128, 255}, greenReference{0, 255, 128},
			pinkReference{255, 0, 128};
	Vec3i orange = orangeReference, field = fieldReference, yellow = yellowReference, blue = blueReference, green = greenReference, pink = pinkReference;
	Vec3i fieldLineColor = fieldReference;

	// The reference reaches the four robot heights through its game-controller socket (Resources.h:91 `gcSocket`, udpsocket.h:121-128: public doubles):
	// `r.gcSocket->maxBotHeight` at Resources.cpp:159, main.cpp:273, hypothesis.cpp:71,143-144.  Same spelling here, so those statements compile unchanged
	// (tests/test_reference_callsites.py); vp::GCSocket, the UDP receiver, refreshes them through heights().
	std::shared_ptr<BotHeights> gcSocket = std::make_shared<BotHeights>();

	// socket: what VisionSocket::getTrackedObjects / getReceivedOffsets hand out
	std::map<unsigned int, std::vector<TrackingState>> trackedObjects;
	std::vector<float> receivedOffsets;
	float ballRadius = 21.5f; // udpsocket.h:98

	std::shared_ptr<Perspective> perspective;
};

struct BallHypothesis { // blobs/hypothesis.h:26-41
	const Match* blob = nullptr;
	Vec2f pos;
	float score = 1.0f;
};

struct BotHypothesis { // blobs/hypothesis.h:44-91: base + Detection / Tracked variants, flattened
	const Match* blobs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
	Vec2f pos;
	float orientation = 0;
	float score = 1.0f;
	float offsetScore = 1.0f;
	int botId = -1;
	int blobAmount = 0;
	bool tracked = false;
	float trackedScore = 0.f;
	Vec3f trackedPosition;
};

// hypothesis.cpp
// Memo of calcPos's pairwise terms (hypothesis.cpp:158-168) over ONE neighbourhood of blobs.  A frame builds ~1 600 robot hypotheses from ~135 blobs, and each one
// evaluates atan2_fast + sinf + cosf for its ten blob pairs -- three quarters of the CPU stage's time (tools/dbg/detect_prof.cpp) -- yet the hypotheses of one centre blob
// (main.cpp:43-83) or one tracked robot (:85-139) keep recombining the same few blobs: the term of blob i in pattern slot a and blob j in slot b depends on nothing else.
// The memo hands back the SAME float values the direct evaluation produces, and calcPos adds them in the same order: bit-identical results.
// Blobs are named by small local ids (0 .. n-1) the caller assigns; begin(n) forgets everything.
class PairTerms {
public:
	void begin(int nBlobs);
	// sin / cos of atan2_fast(pos_j - pos_i) - patternAnglesb2b[b * 5 + a] for blob id i in slot a and blob id j in slot b (a < b)
	void term(int i, int j, int a, int b, const Match* mi, const Match* mj, float& s, float& c) {
		const size_t pair = (size_t)i * n + j, t = pair * 10 + (size_t)(a * (9 - a) / 2 + b - a - 1); // (a, b), a < b < 5, in lexicographic order: 0 .. 9
		if(termGen[t] != gen) fill(pair, t, a, b, mi, mj);
		s = sc[2 * t];
		c = sc[2 * t + 1];
	}

private:
	void fill(size_t pair, size_t t, int a, int b, const Match* mi, const Match* mj);
	int n = 0;
	uint32_t gen = 0;
	std::vector<uint32_t> angleGen, termGen; // entry valid iff == gen
	std::vector<float> angle;                // [i * n + j]
	std::vector<float> sc;                   // [((i * n + j) * 10 + slotPair) * 2]
};
// hypothesis.cpp:97-105 (the shared base of both hypothesis kinds) with the memo; ids[k] names blobs[k] (ignored where blobs[k] is null)
BotHypothesis makeBotHypothesisBase(PairTerms& terms, const Match* const blobs[5], const int ids[5]);
void calcBotId(const BlobResources& r, BotHypothesis& bot); // :216-228: only reads the blobs' colours and r's colour model, only writes bot.botId
BotHypothesis makeTrackedBotHypothesis(const BlobResources& r, const TrackingState& tracked, const Vec3f& trackedPosition, PairTerms& terms, const Match* const blobs[5], const int ids[5]);

BallHypothesis makeBallHypothesis(const BlobResources& r, const Match* blob);                                                   // :61-63
void recalcPostColorCalib(const BlobResources& r, BallHypothesis& ball);                                                       // :65-68
void recalcPostColorCalib(const BlobResources& r, BotHypothesis& bot);                                                         // :212-214, :252-255
BotHypothesis makeDetectionBotHypothesis(const BlobResources& r, const Match* a, const Match* b, const Match* c, const Match* d, const Match* e); // :208-210
BotHypothesis makeTrackedBotHypothesis(const BlobResources& r, const TrackingState& tracked, const Vec3f& trackedPosition, const Match* a, const Match* b, const Match* c,
		const Match* d, const Match* e);                                                                                      // :241-250
bool isClipping(const BlobResources& r, const BotHypothesis& self, const BotHypothesis& other);                                // :108-122
bool isClipping(const BlobResources& r, const BotHypothesis& self, const BallHypothesis& ball);                                // :124-139
void addToDetectionFrame(const BlobResources& r, const BallHypothesis& ball, DetectionFrame& detection);                       // :69-80
void addToDetectionFrame(const BlobResources& r, const BotHypothesis& bot, DetectionFrame& detection);                         // :140-153

// colorupdate.cpp:81-141
void updateColors(BlobResources& r, const std::vector<BotHypothesis>& bestBotModels, const std::vector<BallHypothesis>& ballCandidates);

// main.cpp:43-241
void generateAngleSortedBotHypotheses(const BlobResources& r, std::vector<BotHypothesis>& bots, std::vector<Match>& matches, KDTree& blobs);
void generateRadiusSearchTrackedBotHypotheses(const BlobResources& r, std::vector<BotHypothesis>& bots, std::vector<Match>& matches, KDTree& blobs, double currentTimestamp);
void filterHypothesesScore(std::vector<BotHypothesis>& bots, float threshold);
void filterHypothesesScore(std::vector<BallHypothesis>& balls, float threshold);
void filterStddevScore(std::vector<BallHypothesis>& balls, float threshold);
void filterBallsAtCamEdge(const BlobResources& r, std::vector<BallHypothesis>& balls);
void filterClippingBotBotHypotheses(const BlobResources& r, std::vector<BotHypothesis>& bots);
void generateNonclippingBallHypotheses(const BlobResources& r, const std::vector<BotHypothesis>& bots, std::vector<Match>& matches, std::vector<BallHypothesis>& balls);

// main.cpp:319-362: everything between the read-back of the blob list and socket->send(); t_sent is left for the caller
// (it is a clock reading).  `matches` must outlive the call only.
DetectionFrame detectFrame(BlobResources& r, std::vector<Match>& matches, uint32_t frameId, double startTime, double imgTimestamp);

// udpsocket.cpp:165-256 (VisionSocket::detectionTracking): updates r.trackedObjects[detection.camera_id]
void detectionTracking(BlobResources& r, const DetectionFrame& detection);

// proto2 wire encoding of SSL_WrapperPacket{detection, source = SSL_SOURCE_VISION_PROCESSOR} (proto/ssl_vision_wrapper.proto,
// ssl_vision_detection.proto), byte-identical to what protobuf's serializer emits for the fields main.cpp:345-362 sets
// (fields in field-number order, required/optional scalars always written once set, t_offsets unpacked).
std::vector<uint8_t> encodeWrapperPacket(const DetectionFrame& detection);
// and back (what VisionSocket::parse needs from a received packet: udpsocket.cpp:146-154); false on malformed input
bool decodeWrapperPacket(const uint8_t* data, size_t size, DetectionFrame& detection, bool& hasDetection, int& source);

// The geometry half of a wrapper packet (SSL_WrapperPacket.geometry = SSL_GeometryData: field size with its line segments and arcs,
// one calibration per camera), reduced to the structs Perspective::setGeometry takes.  hasGeometry = the packet carried one.
bool decodeGeometryPacket(const uint8_t* data, size_t size, FieldSize& field, std::vector<CameraCalibration>& calibs, bool& hasGeometry);

// src/udpsocket.{h,cpp}:26-113 -- the datagram socket the wrapper packets travel on.  UDPSocket opens an IPv4 datagram socket
// bound to (ip, port) with SO_REUSEADDR / SO_BROADCAST / multicast TTL 32, joins the group when ip is a multicast address, and
// runs a receiver thread that hands every datagram to parse().  (The reference starts that thread in the base-class constructor,
// i.e. before the derived vtable exists; here the derived class starts it.)
class UDPSocket {
public:
	UDPSocket(const std::string& ip, uint16_t port);
	virtual ~UDPSocket();
	UDPSocket(const UDPSocket&) = delete;
	UDPSocket& operator=(const UDPSocket&) = delete;
	bool send(const uint8_t* data, size_t size); // udpsocket.cpp:94-100 without the protobuf serialisation
	bool ok() const { return socket_ >= 0; }

protected:
	virtual void parse(const uint8_t* data, int length) = 0;
	void start(); // call at the end of the most derived constructor
	void stop();  // call at the start of the most derived destructor

private:
	void run();
	int socket_ = -1;
	unsigned char addr_[16] = {0}; // sockaddr_in
	struct Impl;
	Impl* impl = nullptr;
};

// The two team names of a game-controller Referee packet (what GCSocket::parse reads, udpsocket.cpp:317-330)
bool decodeRefereeTeams(const uint8_t* data, size_t size, std::string& yellow, std::string& blue);

// GCSocket (udpsocket.h:115-133, udpsocket.cpp:304-330): robot heights per team.  The height table (team name -> mm, the reference's
// robot-heights.yml) comes from the caller; maxBotHeight is its maximum, defaultBotHeight its mean, and a Referee packet naming a
// listed team switches that colour's height.
class GCSocket: public UDPSocket {
public:
	GCSocket(const std::string& ip, uint16_t port, const std::map<std::string, double>& botHeights);
	~GCSocket() override;
	void heights(double& maxBotHeight, double& defaultBotHeight, double& yellowBotHeight, double& blueBotHeight);
	int receivedPackets();
	bool waitForPackets(int atLeast, int timeoutMs);

protected:
	void parse(const uint8_t* data, int length) override;

private:
	struct State;
	State* state;
};

// VisionSocket (udpsocket.h:67-113), the part on this path: send the frame's SSL_WrapperPacket, receive everybody's (including
// its own, the group loops back) and keep per-camera tracked objects for the next frame's tracked hypotheses.
class VisionSocket: public UDPSocket {
public:
	VisionSocket(const std::string& ip, uint16_t port, int camId, float defaultBotHeight);
	~VisionSocket() override;
	void send(const DetectionFrame& detection);                                 // socket->send(wrapper), main.cpp:370
	std::map<unsigned int, std::vector<TrackingState>> getTrackedObjects();       // udpsocket.cpp:132-137
	int receivedPackets();                                                        // datagrams that carried a detection
	bool waitForPackets(int atLeast, int timeoutMs);
	std::vector<float> getReceivedOffsets();                                      // udpsocket.cpp:139-144: other.t_sent - local time, per sender id
	double updateTime();                                                          // udpsocket.cpp:259-282: steers the clock offset, returns it
	double getRealTime();                                                         // cameradriver.cpp:25-27: system clock + offset
	int getGeometryVersion();                                                     // udpsocket.cpp:113-130: bumped when a different geometry arrives
	void getGeometry(FieldSize& field, std::vector<CameraCalibration>& calibs);
	bool waitForGeometry(int newerThan, int timeoutMs);

protected:
	void parse(const uint8_t* data, int length) override; // udpsocket.cpp:146-163 (detection part)

private:
	struct State;
	State* state;
	const int camId;
	double getRealTimeLocked() const;
};

} // namespace vp
