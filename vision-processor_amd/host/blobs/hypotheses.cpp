// Ball and robot hypotheses (blobs/hypothesis.cpp) and the pattern tables (pattern.h).  See detection.h for the parity
// statement.  Float expressions keep the reference's association; `2.0f * M_PI` is a double product narrowed at the call.
#include <algorithm>
#include <cmath>

#include "detection.h"

namespace vp {

// pattern.h:18-53: 1 = green, 0 = pink, most significant bit first, counter-clockwise from the robot's heading
const int patterns[16] = {0b0100, 0b1100, 0b1101, 0b0101, 0b0010, 0b1010, 0b1011, 0b0011, 0b1111, 0b0000, 0b0110, 0b1001, 0b1110, 0b1000, 0b0111, 0b0001};
const int patternLUT[16] = {9, 15, 4, 7, 0, 3, 10, 14, 13, 11, 5, 6, 1, 2, 12, 8};
const float patternAnglesb2b[25] = {
		0.f, -2.13940875f, -0.56861242f, 0.56861242f, 2.13940875f,  //
		1.00218391f, 0.f, 0.21678574f, 0.78539816f, 1.57079633f,    //
		2.57298023f, -2.92480691f, 0.f, 1.57079633f, 2.35619449f,   //
		-2.57298023f, -2.35619449f, -1.57079633f, 0.f, 2.92480691f, //
		-1.00218391f, -1.57079633f, -0.78539816f, -0.21678574f, 0.f};
const Vec2f patternPos[5] = {{0.f, 0.f}, {35.f, 54.772f}, {-54.772f, 35.f}, {-54.772f, -35.f}, {35.f, -54.772f}};

namespace {

const float TWO_PI_F = (float)(2.0f * M_PI);

struct Rot2 { // Eigen::Rotation2Df(angle) * v == [[c, -s], [s, c]] * v
	float s, c;
	explicit Rot2(float angle): s(sinf(angle)), c(cosf(angle)) {}
	Vec2f operator*(const Vec2f& v) const { return {c * v.x + (-s) * v.y, s * v.x + c * v.y}; }
};

inline float sq(const Vec2f& v) { return v.x * v.x + v.y * v.y; }

float atanPoly(float x) { // hypothesis.cpp:21-33: odd degree-11 minimax polynomial, Horner with fused multiply-adds
	const float a1 = 0.99997726f, a3 = -0.33262347f, a5 = 0.19354346f, a7 = -0.11643287f, a9 = 0.05265332f, a11 = -0.01172120f;
	const float x2 = x * x;
	float p = fmaf(x2, a11, a9);
	p = fmaf(x2, p, a7);
	p = fmaf(x2, p, a5);
	p = fmaf(x2, p, a3);
	p = fmaf(x2, p, a1);
	return x * p;
}

void calcColorScore(const BlobResources& r, BallHypothesis& ball) { // :82-94
	const int falseOrange = sqDiff(ball.blob->color, r.field);
	const int orange = sqDiff(ball.blob->color, r.orange);
	const int fieldLine = sqDiff(ball.blob->color, r.fieldLineColor);
	if(falseOrange <= orange || fieldLine <= orange) {
		ball.score = 0;
		return;
	}
	ball.score *= 1 - (float)orange / (float)falseOrange;
}

Rot2 calcPos(BotHypothesis& bot, PairTerms* terms, const int* ids) { // :155-191; returns Rotation2Df(orientation), which calcOffsetScore builds again in the reference (same value)
	float oSin = 0, oCos = 0;
	for(int a = 0; a < 5; a++) {
		if(bot.blobs[a] == nullptr) continue;
		for(int b = a + 1; b < 5; b++) {
			if(bot.blobs[b] == nullptr) continue;
			float s, c;
			if(terms) {
				terms->term(ids[a], ids[b], a, b, bot.blobs[a], bot.blobs[b], s, c);
			} else {
				const Vec2f diff{bot.blobs[b]->pos.x - bot.blobs[a]->pos.x, bot.blobs[b]->pos.y - bot.blobs[a]->pos.y};
				const float angleDelta = atan2_fast(diff.y, diff.x) - patternAnglesb2b[b * 5 + a];
				s = sinf(angleDelta);
				c = cosf(angleDelta);
			}
			oSin += s;
			oCos += c;
		}
	}
	if(bot.blobAmount < 1) return Rot2(bot.orientation);
	if(bot.blobAmount > 1) bot.orientation = atan2_fast(oSin, oCos);

	const Rot2 rotation(bot.orientation);
	bot.pos = {0, 0};
	for(int i = 0; i < 5; i++) {
		if(bot.blobs[i] == nullptr) continue;
		const Vec2f rp = rotation * patternPos[i];
		bot.pos.x += bot.blobs[i]->pos.x - rp.x;
		bot.pos.y += bot.blobs[i]->pos.y - rp.y;
	}
	bot.pos.x /= (float)bot.blobAmount;
	bot.pos.y /= (float)bot.blobAmount;
	return rotation;
}

void calcOffsetScore(BotHypothesis& bot, const Rot2& rotation) { // :193-205
	for(int i = 0; i < 5; i++) {
		const Match* blob = bot.blobs[i];
		if(blob == nullptr) continue;
		const Vec2f rp = rotation * patternPos[i];
		const Vec2f offset{(blob->pos.x - (bot.pos.x + rp.x)) / 10.0f, (blob->pos.y - (bot.pos.y + rp.y)) / 10.0f}; // 1 cm offset -> 0.5
		bot.offsetScore = std::min(bot.offsetScore, 1 / (1 + sq(offset)));
	}
	bot.score = bot.offsetScore;
}

BotHypothesis makeBase(const Match* a, const Match* b, const Match* c, const Match* d, const Match* e, PairTerms* terms = nullptr, const int* ids = nullptr) { // :97-105
	BotHypothesis bot;
	const Match* in[5] = {a, b, c, d, e};
	for(int i = 0; i < 5; i++) {
		bot.blobs[i] = in[i];
		if(in[i] != nullptr) bot.blobAmount++;
	}
	const Rot2 rotation = calcPos(bot, terms, ids);
	calcOffsetScore(bot, rotation);
	return bot;
}

} // namespace

void PairTerms::begin(int nBlobs) {
	n = nBlobs;
	if(++gen == 0) { // the stamps wrapped: forget them all
		std::fill(angleGen.begin(), angleGen.end(), 0u);
		std::fill(termGen.begin(), termGen.end(), 0u);
		gen = 1;
	}
	const size_t pairs = (size_t)n * n;
	if(angleGen.size() < pairs) {
		angleGen.assign(pairs, 0u);
		angle.resize(pairs);
		termGen.assign(pairs * 10, 0u);
		sc.resize(pairs * 20);
	}
}

void PairTerms::fill(size_t pair, size_t t, int a, int b, const Match* mi, const Match* mj) {
	if(angleGen[pair] != gen) {
		angle[pair] = atan2_fast(mj->pos.y - mi->pos.y, mj->pos.x - mi->pos.x);
		angleGen[pair] = gen;
	}
	const float angleDelta = angle[pair] - patternAnglesb2b[b * 5 + a];
	sc[2 * t] = sinf(angleDelta);
	sc[2 * t + 1] = cosf(angleDelta);
	termGen[t] = gen;
}

BotHypothesis makeBotHypothesisBase(PairTerms& terms, const Match* const blobs[5], const int ids[5]) { return makeBase(blobs[0], blobs[1], blobs[2], blobs[3], blobs[4], &terms, ids); }

void calcBotId(const BlobResources& r, BotHypothesis& bot) { // :216-228
	Vec3i green = r.green, pink = r.pink;
	const Vec3i sides[4] = {bot.blobs[1]->color, bot.blobs[2]->color, bot.blobs[3]->color, bot.blobs[4]->color};
	kMeans(bot.blobs[0]->color, sides, 4, green, pink);
	auto isGreen = [&](int i) { return sqDiff(bot.blobs[i]->color, green) < sqDiff(bot.blobs[i]->color, pink) ? 1 : 0; };
	const int team = sqDiff(bot.blobs[0]->color, r.blue) < sqDiff(bot.blobs[0]->color, r.yellow) ? 16 : 0;
	bot.botId = team + patternLUT[(isGreen(1) << 3) + (isGreen(2) << 2) + (isGreen(3) << 1) + isGreen(4)];
}

namespace {

void calcTrackingScore(const BlobResources& r, BotHypothesis& bot) { // :257-282
	if(bot.blobAmount < 2) {
		bot.score = 0.0f;
		return;
	}
	for(int i = 0; i < 5; i++) {
		const Match* blob = bot.blobs[i];
		if(blob == nullptr) continue;
		Vec3i blobColor, oppositeColor;
		if(i == 0) {
			blobColor = bot.botId >= 16 ? r.blue : r.yellow;
			oppositeColor = bot.botId >= 16 ? r.yellow : r.blue;
		} else {
			const bool green = (patterns[bot.botId % 16] >> (4 - i)) & 1;
			blobColor = green ? r.green : r.pink;
			oppositeColor = green ? r.pink : r.green;
		}
		if(sqDiff(blob->color, oppositeColor) - sqDiff(blob->color, blobColor) <= 0) {
			bot.score = 0.0f;
			return;
		}
	}
}

} // namespace

float atan2_fast(const float y, const float x) { // :35-58
	const float pi = (float)M_PI, pi_2 = (float)M_PI_2;
	const bool swap = fabsf(x) < fabsf(y);
	const float divisor = swap ? y : x;
	if(divisor == 0.0) return swap ? x : y; // both arguments zero
	const float atanInput = (swap ? x : y) / divisor; // in [-1, +1]
	float res = atanPoly(atanInput);
	res = swap ? copysignf(pi_2, atanInput) - res : res;
	if(x < 0.0f) res = copysignf(pi, y) + res;
	return res;
}

BallHypothesis makeBallHypothesis(const BlobResources& r, const Match* blob) {
	BallHypothesis ball;
	ball.blob = blob;
	ball.pos = blob->pos;
	calcColorScore(r, ball);
	return ball;
}

void recalcPostColorCalib(const BlobResources& r, BallHypothesis& ball) {
	ball.score = 1.0f;
	calcColorScore(r, ball);
}

void recalcPostColorCalib(const BlobResources& r, BotHypothesis& bot) {
	if(bot.tracked) {
		bot.score = bot.offsetScore;
		calcTrackingScore(r, bot);
	} else {
		calcBotId(r, bot);
	}
}

BotHypothesis makeDetectionBotHypothesis(const BlobResources& r, const Match* a, const Match* b, const Match* c, const Match* d, const Match* e) {
	BotHypothesis bot = makeBase(a, b, c, d, e);
	calcBotId(r, bot);
	return bot;
}

static BotHypothesis finishTracked(const BlobResources& r, const TrackingState& tracked, const Vec3f& trackedPosition, BotHypothesis bot);

BotHypothesis makeTrackedBotHypothesis(const BlobResources& r, const TrackingState& tracked, const Vec3f& trackedPosition, const Match* a, const Match* b, const Match* c,
		const Match* d, const Match* e) {
	return finishTracked(r, tracked, trackedPosition, makeBase(a, b, c, d, e));
}

BotHypothesis makeTrackedBotHypothesis(const BlobResources& r, const TrackingState& tracked, const Vec3f& trackedPosition, PairTerms& terms, const Match* const blobs[5], const int ids[5]) {
	return finishTracked(r, tracked, trackedPosition, makeBase(blobs[0], blobs[1], blobs[2], blobs[3], blobs[4], &terms, ids));
}

static BotHypothesis finishTracked(const BlobResources& r, const TrackingState& tracked, const Vec3f& trackedPosition, BotHypothesis bot) { // hypothesis.cpp:241-250
	bot.tracked = true;
	bot.trackedScore = tracked.confidence;
	bot.trackedPosition = trackedPosition;
	bot.botId = tracked.id;

	const float rotationOffset = remainderf(bot.orientation - trackedPosition.z, TWO_PI_F) / (float)M_PI;
	const Vec2f d10{(bot.pos.x - trackedPosition.x) / 10.0f, (bot.pos.y - trackedPosition.y) / 10.0f};
	bot.offsetScore *= 1 / (1 + sq(d10) + rotationOffset * rotationOffset); // 1 cm offset -> 0.5
	bot.offsetScore *= (float)bot.blobAmount / 5.f;
	recalcPostColorCalib(r, bot);
	return bot;
}

bool isClipping(const BlobResources& r, const BotHypothesis& self, const BotHypothesis& other) {
	const Vec2f diff{other.pos.x - self.pos.x, other.pos.y - self.pos.y};
	const float sqDistance = sq(diff);
	if(sqDistance >= (2 * MIN_ROBOT_RADIUS) * (2 * MIN_ROBOT_RADIUS)) return false; // circle - circle early out

	const float diffAngle = atan2_fast(diff.y, diff.x);
	const float selfAngle = remainderf(diffAngle - self.orientation, TWO_PI_F);
	const float otherAngle = remainderf(diffAngle - other.orientation, TWO_PI_F);
	const float minDistance = (fabsf(selfAngle) < MIN_ROBOT_OPENING_ANGLE ? MIN_ROBOT_FRONT_DISTANCE / cosf(selfAngle) : MIN_ROBOT_RADIUS) +
			(fabsf(otherAngle) < MIN_ROBOT_OPENING_ANGLE ? MIN_ROBOT_FRONT_DISTANCE / cosf(otherAngle) : MIN_ROBOT_RADIUS) - r.clippingTolerance;
	return sqDistance < minDistance * minDistance;
}

bool isClipping(const BlobResources& r, const BotHypothesis& self, const BallHypothesis& ball) {
	const float clippedBallRadius = 0.48837f * r.perspective->field.ball_radius; // up to 20 % of the ball's top view may overlap the robot
	const Vec2f diff{ball.pos.x - self.pos.x, ball.pos.y - self.pos.y};
	const float sqDistance = sq(diff);
	float minDistance = MIN_ROBOT_RADIUS + clippedBallRadius;
	if(sqDistance >= minDistance * minDistance) return false;

	const float angle = remainderf(atan2_fast(diff.y, diff.x) - self.orientation, TWO_PI_F);
	if(fabsf(angle) >= MIN_ROBOT_OPENING_ANGLE) return true;

	minDistance = (MIN_ROBOT_FRONT_DISTANCE + clippedBallRadius) / cosf(angle) - r.clippingTolerance;
	return sqDistance < minDistance * minDistance;
}

void addToDetectionFrame(const BlobResources& r, const BallHypothesis& ball, DetectionFrame& detection) {
	const CameraModel& model = r.perspective->model;
	const Vec2f imgPos = model.field2image({ball.pos.x, ball.pos.y, (float)r.gcSocket->maxBotHeight});
	const Vec3f ballPos = model.image2field(imgPos, r.perspective->field.ball_radius);
	detection.balls.push_back({ball.score, ballPos.x, ballPos.y, imgPos.x, imgPos.y});
}

void addToDetectionFrame(const BlobResources& r, const BotHypothesis& bot, DetectionFrame& detection) {
	const CameraModel& model = r.perspective->model;
	const bool yellow = bot.botId < 16;
	const Vec2f imgPos = model.field2image({bot.pos.x, bot.pos.y, (float)r.gcSocket->maxBotHeight});
	const Vec3f botPos = model.image2field(imgPos, (float)(yellow ? r.gcSocket->yellowBotHeight : r.gcSocket->blueBotHeight));
	// botId % 16 is assigned to a uint32 field; a tracked id is never negative here (id -1 entries are skipped, main.cpp:87-88)
	(yellow ? detection.robots_yellow : detection.robots_blue).push_back({bot.score, (uint32_t)(bot.botId % 16), botPos.x, botPos.y, bot.orientation, imgPos.x, imgPos.y, botPos.z});
}

} // namespace vp
