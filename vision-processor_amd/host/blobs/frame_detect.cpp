// Hypothesis generation, filtering and the per-frame sequence (src/main.cpp:43-241, 319-362) plus the detection tracker that
// feeds it (src/udpsocket.cpp:165-256).  See detection.h for the parity statement.  The reference keeps hypotheses in
// std::list<std::unique_ptr<...>>; value vectors with order-preserving erases are equivalent for every loop below.
#include <algorithm>
#include <cmath>

#include "detection.h"

namespace vp {

namespace {

struct Rot2 {
	float s, c;
	explicit Rot2(float angle): s(sinf(angle)), c(cosf(angle)) {}
	Vec2f operator*(const Vec2f& v) const { return {c * v.x + (-s) * v.y, s * v.x + c * v.y}; }
};

template<typename T, typename Pred> void eraseIf(std::vector<T>& v, Pred pred) { v.erase(std::remove_if(v.begin(), v.end(), pred), v.end()); }

bool closerThanCamEdgeDistance(const BlobResources& r, const Vec2f& pos, const Vec2f& border) { // main.cpp:158-168
	const FieldSize& field = r.perspective->field;
	const float halfFieldLength = field.field_length / 2 + goalBoundaryWidth(field);
	const float halfFieldWidth = (float)(field.field_width / 2 + field.boundary_width);
	const Vec3f b3 = r.perspective->model.image2field(border, (float)r.gcSocket->maxBotHeight);
	const Vec2f borderPos{b3.x, b3.y};
	// a border point outside the field cannot cut a robot or ball in half
	const bool borderInsideField = borderPos.x >= -halfFieldLength && borderPos.x <= halfFieldLength && borderPos.y >= -halfFieldWidth && borderPos.y <= halfFieldWidth;
	const float dx = borderPos.x - pos.x, dy = borderPos.y - pos.y;
	return borderInsideField && (double)(dx * dx + dy * dy) < r.minCamEdgeDistance * r.minCamEdgeDistance;
}

} // namespace

void generateAngleSortedBotHypotheses(const BlobResources& r, std::vector<BotHypothesis>& bots, std::vector<Match>& matches, KDTree& blobs) { // main.cpp:43-83
	std::vector<Match*> botBlobs;
	std::vector<std::pair<float, Match*>> keyed;
	PairTerms terms; // calcPos's pairwise terms, shared by the hypotheses of one centre blob (detection.h)
	for(int i = 0; i < blobs.getSize(); i++) {
		Match& blob = matches[i];
		botBlobs.clear();
		blobs.rangeSearch(botBlobs, blob.pos, r.perspective->field.max_robot_radius);
		if(botBlobs.size() < 4) continue;

		// std::sort over the polar angle around the candidate centre blob; the keys are what the reference's comparator
		// recomputes per comparison, so the resulting permutation is the same
		keyed.clear();
		for(Match* m : botBlobs) keyed.emplace_back(atan2_fast(m->pos.y - blob.pos.y, m->pos.x - blob.pos.x), m);
		std::sort(keyed.begin(), keyed.end(), [](const std::pair<float, Match*>& a, const std::pair<float, Match*>& b) { return a.first < b.first; });
		for(size_t k = 0; k < keyed.size(); k++) botBlobs[k] = keyed[k].second;

		float bestBotScore = 0.0f;
		bool haveBest = false;
		BotHypothesis bestBot;
		const int size = (int)botBlobs.size();
		terms.begin(size + 1); // local ids: 0 = the centre blob, k + 1 = botBlobs[k]
		for(int a = 0; a < size; a++)
			for(int b = a + 1; b < a + size - 2; b++)
				for(int c = b + 1; c < a + size - 1; c++)
					for(int d = c + 1; d < a + size; d++) {
						// DetectionBotHypothesis (hypothesis.cpp:208-210) = the base + calcBotId.  calcBotId neither reads nor writes the score the comparison below
						// looks at, and only the winner leaves this loop: its k-means runs once per centre blob, below, instead of once per candidate
						const Match* const five[5] = {&blob, botBlobs[a], botBlobs[b % size], botBlobs[c % size], botBlobs[d % size]};
						const int ids[5] = {0, a + 1, b % size + 1, c % size + 1, d % size + 1};
						const BotHypothesis bot = makeBotHypothesisBase(terms, five, ids);
						if(bot.score > bestBotScore) {
							bestBotScore = bot.score;
							bestBot = bot;
							haveBest = true;
						}
					}
		// The reference appends bestBot even when no candidate scored above 0 (a null pointer, dereferenced by the next filter:
		// main.cpp:82,143-145).  That needs a NaN or infinite offset; here such a centre blob contributes nothing.
		if(haveBest) {
			calcBotId(r, bestBot);
			bots.push_back(bestBot);
		}
	}
}

void generateRadiusSearchTrackedBotHypotheses(const BlobResources& r, std::vector<BotHypothesis>& bots, std::vector<Match>& matches, KDTree& blobs, const double currentTimestamp) { // :85-139
	(void)matches;
	const CameraModel& model = r.perspective->model;
	std::vector<Match*> botBlobs[5];
	std::vector<int> blobIds[5];   // local id of every entry of botBlobs[i] (the same blob may lie within reach of several pattern positions)
	std::vector<Match*> distinct;
	PairTerms terms;
	for(const auto& camTracked : r.trackedObjects) {
		for(const TrackingState& tracked : camTracked.second) {
			if(tracked.id == -1) continue;

			float timeDelta = (float)(currentTimestamp - tracked.timestamp);
			const Vec3f reprojected = model.image2field(model.field2image({tracked.x, tracked.y, tracked.z}), (float)r.gcSocket->maxBotHeight);
			const Vec3f trackedPosition{reprojected.x + tracked.vx * timeDelta, reprojected.y + tracked.vy * timeDelta, tracked.w + tracked.vw * timeDelta};
			const Rot2 rotation(trackedPosition.z);

			// bounded so that a frame-rate drop or unsynchronised clocks cannot blow up the search radius
			timeDelta = std::max(std::min(timeDelta, 0.05f), 0.0f);
			// velocity comes from a two-frame difference, hence twice the acceleration term
			const float blobSearchRadius = (float)r.maxBotAcceleration * timeDelta * timeDelta + (float)r.minTrackingRadius;

			for(int i = 0; i < 5; i++) {
				botBlobs[i].clear();
				botBlobs[i].push_back(nullptr);
				const Vec2f rp = rotation * patternPos[i];
				blobs.rangeSearch(botBlobs[i], {trackedPosition.x + rp.x, trackedPosition.y + rp.y}, blobSearchRadius);
			}

			distinct.clear();
			for(int i = 0; i < 5; i++) {
				blobIds[i].assign(botBlobs[i].size(), 0);
				for(size_t k = 1; k < botBlobs[i].size(); k++) { // [0] is the "no blob at this position" entry
					const auto it = std::find(distinct.begin(), distinct.end(), botBlobs[i][k]);
					blobIds[i][k] = (int)(it - distinct.begin());
					if(it == distinct.end()) distinct.push_back(botBlobs[i][k]);
				}
			}
			terms.begin((int)distinct.size());

			float bestBotScore = 0.0f;
			bool haveBest = false;
			BotHypothesis bestBot;
			for(size_t ia = 0; ia < botBlobs[0].size(); ia++)
				for(size_t ib = 0; ib < botBlobs[1].size(); ib++) {
					Match *const a = botBlobs[0][ia], *const b = botBlobs[1][ib];
					if(b != nullptr && a == b) continue;
					for(size_t ic = 0; ic < botBlobs[2].size(); ic++) {
						Match* const c = botBlobs[2][ic];
						if(c != nullptr && (a == c || b == c)) continue;
						for(size_t id = 0; id < botBlobs[3].size(); id++) {
							Match* const d = botBlobs[3][id];
							if(d != nullptr && (a == d || b == d || c == d)) continue;
							for(size_t ie = 0; ie < botBlobs[4].size(); ie++) {
								Match* const e = botBlobs[4][ie];
								if(e != nullptr && (a == e || b == e || c == e || d == e)) continue;
								const Match* const five[5] = {a, b, c, d, e};
								const int ids[5] = {blobIds[0][ia], blobIds[1][ib], blobIds[2][ic], blobIds[3][id], blobIds[4][ie]};
								BotHypothesis bot = makeTrackedBotHypothesis(r, tracked, trackedPosition, terms, five, ids);
								if(bot.score > bestBotScore) {
									bestBotScore = bot.score;
									bestBot = bot;
									haveBest = true;
								}
							}
						}
					}
				}
			if(haveBest) bots.push_back(bestBot);
		}
	}
}

void filterHypothesesScore(std::vector<BotHypothesis>& bots, float threshold) { // :141-150
	eraseIf(bots, [&](const BotHypothesis& b) { return b.score <= threshold; });
}

void filterHypothesesScore(std::vector<BallHypothesis>& balls, float threshold) {
	eraseIf(balls, [&](const BallHypothesis& b) { return b.score <= threshold; });
}

void filterStddevScore(std::vector<BallHypothesis>& balls, float threshold) { // :152-161
	eraseIf(balls, [&](const BallHypothesis& b) { return b.blob->score <= threshold; });
}

void filterBallsAtCamEdge(const BlobResources& r, std::vector<BallHypothesis>& balls) { // :170-187
	const CameraModel& model = r.perspective->model;
	eraseIf(balls, [&](const BallHypothesis& ball) {
		const Vec2f imgPos = model.field2image({ball.pos.x, ball.pos.y, (float)r.gcSocket->maxBotHeight});
		return closerThanCamEdgeDistance(r, ball.pos, {0.0f, imgPos.y}) || closerThanCamEdgeDistance(r, ball.pos, {(float)(model.size.x - 1), imgPos.y}) ||
				closerThanCamEdgeDistance(r, ball.pos, {imgPos.x, 0.0f}) || closerThanCamEdgeDistance(r, ball.pos, {imgPos.x, (float)(model.size.y - 1)});
	});
}

void filterClippingBotBotHypotheses(const BlobResources& r, std::vector<BotHypothesis>& bots) { // :189-219
	// The reference's first inner loop compares each hypothesis with ITSELF (`bot2 = *it1`, :194) and therefore never removes
	// anything; only the second loop acts: walking in list order, every still-present hypothesis removes all others that
	// clip with it and do not score higher.  Removed entries take no further part, exactly as erased list nodes.
	std::vector<char> gone(bots.size(), 0);
	for(size_t i = 0; i < bots.size(); i++) {
		if(gone[i]) continue;
		for(size_t j = 0; j < bots.size(); j++) {
			if(gone[j] || j == i) continue;
			if(bots[j].score <= bots[i].score && isClipping(r, bots[i], bots[j])) gone[j] = 1;
		}
	}
	size_t k = 0;
	for(size_t i = 0; i < bots.size(); i++)
		if(!gone[i]) bots[k++] = bots[i];
	bots.resize(k);
}

void generateNonclippingBallHypotheses(const BlobResources& r, const std::vector<BotHypothesis>& bots, std::vector<Match>& matches, std::vector<BallHypothesis>& balls) { // :221-239
	for(const Match& match : matches) {
		const BallHypothesis ball = makeBallHypothesis(r, &match);
		bool nextToBot = false;
		for(const BotHypothesis& bot : bots)
			if(isClipping(r, bot, ball)) {
				nextToBot = true;
				break;
			}
		if(!nextToBot) balls.push_back(ball);
	}
}

DetectionFrame detectFrame(BlobResources& r, std::vector<Match>& matches, const uint32_t frameId, const double startTime, const double imgTimestamp) { // main.cpp:319-362
	std::vector<BotHypothesis> botHypotheses;
	std::vector<BallHypothesis> ballHypotheses;

	if(!matches.empty()) {
		KDTree blobs(&matches[0]);
		for(size_t i = 1; i < matches.size(); i++) blobs.insert(&matches[i]);

		generateRadiusSearchTrackedBotHypotheses(r, botHypotheses, matches, blobs, startTime);
		generateAngleSortedBotHypotheses(r, botHypotheses, matches, blobs);
		filterHypothesesScore(botHypotheses, r.minConfidence);
		filterClippingBotBotHypotheses(r, botHypotheses);
		generateNonclippingBallHypotheses(r, botHypotheses, matches, ballHypotheses);
	}

	updateColors(r, botHypotheses, ballHypotheses);
	for(BotHypothesis& bot : botHypotheses) recalcPostColorCalib(r, bot);
	for(BallHypothesis& ball : ballHypotheses) recalcPostColorCalib(r, ball);

	filterHypothesesScore(ballHypotheses, r.minConfidence);
	filterBallsAtCamEdge(r, ballHypotheses);
	filterStddevScore(ballHypotheses, (float)r.minScore);

	DetectionFrame detection;
	detection.frame_number = frameId;
	detection.t_capture = startTime;
	if(imgTimestamp != 0) {
		detection.has_t_capture_camera = true;
		detection.t_capture_camera = imgTimestamp;
	}
	detection.camera_id = (uint32_t)r.camId;
	for(const BotHypothesis& bot : botHypotheses) addToDetectionFrame(r, bot, detection);
	for(const BallHypothesis& ball : ballHypotheses) addToDetectionFrame(r, ball, detection);
	for(const float offset : r.receivedOffsets) detection.t_offsets.push_back(offset);
	return detection;
}

namespace {

void trackBots(const double timestamp, const std::vector<DetectionRobot>& bots, const std::vector<TrackingState>& previous, std::vector<TrackingState>& objects, int idOffset) { // udpsocket.cpp:165-202
	for(const DetectionRobot& bot : bots) {
		const float height = bot.height; // always present in frames of this producer
		double nextDistance = INFINITY;
		const TrackingState* nextOldBot = nullptr;
		for(const TrackingState& oldBot : previous) {
			if(oldBot.id != (int)bot.robot_id + idOffset) continue;
			const float dX = bot.x - oldBot.x, dY = bot.y - oldBot.y;
			const double distance = dX * dX + dY * dY;
			if(distance > nextDistance) continue;
			nextDistance = distance;
			nextOldBot = &oldBot;
		}
		if(nextOldBot == nullptr) {
			objects.push_back({(int)bot.robot_id + idOffset, timestamp, bot.x, bot.y, height, bot.orientation, 0.0f, 0.0f, 0.0f, 0.0f, bot.confidence, 1});
		} else {
			const float timeDelta = (float)(timestamp - nextOldBot->timestamp);
			objects.push_back({nextOldBot->id, timestamp, bot.x, bot.y, height, bot.orientation, (bot.x - nextOldBot->x) / timeDelta, (bot.y - nextOldBot->y) / timeDelta, 0.0f,
					(bot.orientation - nextOldBot->w) / timeDelta, bot.confidence, nextOldBot->age + 1});
		}
	}
}

} // namespace

void detectionTracking(BlobResources& r, const DetectionFrame& detection) { // udpsocket.cpp:204-256
	const double timestamp = detection.t_capture;
	const std::vector<TrackingState> previous = r.trackedObjects[detection.camera_id];
	std::vector<TrackingState> objects;

	for(const DetectionBall& ball : detection.balls) {
		const float z = r.ballRadius; // ball.z is never set by this producer (hypothesis.cpp:77)
		double nextDistance = INFINITY;
		const TrackingState* nextOldBall = nullptr;
		for(const TrackingState& oldBall : previous) {
			if(oldBall.id != -1) continue;
			const float dX = ball.x - oldBall.x, dY = ball.y - oldBall.y, dZ = z - oldBall.z;
			const double distance = dX * dX + dY * dY + dZ * dZ;
			if(distance > nextDistance) continue;
			nextDistance = distance;
			nextOldBall = &oldBall;
		}
		if(nextOldBall == nullptr) {
			objects.push_back({-1, timestamp, ball.x, ball.y, z, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, ball.confidence, 1});
		} else {
			const float timeDelta = (float)(timestamp - nextOldBall->timestamp);
			objects.push_back({-1, timestamp, ball.x, ball.y, z, 0.0f, (ball.x - nextOldBall->x) / timeDelta, (ball.y - nextOldBall->y) / timeDelta, (z - nextOldBall->z) / timeDelta, 0.0f,
					ball.confidence, nextOldBall->age + 1});
		}
	}

	trackBots(timestamp, detection.robots_yellow, previous, objects, 0);
	trackBots(timestamp, detection.robots_blue, previous, objects, 16);
	r.trackedObjects[detection.camera_id] = objects;
}

} // namespace vp
