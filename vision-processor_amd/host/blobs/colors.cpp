// Online colour calibration from the frame's surviving hypotheses (blobs/colorupdate.cpp:21-141).  See detection.h for the
// parity statement.
#include <algorithm>
#include <cmath>

#include "detection.h"

namespace vp {

namespace {

float sqPointLineSegmentDistance(const FieldLine& line, const Vec2f& point) { // colorupdate.cpp:21-30
	const Vec2f v{line.p2.x - line.p1.x, line.p2.y - line.p1.y};
	const Vec2f w{point.x - line.p1.x, point.y - line.p1.y};
	const float t = std::max(0.0f, std::min(1.0f, (w.x * v.x + w.y * v.y) / (v.x * v.x + v.y * v.y)));
	const Vec2f delta{w.x - t * v.x, w.y - t * v.y};
	return delta.x * delta.x + delta.y * delta.y;
}

bool ballAtLine(const BlobResources& r, const BallHypothesis& ball) { // :32-55
	const Perspective& p = *r.perspective;
	const Vec2f imgPos = p.model.field2image({ball.pos.x, ball.pos.y, (float)r.maxBotHeight});
	const Vec3f ballPos3 = p.model.image2field(imgPos, p.field.ball_radius);
	const Vec2f ballPos{ballPos3.x, ballPos3.y};

	const float maxLineDistance = (float)p.field.line_thickness / 2 + r.geometryTolerance;
	const float sqMaxLineDistance = maxLineDistance * maxLineDistance;
	for(const FieldLine& line : p.field.field_lines)
		if(sqPointLineSegmentDistance(line, ballPos) <= sqMaxLineDistance) return true;

	for(const FieldArc& arc : p.field.field_arcs) {
		const Vec2f pixel2center{ballPos.x - arc.center.x, ballPos.y - arc.center.y};
		float angle = atan2f(pixel2center.y, pixel2center.x);
		if(angle < 0) angle = (float)(angle + 2 * M_PI); // float += double: the sum is formed in double, then narrowed
		if(fabsf(sqrtf(pixel2center.x * pixel2center.x + pixel2center.y * pixel2center.y) - arc.radius) <= maxLineDistance && angle >= arc.a1 && angle <= arc.a2) return true;
	}
	return false;
}

void determineFieldLineBlobColor(BlobResources& r, const std::vector<BallHypothesis>& balls) { // :57-71
	Vec3i colorSum;
	int amount = 0;
	for(const BallHypothesis& ball : balls) {
		if(ballAtLine(r, ball)) {
			colorSum = {colorSum.x + ball.blob->color.x, colorSum.y + ball.blob->color.y, colorSum.z + ball.blob->color.z};
			amount++;
		}
	}
	if(amount > 2) r.fieldLineColor = {colorSum.x / amount, colorSum.y / amount, colorSum.z / amount};
}

void updateColor(const BlobResources& r, const Vec3i& reference, const Vec3i& oldColor, Vec3i& color) { // :73-76
	const float updateForce = 1.0f - r.referenceForce - r.historyForce;
	auto mix = [&](int ref, int old, int cur) { return (int)(r.referenceForce * (float)ref + r.historyForce * (float)old + updateForce * (float)cur); };
	color = {mix(reference.x, oldColor.x, color.x), mix(reference.y, oldColor.y, color.y), mix(reference.z, oldColor.z, color.z)};
}

} // namespace

void updateColors(BlobResources& r, const std::vector<BotHypothesis>& bestBotModels, const std::vector<BallHypothesis>& ballCandidates) { // :78-141
	const Vec3i oldField = r.field, oldOrange = r.orange, oldYellow = r.yellow, oldBlue = r.blue, oldGreen = r.green, oldPink = r.pink;

	std::vector<Vec3i> centerBlobs;
	Vec3i pink, green;
	int pinkN = 0, greenN = 0;
	for(const BotHypothesis& model : bestBotModels) {
		if(model.blobs[0] != nullptr) centerBlobs.push_back(model.blobs[0]->color);
		const int botId = model.botId % 16;
		for(int i = 1; i < 5; i++) {
			const Match* blob = model.blobs[i];
			if(blob == nullptr) continue;
			if((patterns[botId] >> (4 - i)) & 1) {
				green = {green.x + blob->color.x, green.y + blob->color.y, green.z + blob->color.z};
				greenN++;
			} else {
				pink = {pink.x + blob->color.x, pink.y + blob->color.y, pink.z + blob->color.z};
				pinkN++;
			}
		}
	}

	if(pinkN > 0) {
		r.pink = {pink.x / pinkN, pink.y / pinkN, pink.z / pinkN};
		updateColor(r, r.pinkReference, oldPink, r.pink);
	}
	if(greenN > 0) {
		r.green = {green.x / greenN, green.y / greenN, green.z / greenN};
		updateColor(r, r.greenReference, oldGreen, r.green);
	}

	if(kMeans(r.pink, centerBlobs, r.yellow, r.blue)) {
		updateColor(r, r.yellowReference, oldYellow, r.yellow);
		updateColor(r, r.blueReference, oldBlue, r.blue);
	}

	std::vector<Vec3i> ballBlobs; // NB the ball colour model is fed the CENTRE pixel colours, not the disc means (:131-132)
	for(const BallHypothesis& ball : ballCandidates) ballBlobs.push_back(ball.blob->center);
	if(kMeans(r.blue, ballBlobs, r.orange, r.field)) {
		updateColor(r, r.orangeReference, oldOrange, r.orange);
		updateColor(r, r.fieldReference, oldField, r.field);
	}

	determineFieldLineBlobColor(r, ballCandidates);
}

} // namespace vp
