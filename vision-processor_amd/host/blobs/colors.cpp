// Online colour calibration from the frame's surviving hypotheses (blobs/colorupdate.cpp:21-141).  See detection.h for the
// parity statement.  Written around two small helpers -- an integer colour accumulator and a pull towards (reference, history) --
// instead of the reference's six parallel variable sets; the arithmetic per channel is the reference's, in its order.
#include <algorithm>
#include <cmath>

#include "detection.h"

namespace vp {

namespace {

struct ColorSum { // Vector3i sum + count; mean() is the component-wise integer division the reference writes as `sum / n`
	int r = 0, g = 0, b = 0, n = 0;
	void add(const Vec3i& c) {
		r += c.x;
		g += c.y;
		b += c.z;
		n++;
	}
	Vec3i mean() const { return {r / n, g / n, b / n}; }
};

// colorupdate.cpp:73-76: new = referenceForce * reference + historyForce * previous + (1 - referenceForce - historyForce) * measured,
// per channel in binary32, left to right, truncated towards zero
struct Pull {
	float toReference, toHistory, toMeasurement;
	explicit Pull(const BlobResources& r): toReference(r.referenceForce), toHistory(r.historyForce), toMeasurement(1.0f - r.referenceForce - r.historyForce) {}
	int channel(int reference, int previous, int measured) const { return (int)(toReference * (float)reference + toHistory * (float)previous + toMeasurement * (float)measured); }
	Vec3i operator()(const Vec3i& reference, const Vec3i& previous, const Vec3i& measured) const {
		return {channel(reference.x, previous.x, measured.x), channel(reference.y, previous.y, measured.y), channel(reference.z, previous.z, measured.z)};
	}
};

// squared distance of a point to a line segment, projection parameter clamped to the segment (colorupdate.cpp:21-30)
float segmentDistanceSq(const FieldLine& segment, const Vec2f& q) {
	const float ax = segment.p2.x - segment.p1.x, ay = segment.p2.y - segment.p1.y;
	const float bx = q.x - segment.p1.x, by = q.y - segment.p1.y;
	const float along = std::max(0.0f, std::min(1.0f, (bx * ax + by * ay) / (ax * ax + ay * ay)));
	const float ex = bx - along * ax, ey = by - along * ay;
	return ex * ex + ey * ey;
}

// Is the ball candidate, re-projected to ball height, on a painted line or arc (within half the line thickness plus the geometry
// tolerance)?  colorupdate.cpp:32-55
bool onFieldMarking(const BlobResources& r, const BallHypothesis& candidate) {
	const Perspective& view = *r.perspective;
	const Vec2f pixel = view.model.field2image({candidate.pos.x, candidate.pos.y, (float)r.gcSocket->maxBotHeight});
	const Vec3f onGround = view.model.image2field(pixel, view.field.ball_radius);
	const Vec2f q{onGround.x, onGround.y};
	const float reach = (float)view.field.line_thickness / 2 + r.geometryTolerance;

	const float reachSq = reach * reach;
	for(const FieldLine& segment : view.field.field_lines)
		if(segmentDistanceSq(segment, q) <= reachSq) return true;

	for(const FieldArc& arc : view.field.field_arcs) {
		const float dx = q.x - arc.center.x, dy = q.y - arc.center.y;
		float bearing = atan2f(dy, dx);
		if(bearing < 0) bearing = (float)(bearing + 2 * M_PI); // float += double: summed in double, then narrowed
		const float radial = sqrtf(dx * dx + dy * dy) - arc.radius;
		if(fabsf(radial) <= reach && bearing >= arc.a1 && bearing <= arc.a2) return true;
	}
	return false;
}

} // namespace

void updateColors(BlobResources& r, const std::vector<BotHypothesis>& bestBotModels, const std::vector<BallHypothesis>& ballCandidates) { // :78-141
	const struct { Vec3i orange, field, yellow, blue, green, pink; } before = {r.orange, r.field, r.yellow, r.blue, r.green, r.pink}; // previous frame's colours: the "history"
	const Pull pull(r);

	// side blobs vote for green / pink according to the id their robot was given; centre blobs are collected for the team split
	ColorSum sides[2]; // [0] pink, [1] green
	std::vector<Vec3i> centres;
	for(const BotHypothesis& bot : bestBotModels) {
		if(bot.blobs[0]) centres.push_back(bot.blobs[0]->color);
		const int code = patterns[bot.botId % 16];
		for(int slot = 1; slot < 5; slot++)
			if(bot.blobs[slot]) sides[(code >> (4 - slot)) & 1].add(bot.blobs[slot]->color);
	}
	if(sides[0].n > 0) r.pink = pull(r.pinkReference, before.pink, sides[0].mean());
	if(sides[1].n > 0) r.green = pull(r.greenReference, before.green, sides[1].mean());

	// yellow / blue: two-means over the centre blobs, seeded with the current team colours, contrasted against pink (:123-126)
	if(kMeans(r.pink, centres, r.yellow, r.blue)) {
		r.yellow = pull(r.yellowReference, before.yellow, r.yellow);
		r.blue = pull(r.blueReference, before.blue, r.blue);
	}

	// orange / field: two-means over the CENTRE-PIXEL colours of the ball candidates (not their disc means, :131-132), contrasted against blue
	std::vector<Vec3i> cores;
	cores.reserve(ballCandidates.size());
	for(const BallHypothesis& candidate : ballCandidates) cores.push_back(candidate.blob->center);
	if(kMeans(r.blue, cores, r.orange, r.field)) {
		r.orange = pull(r.orangeReference, before.orange, r.orange);
		r.field = pull(r.fieldReference, before.field, r.field);
	}

	// field-line colour: mean disc colour of the candidates that sit on a marking, once there are more than two (:57-71)
	ColorSum onLines;
	for(const BallHypothesis& candidate : ballCandidates)
		if(onFieldMarking(r, candidate)) onLines.add(candidate.blob->color);
	if(onLines.n > 2) r.fieldLineColor = onLines.mean();
}

} // namespace vp
