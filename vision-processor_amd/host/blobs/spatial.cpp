// KDTree (blobs/kdtree.cpp:18-37) and the two-cluster colour k-means (blobs/kmeans.cpp:19-90).  See detection.h for the
// parity statement.
#include <algorithm>
#include <cmath>

#include "detection.h"

namespace vp {

namespace {
inline float coord(const Vec2f& v, int dim) { return dim == 0 ? v.x : v.y; }
} // namespace

void KDTree::insert(Match* m) { // kdtree.cpp:18-27: strictly-less goes left, everything else (ties, NaN) right
	int n = 0;
	while(true) {
		const Node& node = nodes[n];
		const bool goLeft = coord(m->pos, node.dim) < coord(node.data->pos, node.dim);
		const int child = goLeft ? node.left : node.right;
		if(child >= 0) {
			n = child;
			continue;
		}
		const int created = (int)nodes.size();
		const int dim = (node.dim + 1) % 2;
		if(goLeft) nodes[n].left = created;
		else nodes[n].right = created;
		nodes.push_back({m, -1, -1, dim});
		return;
	}
}

void KDTree::search(int n, std::vector<Match*>& values, const Vec2f& point, const float radius) const { // kdtree.cpp:29-37
	const Node& node = nodes[n];
	const float dx = node.data->pos.x - point.x, dy = node.data->pos.y - point.y;
	if(std::sqrt(dx * dx + dy * dy) <= radius) values.push_back(node.data);
	const float p = coord(point, node.dim), d = coord(node.data->pos, node.dim);
	if(node.left >= 0 && p <= d + radius) search(node.left, values, point, radius);
	if(node.right >= 0 && p >= d - radius) search(node.right, values, point, radius);
}

void KDTree::rangeSearch(std::vector<Match*>& values, const Vec2f& point, const float radius) const {
	if(!nodes.empty()) search(0, values, point, radius);
}

bool kMeans(const Vec3i& contrast, const Vec3i* values, const size_t n, Vec3i& c1, Vec3i& c2) {
	if(n < 2) return false; // kmeans.cpp:20-21

	float inGroupDiff = INFINITY, outGroupDiff = INFINITY; // :23-33
	for(size_t i = 0; i < n; i++) {
		outGroupDiff = std::min(outGroupDiff, (float)sqDiff(values[i], contrast));
		for(size_t j = i + 1; j < n; j++) inGroupDiff = std::min(inGroupDiff, (float)sqDiff(values[j], values[i]));
	}
	if(inGroupDiff > outGroupDiff) return false; // :35-36

	const Vec3i c1backup = c1, c2backup = c2;
	auto nearest = [&](const Vec3i& c) { // std::min_element: the first of equally near values (:43-44)
		size_t best = 0;
		for(size_t i = 1; i < n; i++)
			if(sqDiff(values[i], c) < sqDiff(values[best], c)) best = i;
		return values[best];
	};
	const Vec3i seed1 = nearest(c1), seed2 = nearest(c2);
	c1 = seed1;
	c2 = seed2;
	if(c1 == c2) { // :45-49
		c1 = c1backup;
		c2 = c2backup;
		return false;
	}

	Vec3i oldC1 = c2, oldC2 = c1; // :51-52 (crossed on purpose: guarantees the first iteration)
	while(oldC1 != c1 && oldC2 != c2) { // :55-81: stops as soon as EITHER centre is unchanged
		Vec3i sum1, sum2;
		int n1 = 0, n2 = 0;
		for(size_t k = 0; k < n; k++) {
			const Vec3i& v = values[k];
			if(sqDiff(v, c1) < sqDiff(v, c2)) {
				sum1 = {sum1.x + v.x, sum1.y + v.y, sum1.z + v.z};
				n1++;
			} else {
				sum2 = {sum2.x + v.x, sum2.y + v.y, sum2.z + v.z};
				n2++;
			}
		}
		if(n1 == 0 || n2 == 0) {
			c1 = c1backup;
			c2 = c2backup;
			return false;
		}
		oldC1 = c1;
		oldC2 = c2;
		c1 = {sum1.x / n1, sum1.y / n1, sum1.z / n1};
		c2 = {sum2.x / n2, sum2.y / n2, sum2.z / n2};
	}

	// :83-87 -- Eigen's norm() of an INTEGER vector returns an int: sqrt is taken in double and truncated before the float cast
	const int centreDistance = (int)std::sqrt((double)sqDiff(c1, c2));
	if((float)centreDistance < sqrtf(outGroupDiff) / 2.0f) {
		c1 = c1backup;
		c2 = c2backup;
		return false;
	}
	return true;
}

} // namespace vp
