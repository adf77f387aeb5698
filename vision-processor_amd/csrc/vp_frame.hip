// Whole-frame entry point: Resources::raw2quad (Resources.cpp:138-143) + Resources::rgba2blobCenter (:151-164) + the
// blobList launch and counter protocol of main.cpp:283-289, for `batch` frames on one stream with no host sync inside.
//
// Hot path = three launches groups: [raw2quad+resampling fused] -> [gradientDot+SAT] -> [circle + blobList]; the four
// quad planes, gradDot and blobCenter are only written when the caller asks for them as debug / stream views.
#include "vp_common.h"

#include <cstring>
#include <vector>

int vp_resample_raw_mapped_impl(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, const float* map, int sampler_mode,
		int fw, int fh, int batch, uint8_t* flat, const VpStrips& strips);
int vp_resample_impl(VpContext* ctx, int stream, const uint8_t* src, bool from_raw, int layout, int qw, int qh, const VpCameraModel* model,
		float max_robot_height, float field_scale, float off_x, float off_y, int sampler_mode, int fw, int fh, int batch, uint8_t* flat, const VpStrips& strips);
int vp_sat_common(VpContext* ctx, int stream, const void* in, bool fused, int w, int h, int offset, int batch, int variant, int32_t* sat, float* gd, const char* name, const uint32_t* strips,
		int32_t* zero3, bool* zeroed);
int vp_flat_bloblist_impl(VpContext* ctx, int stream, const uint8_t* flat, int w, int h, int batch, int grad_offset, int circle_radius, float thr, float min_score, int radius,
		int max_matches, int order, VpMatch* matches, uint32_t* idx, int32_t* counter);
bool vp_flat_walk_supported(int r, int o);
int vp_list_from_blobcenter(VpContext* ctx, int stream, const uint8_t* flat, const float* blobcenter, const int32_t* sat, int circle_radius, float* bc_out,
		int w, int h, int batch, float thr, float min_score, int radius, int max_matches, int order, VpMatch* matches, uint32_t* idx, int32_t* counter, bool counters_zeroed);

static int frame_body(VpContext* ctx, int stream, const VpFrameParams* p, const uint8_t* raw, int batch, const VpFrameBuffers* b);

// ---- vp_ctx_tune "frame_graph": the launches of one frame as a captured hipGraph -----------------------------------------------------------
// Nothing that changes from one vp_frame call to the next is a kernel argument (the single-pass SAT tells its launches apart by their tickets,
// vp_sat_tiled.hip sat_launch_state; the list counters clean themselves), so the launch sequence of a (stream, parameters, batch, buffers) tuple is
// captured the second time it is seen and replayed from then on.  Host-side state the captured launches rely on is re-checked before every
// replay -- workspace generation (a grown workspace moves), the geometry the stream's coordinate map was built for, the SAT shape the ticket
// counters were advanced with, the look-back timeout flag, the ticket wrap -- and any mismatch takes the ordinary path, which repairs it.
static void graph_key(int stream, const VpFrameParams* p, const uint8_t* raw, int batch, const VpFrameBuffers* b, const VpTuning& tune, std::vector<uint8_t>& key) {
	key.resize(sizeof(int) * 2 + sizeof(*p) + sizeof(raw) + sizeof(*b) + sizeof(tune));
	uint8_t* k = key.data();
	memcpy(k, &stream, sizeof(int)); k += sizeof(int);
	memcpy(k, &batch, sizeof(int)); k += sizeof(int);
	memcpy(k, p, sizeof(*p)); k += sizeof(*p);
	memcpy(k, &raw, sizeof(raw)); k += sizeof(raw);
	memcpy(k, b, sizeof(*b)); k += sizeof(*b);
	memcpy(k, &tune, sizeof(tune));
}

static void graph_drop(VpFrameGraph& g) {
	if(g.exec) (void)hipGraphExecDestroy(g.exec);
	if(g.graph) (void)hipGraphDestroy(g.graph);
	g.exec = nullptr;
	g.graph = nullptr;
}

void vp_frame_graphs_clear(VpContext* ctx) {
	for(VpFrameGraph& g : ctx->graphs) graph_drop(g);
	ctx->graphs.clear();
}

// host-side state of a stream that a vp_frame call advances while it enqueues: a capture that is thrown away must put all of it back, because
// none of the launches it recorded will ever run
struct StreamState {
	unsigned char map_key[96];
	uint32_t sat_epoch, sat_tickets;
	uint64_t sat_shape;
};
static StreamState stream_state_save(const VpContext* ctx, int stream) {
	StreamState st;
	memcpy(st.map_key, ctx->map_key[stream], sizeof(st.map_key));
	st.sat_epoch = ctx->sat_epoch[stream];
	st.sat_tickets = ctx->sat_tickets[stream];
	st.sat_shape = ctx->sat_shape[stream];
	return st;
}
static void stream_state_restore(VpContext* ctx, int stream, const StreamState& st) {
	memcpy(ctx->map_key[stream], st.map_key, sizeof(st.map_key));
	ctx->sat_epoch[stream] = st.sat_epoch;
	ctx->sat_tickets[stream] = st.sat_tickets;
	ctx->sat_shape[stream] = st.sat_shape;
}

// what the stream's host-side state must look like for the launches of this tuple to run without a repair (recorded after every ordinary call of the tuple)
static void graph_note_needs(VpContext* ctx, int stream, VpFrameGraph* g) {
	g->ws_generation = ctx->ws_generation[stream];
	memcpy(g->map_key, ctx->map_key[stream], sizeof(g->map_key));
	g->sat_shape = ctx->sat_shape[stream];
	g->sat_tickets = ctx->sat_tickets[stream];
	g->noted = true;
}

static int frame_ordinary(VpContext* ctx, int stream, const VpFrameParams* p, const uint8_t* raw, int batch, const VpFrameBuffers* b, VpFrameGraph* g) {
	const int rc = frame_body(ctx, stream, p, raw, batch, b);
	if(rc == VP_OK) graph_note_needs(ctx, stream, g);
	return rc;
}

static int frame_graphed(VpContext* ctx, int stream, const VpFrameParams* p, const uint8_t* raw, int batch, const VpFrameBuffers* b) {
	VP_CHECK_HIP(hipSetDevice(ctx->device)); // capture and replay act on this context's stream, whatever device the calling thread used last
	std::vector<uint8_t> key;
	graph_key(stream, p, raw, batch, b, ctx->tune, key);
	VpFrameGraph* g = nullptr;
	for(VpFrameGraph& c : ctx->graphs)
		if(c.key == key) g = &c;
	if(!g) { // first sighting: the ordinary path (sizes the workspaces, builds the coordinate map, resets the SAT state if the shape is new)
		if(ctx->graphs.size() >= 64) vp_frame_graphs_clear(ctx);
		ctx->graphs.emplace_back();
		g = &ctx->graphs.back();
		g->key = key;
		return frame_ordinary(ctx, stream, p, raw, batch, b, g);
	}
	// the single-pass SAT state (ticket counters, tile order, carry granules, launch count) is used by VP_SAT_TILED and by VP_SAT_LOCAL calls whose
	// (radius, offset) pair has no walker: those build a private table with the same kernel (vp_tail.hip, workspace slot 5)
	const bool tiled = p->sat_variant == VP_SAT_TILED || (p->sat_variant == VP_SAT_LOCAL && !vp_flat_walk_supported(p->circle_radius, p->grad_offset));
	int* flag = ctx->host_flag[stream];
	const bool poisoned = flag && *reinterpret_cast<volatile int*>(flag) != 0;
	// does the stream still look the way this tuple left it?  (another shape resets the SAT state, another geometry rebuilds the map, a grown workspace moves)
	const bool state_ok = g->noted && g->ws_generation == ctx->ws_generation[stream] && memcmp(g->map_key, ctx->map_key[stream], sizeof(g->map_key)) == 0 &&
			(!tiled || (g->sat_shape == ctx->sat_shape[stream] && g->sat_tickets == ctx->sat_tickets[stream]));
	if(g->exec) {
		const bool near_wrap = tiled && (((uint64_t)ctx->sat_epoch[stream] + 4) * g->sat_tickets >= (uint64_t(1) << 32) || ctx->sat_epoch[stream] >= (1u << 30) - 4);
		if(state_ok && !poisoned && !near_wrap) {
			if(tiled) ++ctx->sat_epoch[stream]; // the replayed SAT launch draws its tickets like any other
			VP_CHECK_HIP(hipGraphLaunch(g->exec, ctx->streams[stream]));
			ctx->graph_stats[1]++;
			return VP_OK;
		}
		graph_drop(*g); // something the captured launches rely on has moved: ordinary path now (it repairs), capture again next time
		ctx->graph_stats[3]++;
		return frame_ordinary(ctx, stream, p, raw, batch, b, g);
	}
	// Pre-flight (frame_graph = 1): a stream whose state belongs to another tuple needs a repair that cannot be captured -- take the ordinary path and
	// capture the next time the tuple follows itself.  frame_graph = 2 skips the pre-flight: every repeated tuple attempts a capture, which exercises the
	// refusal path below (tests).
	if(g->refused >= 3 || poisoned || (ctx->tune.frame_graph == 1 && !state_ok)) return frame_ordinary(ctx, stream, p, raw, batch, b, g);
	hipStream_t s = ctx->streams[stream];
	if(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
		(void)hipGetLastError();
		g->refused = 3; // this stream cannot be captured at all
		return frame_ordinary(ctx, stream, p, raw, batch, b, g);
	}
	// A reset of the SAT state, a rebuild of the coordinate map or a growing workspace inside the capture is refused by those functions
	// (ctx->capturing) BEFORE they change anything; what frame_body advanced up to that point is put back here, because none of the recorded launches will run.
	ctx->capturing[stream] = true;
	const StreamState before = stream_state_save(ctx, stream);
	const int rc = frame_body(ctx, stream, p, raw, batch, b);
	ctx->capturing[stream] = false;
	hipGraph_t graph = nullptr;
	const hipError_t e = hipStreamEndCapture(s, &graph);
	if(rc != VP_OK || e != hipSuccess || !graph || hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) != hipSuccess) {
		(void)hipGetLastError();
		if(graph) (void)hipGraphDestroy(graph);
		g->exec = nullptr;
		// the runtime could not capture / instantiate (rc == VP_OK): a property of the tuple, three strikes.  A body that failed inside the capture -- a
		// refused repair, which the ordinary call below makes good, or a genuine error that will fail there too -- is counted separately and only while it
		// repeats: eight in a row (frame_graph = 2 alternating shapes, a persistent launch error) and the tuple stops paying for two bodies per call.
		if(rc == VP_OK) g->refused++;
		else if(++g->refused_bodies >= 8) g->refused = 3;
		ctx->graph_stats[2]++;
		stream_state_restore(ctx, stream, before);
		const int rc2 = frame_ordinary(ctx, stream, p, raw, batch, b, g); // sees the stream as it was and repairs it
		if(rc2 == VP_OK) vp_clear_error(); // the refusal's message ("... inside a graph capture") is not this call's outcome
		return rc2;
	}
	g->refused_bodies = 0;
	g->graph = graph;
	graph_note_needs(ctx, stream, g);
	VP_CHECK_HIP(hipGraphLaunch(g->exec, s)); // (the capture counted this launch's SAT tickets already)
	ctx->graph_stats[0]++;
	return VP_OK;
}

// captures made, replays launched, captures refused, graphs dropped because the stream's state had moved -- since the context was created
extern "C" int vp_frame_graph_stats(VpContext* ctx, int64_t out[4]) {
	VP_REQUIRE(ctx && out, "vp_frame_graph_stats: bad arguments");
	for(int i = 0; i < 4; i++) out[i] = ctx->graph_stats[i];
	return VP_OK;
}

extern "C" int vp_frame(VpContext* ctx, int stream, const VpFrameParams* p, const uint8_t* raw, int batch, const VpFrameBuffers* b) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(p && raw && b, "vp_frame: null pointer");
	// stage events are per-launch objects and the debug views are occasional: those calls take the ordinary path
	if(ctx->tune.frame_graph && !ctx->events_on && !b->planes && !b->graddot && !b->blobcenter && ctx->streams[stream] != nullptr) return frame_graphed(ctx, stream, p, raw, batch, b);
	return frame_body(ctx, stream, p, raw, batch, b);
}

static int frame_body(VpContext* ctx, int stream, const VpFrameParams* p, const uint8_t* raw, int batch, const VpFrameBuffers* b) {
	VP_REQUIRE(p && raw && b, "vp_frame: null pointer");
	const bool local = p->sat_variant == VP_SAT_LOCAL;
	VP_REQUIRE(b->flat && (b->sat || local) && b->matches && b->counter, "vp_frame: flat, sat, matches and counter buffers are required");
	VP_REQUIRE(!local || (!b->sat && !b->graddot && !b->blobcenter), "vp_frame: VP_SAT_LOCAL builds no global table: sat, graddot and blobcenter must be NULL");
	// every argument is checked before a workspace is sized from it or a kernel is enqueued: a refused call leaves the caller's buffers untouched
	VP_REQUIRE(p->qw > 0 && p->qh > 0 && p->fw > 0 && p->fh > 0 && batch > 0 && batch <= 65535 && p->grad_offset >= 0 && p->list_radius >= 0 && p->circle_radius >= 0 && p->max_matches > 0 &&
			(size_t)p->fw * p->fh < (size_t(1) << 31), "vp_frame: bad shape");
	VP_REQUIRE(p->layout >= VP_LAYOUT_RGGB && p->layout <= VP_LAYOUT_BGR, "vp_frame: bad layout %d", p->layout);
	VP_REQUIRE(p->sampler_mode == VP_SAMPLER_NEAREST || p->sampler_mode == VP_SAMPLER_BILINEAR_RTZ, "vp_frame: bad sampler mode %d", p->sampler_mode);
	VP_REQUIRE(p->sat_variant >= VP_SAT_TILED && p->sat_variant <= VP_SAT_LOCAL, "vp_frame: bad SAT variant %d", p->sat_variant);
	VP_REQUIRE(p->order == VP_ORDER_ATOMIC || p->order == VP_ORDER_RASTER, "vp_frame: bad order");
	VP_REQUIRE((reinterpret_cast<uintptr_t>(b->matches) & 1) == 0, "vp_frame: matches must be 2-byte aligned");
	// the workspaces below are allocated on the CURRENT device: a caller driving several contexts from one thread (host/feeds_multi.cpp) must land on ours
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	// halo strips for the single-pass SAT: written by whichever resampling kernel produces flat (6 % of flat, saves the SAT
	// kernel a 128-byte line per row and side of every tile)
	VpStrips strips = vp_make_strips(nullptr, 64, 1, 1);
	const size_t strip_bytes = p->sat_variant == VP_SAT_TILED && ctx->tune.sat_halo_strips ? vp_halo_strip_bytes(p->fw, p->fh, p->grad_offset) : 0;
	if(strip_bytes) {
		void* sp = nullptr;
		VP_TRY(vp_workspace(ctx, stream, 3, strip_bytes * (size_t)batch, &sp));
		strips = vp_make_strips(static_cast<uint32_t*>(sp), p->fw, p->fh, p->grad_offset);
	}
	if(b->planes) {
		VP_TRY(vp_raw2quad(ctx, stream, raw, p->layout, p->qw, p->qh, batch, b->planes));
		VP_TRY(vp_resample_impl(ctx, stream, b->planes, false, p->layout, p->qw, p->qh, &p->model, p->max_robot_height, p->field_scale, p->off_x, p->off_y, p->sampler_mode, p->fw, p->fh, batch, b->flat, strips));
	} else if(p->use_coord_map) {
		// the geometry changes once per calibration, not per frame: keep field2image() of every flat pixel in a per-stream map
		unsigned char key[96] = {0};
		memcpy(key, &p->model, sizeof(p->model));
		const float sc[4] = {p->max_robot_height, p->field_scale, p->off_x, p->off_y};
		memcpy(key + 72, sc, sizeof(sc));
		const int32_t sz[2] = {p->fw, p->fh};
		memcpy(key + 88, sz, sizeof(sz));
		void* map = nullptr;
		bool grew = false;
		VP_TRY(vp_workspace(ctx, stream, 2, sizeof(float) * 2 * (size_t)p->fw * p->fh, &map, &grew));
		if(grew || memcmp(key, ctx->map_key[stream], sizeof(key)) != 0) {
			// a map built inside a capture would be rebuilt by every replay, and a discarded capture would leave map_key claiming a map that was never built
			VP_REQUIRE(!ctx->capturing[stream], "the coordinate map would have to be rebuilt inside a graph capture");
			VP_TRY(vp_resample_map(ctx, stream, &p->model, p->max_robot_height, p->field_scale, p->off_x, p->off_y, p->fw, p->fh, static_cast<float*>(map)));
			memcpy(ctx->map_key[stream], key, sizeof(key));
		}
		VP_TRY(vp_resample_raw_mapped_impl(ctx, stream, raw, p->layout, p->qw, p->qh, static_cast<const float*>(map), p->sampler_mode, p->fw, p->fh, batch, b->flat, strips));
	} else {
		VP_TRY(vp_resample_impl(ctx, stream, raw, true, p->layout, p->qw, p->qh, &p->model, p->max_robot_height, p->field_scale, p->off_x, p->off_y, p->sampler_mode, p->fw, p->fh, batch, b->flat, strips));
	}
	if(local) { // resampling -> [gradientDot + local table + circle + blobList]
		VpStageTimer t(ctx, stream, "gradientDot+satBlobCenter+blobList (local)");
		return vp_flat_bloblist_impl(ctx, stream, b->flat, p->fw, p->fh, batch, p->grad_offset, p->circle_radius, p->circ_threshold, p->min_score, p->list_radius, p->max_matches, p->order,
				b->matches, b->idx, b->counter);
	}
	// three launches per call: [raw2quad+resampling] -> [gradientDot+SAT, which also clears the list counters] -> [circle+detect] -> [emit (+ sort)]
	bool zeroed = false;
	VP_TRY(vp_sat_common(ctx, stream, b->flat, true, p->fw, p->fh, p->grad_offset, batch, p->sat_variant, b->sat, b->graddot, "gradientDot+sat", strips.base, b->counter, &zeroed));
	VpStageTimer t(ctx, stream, "satBlobCenter+blobList");
	return vp_list_from_blobcenter(ctx, stream, b->flat, nullptr, b->sat, p->circle_radius, b->blobcenter, p->fw, p->fh, batch, p->circ_threshold, p->min_score, p->list_radius, p->max_matches, p->order,
			b->matches, b->idx, b->counter, zeroed);
}
