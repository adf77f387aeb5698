// Whole-frame entry point: Resources::raw2quad (Resources.cpp:138-143) + Resources::rgba2blobCenter (:151-164) + the
// blobList launch and counter protocol of main.cpp:283-289, for `batch` frames on one stream with no host sync inside.
//
// Hot path = three launches groups: [raw2quad+resampling fused] -> [gradientDot+SAT] -> [circle + blobList]; the four
// quad planes, gradDot and blobCenter are only written when the caller asks for them as debug / stream views.
#include "vp_common.h"

#include <cstring>

int vp_resample_raw_mapped_impl(VpContext* ctx, int stream, const uint8_t* raw, int layout, int qw, int qh, const float* map, int sampler_mode,
		int fw, int fh, int batch, uint8_t* flat, const VpStrips& strips);
int vp_resample_impl(VpContext* ctx, int stream, const uint8_t* src, bool from_raw, int layout, int qw, int qh, const VpCameraModel* model,
		float max_robot_height, float field_scale, float off_x, float off_y, int sampler_mode, int fw, int fh, int batch, uint8_t* flat, const VpStrips& strips);

extern "C" int vp_frame(VpContext* ctx, int stream, const VpFrameParams* p, const uint8_t* raw, int batch, const VpFrameBuffers* b) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(p && raw && b, "vp_frame: null pointer");
	VP_REQUIRE(b->flat && b->sat && b->matches && b->counter, "vp_frame: flat, sat, matches and counter buffers are required");
	// halo strips for the single-pass SAT: written by whichever resampling kernel produces flat (6 % of flat, saves the SAT
	// kernel a 128-byte line per row and side of every tile)
	VpStrips strips = vp_make_strips(nullptr, 64, 1, 1);
	const size_t strip_bytes = p->sat_variant == VP_SAT_TILED ? vp_halo_strip_bytes(p->fw, p->fh, p->grad_offset) : 0;
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
			VP_TRY(vp_resample_map(ctx, stream, &p->model, p->max_robot_height, p->field_scale, p->off_x, p->off_y, p->fw, p->fh, static_cast<float*>(map)));
			memcpy(ctx->map_key[stream], key, sizeof(key));
		}
		VP_TRY(vp_resample_raw_mapped_impl(ctx, stream, raw, p->layout, p->qw, p->qh, static_cast<const float*>(map), p->sampler_mode, p->fw, p->fh, batch, b->flat, strips));
	} else {
		VP_TRY(vp_resample_impl(ctx, stream, raw, true, p->layout, p->qw, p->qh, &p->model, p->max_robot_height, p->field_scale, p->off_x, p->off_y, p->sampler_mode, p->fw, p->fh, batch, b->flat, strips));
	}
	VP_TRY(vp_graddot_sat_strips(ctx, stream, b->flat, strips.base, p->fw, p->fh, p->grad_offset, batch, p->sat_variant, b->sat, b->graddot));
	return vp_blobcenter_list(ctx, stream, b->flat, b->sat, p->fw, p->fh, batch, p->circle_radius, p->circ_threshold, p->min_score, p->list_radius, p->max_matches, p->order,
			b->matches, b->idx, b->counter, b->blobcenter);
}
