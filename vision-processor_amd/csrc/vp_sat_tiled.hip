// Single-pass LDS-tiled SAT (placeholder until the look-back kernel lands): refuses loudly, never falls back.
#include "vp_common.h"

int vp_sat_tiled(VpContext*, int, const void*, bool, int, int, int, int, int32_t*, float*) {
	vp_set_error("VP_SAT_TILED is not built in this revision; use VP_SAT_TWOPASS");
	return VP_ERR_INVALID;
}
