// Shared between feeds_main.cpp (argument parsing, the one-GPU loop) and feeds_multi.cpp (vp_feeds --gpus N).
#pragma once

#include "vp_hip.h"

struct MultiOptions {
	int gpus = 1, feeds = 128, steps = 20, warmup = 3;
	int verify = 0, local = 0, detect = 0, threads = 16, cap = 192, full = 0;
	int same_device = 0; // TEST-ONLY: all `gpus` ranks on device 0, the gather through vp_gather_create_loopback (no RCCL) -- the N > 1 host pipeline on a one-GPU box
	const char* dump = nullptr;
};
int vp_feeds_run_multi(const MultiOptions& o);

void vp_feeds_cfg2_params(VpFrameParams* p); // cfg2: kernel scalars assigned directly
