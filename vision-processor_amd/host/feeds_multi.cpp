// vp_feeds --gpus N: the multi-GPU form of the feed pipeline, ONE process on the C ABIs alone (include/vp_hip.h, vp_multi.h).
// Feeds are sharded over the devices (feed i of device g has the seed feed_seed(g, i) of vision-processor_amd/feeds.py); every
// device runs vp_frame for its own feeds on its own streams, writing counters / records / indices straight into its record
// block; after each step ONE grouped RCCL all-gather (vp_gather_start) hands every device every device's block, on the gather
// streams, while the next step's kernels run.  This is what replaces the reference's per-camera processes exchanging packets
// over UDP multicast (src/udpsocket.cpp:94-100, 147-163; one loop per camera: src/main.cpp:262-317).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vp_hip.h"
#include "vp_multi.h"

#define CK(call) \
	do { \
		const int rc_ = (call); \
		if(rc_ != VP_OK) { \
			fprintf(stderr, "[FATAL] %s failed (%d): %s\n", #call, rc_, vp_last_error()); \
			exit(1); \
		} \
	} while(0)

void vp_feeds_cfg2_params(VpFrameParams* p); // feeds_main.cpp

int vp_feeds_run_multi(int gpus, int feeds, int steps, int warmup, int verify, int local) {
	const int visible = vp_device_count();
	if(gpus < 1 || gpus > visible) {
		fprintf(stderr, "[FATAL] --gpus %d but %d HIP device(s) are visible\n", gpus, visible);
		return 1;
	}
	VpFrameParams p;
	vp_feeds_cfg2_params(&p);
	if(local) p.sat_variant = VP_SAT_LOCAL; // no global table: resampling + ONE kernel per step and device
	const int qw = p.qw, qh = p.qh, mm = p.max_matches;
	const size_t rawBytes = (size_t)4 * qw * qh, npx = (size_t)p.fw * p.fh;
	size_t offC = 0, offM = 0, offI = 0;
	const size_t recBytes = vp_record_layout(feeds, mm, &offC, &offM, &offI);

	std::vector<int> devs(gpus);
	for(int g = 0; g < gpus; g++) devs[g] = g;
	VpGather* gather = nullptr;
	CK(vp_gather_create(gpus, devs.data(), recBytes, 2, &gather));
	std::vector<VpContext*> ctx(gpus, nullptr);
	std::vector<uint8_t*> dRaw(gpus, nullptr);
	std::vector<VpFrameBuffers> buf((size_t)gpus * 2);
	{
		std::vector<uint8_t> frame(rawBytes);
		for(int g = 0; g < gpus; g++) {
			CK(vp_ctx_create(g, 2, &ctx[g]));
			CK(vp_malloc_device(ctx[g], rawBytes * feeds, (void**)&dRaw[g]));
			for(int i = 0; i < feeds; i++) { // feeds.py feed_seed(rank, frame) = rank * 100000 + frame
				CK(vp_synth_frame(&p.model, p.layout, qw, qh, p.max_robot_height, 9000.f, 6000.f, 0x5EED0000u + (uint32_t)(g * 100000 + i), 16, 8, 3.0f, frame.data()));
				CK(vp_copy_h2d(ctx[g], 0, dRaw[g] + rawBytes * i, frame.data(), rawBytes));
				CK(vp_sync(ctx[g], 0));
			}
			for(int s = 0; s < 2; s++) {
				VpFrameBuffers& b = buf[(size_t)g * 2 + s];
				memset(&b, 0, sizeof(b));
				CK(vp_malloc_device(ctx[g], npx * 4 * feeds, (void**)&b.flat));
				if(!local) CK(vp_malloc_device(ctx[g], npx * 4 * feeds, (void**)&b.sat));
				uint8_t* rec = static_cast<uint8_t*>(vp_gather_send(gather, g, s)); // vp_frame writes its outputs into the record block itself
				b.counter = reinterpret_cast<int32_t*>(rec + offC);
				b.matches = reinterpret_cast<VpMatch*>(rec + offM);
				b.idx = reinterpret_cast<uint32_t*>(rec + offI);
			}
		}
	}
	std::vector<int> streams(gpus, 0);
	auto step = [&](int k) {
		const int s = k & 1;
		CK(vp_gather_wait(gather, s)); // the gather issued two steps ago has released this slot's send block
		for(int g = 0; g < gpus; g++) {
			CK(vp_frame(ctx[g], s, &p, dRaw[g], feeds, &buf[(size_t)g * 2 + s]));
			streams[g] = s;
		}
		CK(vp_gather_start(gather, s, ctx.data(), streams.data()));
	};
	int k = 0;
	for(; k < warmup; k++) step(k);
	CK(vp_gather_wait(gather, 0));
	CK(vp_gather_wait(gather, 1));
	for(int g = 0; g < gpus; g++) CK(vp_sync(ctx[g], -1));
	const auto t0 = std::chrono::steady_clock::now();
	for(int j = 0; j < steps; j++, k++) step(k);
	CK(vp_gather_wait(gather, 0)); // every step's gather has completed inside the timed region
	CK(vp_gather_wait(gather, 1));
	for(int g = 0; g < gpus; g++) CK(vp_sync(ctx[g], -1));
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

	// what the last step gathered: every device holds every device's block, byte for byte what the owners produced
	const int last = (k - 1) & 1;
	int verified = -1;
	std::vector<uint8_t> own((size_t)gpus * recBytes), got((size_t)gpus * recBytes);
	for(int g = 0; g < gpus; g++) CK(vp_copy_d2h_sync(ctx[g], 0, own.data() + (size_t)g * recBytes, vp_gather_send(gather, g, last), recBytes));
	if(verify) {
		verified = 1;
		for(int g = 0; g < gpus; g++) {
			CK(vp_copy_d2h_sync(ctx[g], 0, got.data(), vp_gather_recv(gather, g, last), (size_t)gpus * recBytes));
			if(memcmp(got.data(), own.data(), (size_t)gpus * recBytes) != 0) {
				fprintf(stderr, "[FATAL] device %d received records that differ from what their owners produced\n", g);
				verified = 0;
			}
		}
	}
	long blobs = 0;
	printf("{\"program\": \"vp_feeds\", \"gpus\": %d, \"feeds_per_gpu\": %d, \"steps\": %d, \"frames_per_s\": %.1f, \"ms_per_step\": %.4f, \"record_bytes_per_gpu\": %zu, \"gather\": \"RCCL ncclAllGather, one grouped call per step, %d rank(s) in one process\", \"sat_variant\": \"%s\"",
			gpus, feeds, steps, (double)gpus * feeds * steps / dt, 1e3 * dt / steps, recBytes, gpus, local ? "local" : "tiled");
	printf(", \"counters\": [");
	for(int g = 0; g < gpus; g++) {
		const int32_t* c = reinterpret_cast<const int32_t*>(own.data() + (size_t)g * recBytes + offC);
		printf("%s[", g ? ", " : "");
		for(int i = 0; i < feeds; i++) {
			printf("%s[%d, %d, %d]", i ? ", " : "", c[3 * i], c[3 * i + 1], c[3 * i + 2]);
			blobs += c[3 * i] < mm ? c[3 * i] : mm;
		}
		printf("]");
	}
	printf("], \"blobs_per_frame\": %.2f", blobs / ((double)gpus * feeds));
	if(verify) printf(", \"gather_verified\": %s", verified == 1 ? "true" : "false");
	printf("}\n");

	for(int g = 0; g < gpus; g++) {
		for(int s = 0; s < 2; s++) {
			vp_free_device(ctx[g], buf[(size_t)g * 2 + s].flat);
			vp_free_device(ctx[g], buf[(size_t)g * 2 + s].sat);
		}
		vp_free_device(ctx[g], dRaw[g]);
	}
	vp_gather_destroy(gather);
	for(int g = 0; g < gpus; g++) vp_ctx_destroy(ctx[g]);
	return verified == 0 ? 1 : 0;
}
