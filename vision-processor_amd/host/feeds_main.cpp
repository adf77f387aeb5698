// vp_feeds: `feeds` independent camera feeds through the whole chain on one GPU, in C++ on the C ABI alone (no Python, no PyTorch):
//   vp_frame (one launch group per step) -> vp_records_compact -> read-back of the compact slots -> per feed: hypotheses, SSL_DetectionFrame,
//   tracker, SSL_WrapperPacket (vph_detect_batch_slots on host threads), the CPU stage of step k running while the GPU works on step k + 1.
// This is the loop of src/main.cpp:262-372 for many cameras at once: what a deployment that puts several cameras on one MI355X runs.
//
// usage: vp_feeds [feeds=128] [steps=20] [warmup=3] [host_threads=16] [detect=1] [readback=2: completion word | 1: host block + vp_sync | 0: device block + copy + vp_sync] [stage_events=0]
//        vp_feeds --gpus N [feeds per GPU=128] [steps=20] [warmup=3] [--detect] [--threads T=16] [--cap C=192] [--full-records] [--local] [--verify] [--dump FILE]
//            feeds_multi.cpp: N devices, one driver thread each, one RCCL all-gather per step of the encoded SSL_WrapperPackets (--detect: the CPU stage
//            runs per device on T / N host threads) or of the compact blob records (--cap records per feed; --full-records: the fixed-capacity arrays
//            as vp_frame writes them); --local = VP_SAT_LOCAL; --verify: every device received every owner's block; --dump: the last device's receive area;
//            --same-device (TEST-ONLY): all N ranks on device 0 with a loop-back gather (device-to-device copies, no RCCL) so that the N > 1 pipeline
//            -- driver threads, rendezvous, last-arriver gather, slot reuse, drain -- can run on a one-GPU box
// Scene and geometry: SURVEY.md 8(d) cfg2 (2048x1536 RGGB, nadir camera 5 m over a 9.6 x 7.2 m patch, 16 robots, 8 balls per feed).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vp_hip.h"
#include "vp_host.h"
#include "feeds.h"

#define CK(call) \
	do { \
		const int rc_ = (call); \
		if(rc_ != VP_OK) { \
			fprintf(stderr, "[FATAL] %s failed (%d): %s\n", #call, rc_, vp_last_error()); \
			exit(1); \
		} \
	} while(0)


// cfg2: kernel scalars assigned directly
void vp_feeds_cfg2_params(VpFrameParams* out) {
	const int qw = 1024, qh = 768;
	const float fieldScale = 4.6875f;
	VpFrameParams p;
	memset(&p, 0, sizeof(p));
	p.layout = VP_LAYOUT_RGGB;
	p.qw = qw;
	p.qh = qh;
	vp_synth_camera(qw, qh, 5000.f, 180.f, 9600.f, 0.08f, &p.model);
	p.max_robot_height = 180.f;
	p.field_scale = fieldScale;
	p.off_x = -4800.f;
	p.off_y = -3600.f;
	p.fw = 2048;
	p.fh = 1536;
	p.grad_offset = (int)ceilf(25.0f / fieldScale) / 3; // Resources.cpp:160
	p.circle_radius = (int)ceilf(20.0f / fieldScale);  // :163
	p.list_radius = (int)floorf(20.0f / fieldScale);   // main.cpp:289
	p.circ_threshold = 15.0f;
	p.min_score = 0.0f;
	p.max_matches = 2000;
	p.sampler_mode = VP_SAMPLER_BILINEAR_RTZ;
	p.sat_variant = VP_SAT_TILED;
	p.order = VP_ORDER_RASTER;
	p.use_coord_map = 1;
	*out = p;
}

int main(int argc, char** argv) {
	if(argc > 2 && strcmp(argv[1], "--gpus") == 0) {
#ifdef VP_HAVE_MULTI
		MultiOptions o;
		o.gpus = atoi(argv[2]);
		int pos[3] = {128, 20, 3}, np = 0;
		for(int i = 3; i < argc; i++) {
			if(strcmp(argv[i], "--verify") == 0) o.verify = 1;
			else if(strcmp(argv[i], "--local") == 0) o.local = 1;
			else if(strcmp(argv[i], "--same-device") == 0) o.same_device = 1;
			else if(strcmp(argv[i], "--detect") == 0) o.detect = 1;
			else if(strcmp(argv[i], "--full-records") == 0) o.full = 1;
			else if(strcmp(argv[i], "--threads") == 0 && i + 1 < argc) o.threads = atoi(argv[++i]);
			else if(strcmp(argv[i], "--cap") == 0 && i + 1 < argc) o.cap = atoi(argv[++i]);
			else if(strcmp(argv[i], "--dump") == 0 && i + 1 < argc) o.dump = argv[++i];
			else if(argv[i][0] != '-' && np < 3) pos[np++] = atoi(argv[i]);
			else np = 99;
		}
		o.feeds = pos[0];
		o.steps = pos[1];
		o.warmup = pos[2];
		if(np > 3 || o.feeds < 1 || o.steps < 1 || o.warmup < 0 || o.threads < 1 || o.cap < 1) {
			fprintf(stderr, "usage: vp_feeds --gpus N [feeds per GPU] [steps] [warmup] [--detect] [--threads T] [--cap C] [--full-records] [--local] [--verify] [--dump FILE] [--same-device]\n");
			return 2;
		}
		return vp_feeds_run_multi(o);
#else
		fprintf(stderr, "[FATAL] vp_feeds was built without libvp_multi.so (no librccl on the build box): --gpus is not available\n");
		return 1;
#endif
	}
	const int feeds = argc > 1 ? atoi(argv[1]) : 128, steps = argc > 2 ? atoi(argv[2]) : 20, warmup = argc > 3 ? atoi(argv[3]) : 3;
	const int threads = argc > 4 ? atoi(argv[4]) : 16, detect = argc > 5 ? atoi(argv[5]) : 1;
	const int stageEvents = argc > 7 ? atoi(argv[7]) : 0; // 1: per-stage hipEvent times of the timed steps on stderr (printRuntimes)
	const int readback = argc > 6 ? atoi(argv[6]) : 2; // 2: the compact kernel stores the pinned host block AND a completion word the host polls (no vp_sync on the way); 1: host block, vp_sync; 0: device block + hipMemcpyAsync, vp_sync
	if(feeds < 1 || steps < 1 || warmup < 0 || threads < 1) {
		fprintf(stderr, "usage: vp_feeds [feeds] [steps] [warmup] [host_threads] [detect 0|1]\n");
		return 2;
	}
	VpFrameParams p;
	vp_feeds_cfg2_params(&p);
	const int qw = p.qw, qh = p.qh, fw = p.fw, fh = p.fh, maxBlobs = p.max_matches, robots = 16, balls = 8;
	const float fieldScale = p.field_scale, offX = p.off_x, offY = p.off_y, planeHeight = p.max_robot_height, fieldLen = 9000.f, fieldWid = 6000.f;

	VpContext* ctx = nullptr;
	CK(vp_ctx_create(0, 2, &ctx));
	CK(vp_ctx_tune(ctx, "placement_max_mib", -1)); // the program owns its GPU: the placement search may hold nine tenths of the free memory
	const size_t rawBytes = (size_t)4 * qw * qh;
	// synthetic feeds, resident in HBM before the clock starts
	uint8_t* dRaw = nullptr;
	CK(vp_malloc_device(ctx, rawBytes * feeds, (void**)&dRaw));
	{
		std::vector<uint8_t> frame(rawBytes);
		for(int i = 0; i < feeds; i++) {
			CK(vp_synth_frame(&p.model, p.layout, qw, qh, planeHeight, fieldLen, fieldWid, 0x5EED0000u + (uint32_t)i, robots, balls, 3.0f, frame.data()));
			CK(vp_copy_h2d(ctx, 0, dRaw + rawBytes * i, frame.data(), rawBytes));
			CK(vp_sync(ctx, 0));
		}
	}
	// two sets of per-step buffers: step k runs on stream k & 1 while the host reads step k - 1 back from the other set
	VpFrameBuffers buf[2];
	VpFrameSlab* slab[2];
	// what the host reads back per step is the compact form: counter[3], n and the n records of each feed (main.cpp:299-301), not the
	// fixed-capacity arrays (0.64 MB instead of 5.6 MB at 128 feeds)
	const int cap = 192, packetSlot = 4096;
	size_t slotBytes = 0;
	const size_t compactBytes = vp_record_layout_compact(feeds, cap, &slotBytes, nullptr, nullptr);
	uint8_t* dSlots[2];
	void* hSlots[2];
	for(int s = 0; s < 2; s++) {
		CK(vp_frame_buffers_create(ctx, &p, feeds, 0, &buf[s], &slab[s])); // allocated once, the table and flat from placement-checked blocks
		CK(vp_malloc_device(ctx, compactBytes, (void**)&dSlots[s]));
		CK(vp_malloc_host(ctx, compactBytes, &hSlots[s]));
	}
	uint32_t* doneWord = nullptr; // one completion word per buffer set, a cache line apart
	CK(vp_malloc_host(ctx, 256, (void**)&doneWord));
	memset(doneWord, 0, 256);
	// one detector per feed: own colour model, own tracker (main.cpp runs one process per camera)
	std::vector<VphDetector*> detectors;
	const float calib[11] = {p.model.f, p.model.p[0], p.model.p[1], p.model.d, -1.f, 0.f, 0.f, 0.f, -p.model.c[0], p.model.c[1], p.model.c[2]}; // nadir: t = R (-pos)
	const int32_t field[5] = {(int32_t)fieldLen, (int32_t)fieldWid, 300, -1, 10};
	if(detect)
		for(int i = 0; i < feeds; i++) {
			VphConfig cfg;
			vph_config_defaults(&cfg);
			cfg.cam_id = i;
			VphDetector* d = vph_detector_create(&cfg);
			if(!d || vph_detector_set_geometry_direct(d, calib, qw, qh, fieldScale, offX, offY, fw, fh, field, 21.5f, 90.f, nullptr, 0, nullptr, 0) != VP_OK) {
				fprintf(stderr, "[FATAL] detector %d\n", i);
				return 1;
			}
			detectors.push_back(d);
		}
	std::vector<uint8_t> packets((size_t)feeds * packetSlot); // one slot per feed: [int32 length | SSL_WrapperPacket]

	long blobs = 0, packetBytes = 0;
	double waitMs = 0, stageMs = 0, launchMs = 0;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	auto cpuStage = [&](int s, int frameNo) { // the read-back of step `frameNo` sits in host set s
		const auto w0 = now();
		if(readback == 2) CK(vp_wait_word(doneWord + 16 * s, (uint32_t)frameNo, 30.0)); // step frameNo - 1 stored frameNo (launch: k + 1)
		else CK(vp_sync(ctx, s));
		const auto w1 = now();
		waitMs += ms(w0, w1);
		const uint8_t* slots = static_cast<const uint8_t*>(hSlots[s]);
		for(int i = 0; i < feeds; i++) {
			int32_t head[4]; // counter[3], n
			memcpy(head, slots + (size_t)i * slotBytes, sizeof(head));
			blobs += head[3];
		}
		if(!detect) return;
		const double t = frameNo / 70.0;
		if(vph_detect_batch_slots(detectors.data(), feeds, slots, slotBytes, maxBlobs, (uint32_t)frameNo, t, t, t + 0.002, threads, packets.data(), packetSlot) != VP_OK) {
			fprintf(stderr, "[FATAL] the CPU stage refused the record slots (a frame has more than %d blobs?)\n", cap);
			exit(1);
		}
		for(int i = 0; i < feeds; i++) {
			int32_t len;
			memcpy(&len, packets.data() + (size_t)i * packetSlot, 4);
			packetBytes += len > 0 ? len : 0;
		}
		stageMs += ms(w1, now());
	};
	auto launch = [&](int k) {
		const auto l0 = now();
		const int s = k & 1;
		CK(vp_frame(ctx, s, &p, dRaw, feeds, &buf[s]));
		// the compact block is 1.3 MB per 256 feeds: the kernel stores it straight into the pinned host block (hipHostMalloc memory is device-visible) and its
		// last workgroup then stores the step number into the set's completion word, which the CPU stage polls instead of synchronising the stream.  Same rate as
		// device block + copy + vp_sync here (the step is bound by the CPU stage or by the kernels, profiles/r06_vp_feeds.jsonl); kept as the form without a
		// copy command and without a round trip through the command processor
		if(readback == 2) CK(vp_records_compact_signal(ctx, s, buf[s].counter, buf[s].matches, nullptr, feeds, maxBlobs, cap, static_cast<uint8_t*>(hSlots[s]), doneWord + 16 * s, (uint32_t)(k + 1)));
		else if(readback) CK(vp_records_compact(ctx, s, buf[s].counter, buf[s].matches, nullptr, feeds, maxBlobs, cap, static_cast<uint8_t*>(hSlots[s])));
		else {
			CK(vp_records_compact(ctx, s, buf[s].counter, buf[s].matches, nullptr, feeds, maxBlobs, cap, dSlots[s]));
			CK(vp_copy_d2h(ctx, s, hSlots[s], dSlots[s], compactBytes));
		}
		launchMs += ms(l0, now());
	};

	int k = 0;
	for(; k < warmup; k++) {
		launch(k);
		if(k > 0) cpuStage((k - 1) & 1, k);
	}
	if(k > 0) cpuStage((k - 1) & 1, k);
	CK(vp_sync(ctx, -1));
	blobs = packetBytes = 0;
	waitMs = stageMs = launchMs = 0;
	if(stageEvents) CK(vp_events_enable(ctx, 1));
	const auto t0 = std::chrono::steady_clock::now();
	for(int j = 0; j < steps; j++, k++) {
		launch(k);
		if(j > 0) cpuStage((k - 1) & 1, k); // steps are numbered from 1: step k's blobs while the GPU works on step k + 1
	}
	cpuStage((k - 1) & 1, k);
	CK(vp_sync(ctx, -1));
	const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

	if(stageEvents) {
		std::vector<VpStageTime> st(4096);
		const int n = vp_event_times(ctx, st.data(), (int)st.size());
		double sum = 0;
		for(int i = 0; i < n; i++) sum += st[i].ms;
		fprintf(stderr, "stage events: %d launches, %.3f ms of GPU time per step\n", n, sum / steps);
		for(int i = 0; i < n && i < 12; i++) fprintf(stderr, "  %-40s %.3f ms\n", st[i].stage, st[i].ms);
	}
	int robotsSeen = 0, ballsSeen = 0;
	if(detect)
		for(VphDetector* d : detectors) {
			VphFrameHeader h;
			vph_frame_header(d, &h);
			robotsSeen += h.n_robots_yellow + h.n_robots_blue;
			ballsSeen += h.n_balls;
		}
	printf("{\"program\": \"vp_feeds\", \"readback\": \"%s\", \"feeds\": %d, \"steps\": %d, \"host_threads\": %d, \"detect\": %d, \"frames_per_s\": %.1f, \"ms_per_step\": %.4f, \"blobs_per_frame\": %.2f", readback == 2 ? "kernel stores into pinned host memory + completion word" : (readback ? "kernel stores into pinned host memory, vp_sync" : "hipMemcpyAsync, vp_sync"), feeds,
			steps, threads, detect, feeds * (double)steps / dt, 1e3 * dt / steps, blobs / ((double)feeds * steps));
	if(detect)
		printf(", \"robots_per_frame\": %.3f, \"balls_per_frame\": %.3f, \"packet_bytes_per_frame\": %.1f, \"rendered\": {\"robots\": %d, \"balls\": %d}", robotsSeen / (double)feeds,
				ballsSeen / (double)feeds, packetBytes / ((double)feeds * steps), robots, balls);
	printf(", \"host_ms_per_step\": {\"launch\": %.3f, \"wait_for_gpu\": %.3f, \"cpu_stage\": %.3f}", launchMs / steps, waitMs / steps, stageMs / steps);
	{ // where the two buffer sets' tables landed (vp_malloc_device_placed): a search that ran out of budget is a slow draw, not the kernel
		VpPlacement pl[2];
		int exhausted = 0;
		for(int s = 0; s < 2; s++) {
			CK(vp_frame_buffers_placement(slab[s], &pl[s], nullptr));
			exhausted += pl[s].selected == 2 && pl[s].write_gbs < pl[s].fast_gbs;
		}
		printf(", \"placement\": {\"sat_write_gbs\": [%.0f, %.0f], \"candidates\": [%d, %d]", pl[0].write_gbs, pl[1].write_gbs, pl[0].candidates, pl[1].candidates);
		if(exhausted) {
			printf(", \"warning\": \"%d of 2 searches kept the best block seen although none reached the write rate of a fast block\"", exhausted);
			fprintf(stderr, "[WARN] placement search ran out of candidates, bytes or memory for %d of 2 buffer sets: expect the SAT pass 5-15 %% slower\n", exhausted);
		}
		printf("}");
	}
	printf("}\n");

	for(VphDetector* d : detectors) vph_detector_destroy(d);
	for(int s = 0; s < 2; s++) {
		vp_frame_buffers_destroy(ctx, slab[s]);
		vp_free_device(ctx, dSlots[s]);
		vp_free_host(ctx, hSlots[s]);
	}
	vp_free_device(ctx, dRaw);
	vp_free_host(ctx, doneWord);
	vp_ctx_destroy(ctx);
	return 0;
}
