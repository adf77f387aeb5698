// vp_feeds --gpus N: the multi-GPU form of the feed pipeline, ONE process on the C ABIs alone (include/vp_hip.h, vp_multi.h, vp_host.h).
// Feeds are sharded over the devices (feed i of device g has the seed feed_seed(g, i) of vision-processor_amd/feeds.py and the camera id
// g * feeds + i).  ONE DRIVER THREAD PER DEVICE with its own context, two streams, two sets of frame buffers (vp_frame_buffers_create) and -- with
// --detect -- its own detectors and host threads:
//     step k on stream k & 1:  vp_frame -> vp_records_compact -> (detect: read-back of the compact slots)
//     detect, for step k - 1 while the GPU works on step k:  vph_detect_batch_slots -> packet block -> copy into the device's gather block
// and ONE grouped RCCL all-gather per step (vp_gather_start, issued by the thread that reaches the step's rendezvous last) hands every device every
// device's block -- the encoded SSL_WrapperPackets with --detect, the compact blob records without -- on the gather streams, beside the next step's
// kernels.  This replaces the reference's per-camera processes (one loop per camera, src/main.cpp:262-317; hypotheses -> SSL_WrapperPacket,
// main.cpp:319-370) exchanging packets over UDP multicast (src/udpsocket.cpp:94-100 send, :147-163 receive).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "vp_hip.h"
#include "vp_host.h"
#include "vp_multi.h"
#include "feeds.h"

#define CK(call) \
	do { \
		const int rc_ = (call); \
		if(rc_ != VP_OK) { \
			fprintf(stderr, "[FATAL] %s failed (%d): %s\n", #call, rc_, vp_last_error()); \
			exit(1); \
		} \
	} while(0)


namespace {

// every driver thread arrives once per step; the last one runs `last()` (the grouped gather) before anyone is released
class Rendezvous {
	std::mutex mu;
	std::condition_variable cv;
	int n, waiting = 0;
	unsigned long generation = 0;
public:
	explicit Rendezvous(int n_): n(n_) {}
	template<typename F> void arrive(F last) {
		std::unique_lock<std::mutex> lk(mu);
		if(++waiting == n) {
			last();
			waiting = 0;
			generation++;
			cv.notify_all();
			return;
		}
		const unsigned long g = generation;
		cv.wait(lk, [&] { return generation != g; });
	}
};

struct Dev {
	int g = 0;
	VpContext* ctx = nullptr;
	uint8_t* dRaw = nullptr;
	VpFrameSlab* slab[2] = {nullptr, nullptr};
	VpFrameBuffers buf[2];
	uint8_t* dSlots[2] = {nullptr, nullptr}; // detect: compact record slots of a step (device), read back into hSlots
	void* hSlots[2] = {nullptr, nullptr};
	void* hPackets[2] = {nullptr, nullptr};
	std::vector<VphDetector*> det;
	uint8_t* dSlotsFull = nullptr; // a frame with more blobs than --cap: the step's records once more with cap = maxBlobs (allocated on first use)
	void* hSlotsFull = nullptr;
	long blobs = 0, packetBytes = 0, robots = 0, balls = 0, overflowSteps = 0;
	double stageMs = 0, waitMs = 0;
	VpPlacement place[2];
};

} // namespace

int vp_feeds_run_multi(const MultiOptions& o) {
	const int visible = vp_device_count();
	if(o.gpus < 1 || visible < 1 || (o.gpus > visible && !o.same_device)) {
		fprintf(stderr, "[FATAL] --gpus %d but %d HIP device(s) are visible\n", o.gpus, visible);
		return 1;
	}
	VpFrameParams p;
	vp_feeds_cfg2_params(&p);
	if(o.local) p.sat_variant = VP_SAT_LOCAL; // no global table: resampling + ONE kernel per step and device
	const int N = o.gpus, feeds = o.feeds, mm = p.max_matches, packetSlot = 4096;
	const size_t rawBytes = (size_t)4 * p.qw * p.qh;
	// what a device contributes per step
	size_t offC = 0, offM = 0, offI = 0, slotBytes = 0;
	const size_t fullBytes = vp_record_layout(feeds, mm, &offC, &offM, &offI);
	const size_t compactBytes = vp_record_layout_compact(feeds, o.cap, &slotBytes, nullptr, nullptr);
	size_t fullSlotBytes = 0;
	const size_t fullSlotBlock = vp_record_layout_compact(feeds, mm, &fullSlotBytes, nullptr, nullptr); // the overflow path of --detect
	const size_t packetBytes = vp_packet_layout(feeds, packetSlot);
	const size_t blockBytes = o.detect ? packetBytes : (o.full ? fullBytes : compactBytes);
	const char* blockKind = o.detect ? "SSL_WrapperPacket slots" : (o.full ? "full records" : "compact records");

	std::vector<int> devs(N);
	for(int g = 0; g < N; g++) devs[g] = o.same_device ? 0 : g;
	VpGather* gather = nullptr;
	if(o.same_device) CK(vp_gather_create_loopback(N, 0, blockBytes, 2, &gather)); // test-only: N ranks on device 0, asked for by name, never a fallback
	else CK(vp_gather_create(N, devs.data(), blockBytes, 2, &gather));
	std::vector<Dev> dev(N);
	std::vector<VpContext*> ctxs(N, nullptr);
	const int threadsPerDev = std::max(1, o.threads / N);
	const float calibBase[8] = {p.model.f, p.model.p[0], p.model.p[1], p.model.d, -1.f, 0.f, 0.f, 0.f};
	const int32_t field[5] = {9000, 6000, 300, -1, 10};

	// set-up, one thread per device (rendering the synthetic feeds is the slow part)
	{
		std::vector<std::thread> th;
		for(int g = 0; g < N; g++)
			th.emplace_back([&, g] {
				Dev& d = dev[g];
				d.g = g;
				CK(vp_ctx_create(devs[g], 2, &d.ctx));
				if(!o.same_device) CK(vp_ctx_tune(d.ctx, "placement_max_mib", -1)); // one context per device, the program owns its GPUs: the search may hold nine tenths of the free memory
				if(o.same_device) CK(vp_ctx_tune(d.ctx, "placement_candidates", 0)); // N contexts searching one device's memory at once would starve each other; the rehearsal does not care where its blocks land
				ctxs[g] = d.ctx;
				CK(vp_malloc_device(d.ctx, rawBytes * feeds, (void**)&d.dRaw));
				std::vector<uint8_t> frame(rawBytes);
				for(int i = 0; i < feeds; i++) { // feeds.py feed_seed(rank, frame) = rank * 100000 + frame
					CK(vp_synth_frame(&p.model, p.layout, p.qw, p.qh, p.max_robot_height, 9000.f, 6000.f, 0x5EED0000u + (uint32_t)(g * 100000 + i), 16, 8, 3.0f, frame.data()));
					CK(vp_copy_h2d(d.ctx, 0, d.dRaw + rawBytes * i, frame.data(), rawBytes));
					CK(vp_sync(d.ctx, 0));
				}
				for(int s = 0; s < 2; s++) {
					CK(vp_frame_buffers_create(d.ctx, &p, feeds, 0, &d.buf[s], &d.slab[s]));
					if(o.full && !o.detect) { // vp_frame writes its outputs into the gather block itself
						uint8_t* rec = static_cast<uint8_t*>(vp_gather_send(gather, g, s));
						d.buf[s].counter = reinterpret_cast<int32_t*>(rec + offC);
						d.buf[s].matches = reinterpret_cast<VpMatch*>(rec + offM);
						d.buf[s].idx = reinterpret_cast<uint32_t*>(rec + offI);
					}
					if(o.detect) {
						CK(vp_malloc_device(d.ctx, compactBytes, (void**)&d.dSlots[s]));
						CK(vp_malloc_host(d.ctx, compactBytes, &d.hSlots[s]));
						CK(vp_malloc_host(d.ctx, packetBytes, &d.hPackets[s]));
						memset(d.hPackets[s], 0, packetBytes);
					}
				}
				CK(vp_frame_buffers_placement(d.slab[0], &d.place[0], &d.place[1]));
				if(o.detect) {
					const float calib[11] = {calibBase[0], calibBase[1], calibBase[2], calibBase[3], calibBase[4], calibBase[5], calibBase[6], calibBase[7], -p.model.c[0], p.model.c[1],
							p.model.c[2]}; // nadir: t = R (-pos)
					for(int i = 0; i < feeds; i++) {
						VphConfig cfg;
						vph_config_defaults(&cfg);
						cfg.cam_id = g * feeds + i; // every feed is its own camera: own colour model, own tracker
						VphDetector* dt = vph_detector_create(&cfg);
						if(!dt || vph_detector_set_geometry_direct(dt, calib, p.qw, p.qh, p.field_scale, p.off_x, p.off_y, p.fw, p.fh, field, 21.5f, 90.f, nullptr, 0, nullptr, 0) != VP_OK) {
							fprintf(stderr, "[FATAL] detector %d of device %d\n", i, g);
							exit(1);
						}
						d.det.push_back(dt);
					}
				}
			});
		for(std::thread& t : th) t.join();
	}

	Rendezvous meet(N);
	std::vector<int> gstreams(N, 0);
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	const int total = o.warmup + o.steps;
	std::chrono::steady_clock::time_point t0, t1;

	// The gather of slot s for step j is started by the last thread to reach the rendezvous; it records an event on every device's stream s, so all
	// work queued there so far (vp_frame + compaction, or the packet copy) precedes it.
	auto gatherSlot = [&](int s) {
		meet.arrive([&] {
			for(int g = 0; g < N; g++) gstreams[g] = s;
			CK(vp_gather_start(gather, s, ctxs.data(), gstreams.data()));
		});
	};
	auto driver = [&](int g) {
		Dev& d = dev[g];
		auto launch = [&](int k) { // step k (0-based) on stream / buffer set k & 1
			const int s = k & 1;
			CK(vp_frame(d.ctx, s, &p, d.dRaw, feeds, &d.buf[s]));
			if(o.detect) {
				CK(vp_records_compact(d.ctx, s, d.buf[s].counter, d.buf[s].matches, nullptr, feeds, mm, o.cap, d.dSlots[s]));
				CK(vp_copy_d2h(d.ctx, s, d.hSlots[s], d.dSlots[s], compactBytes));
			} else if(!o.full) {
				CK(vp_records_compact(d.ctx, s, d.buf[s].counter, d.buf[s].matches, d.buf[s].idx, feeds, mm, o.cap, static_cast<uint8_t*>(vp_gather_send(gather, g, s))));
			}
		};
		auto cpuStage = [&](int k, bool timed) { // step k's blobs -> packets -> this device's gather block of slot k & 1
			const int s = k & 1;
			const auto w0 = now();
			CK(vp_sync(d.ctx, s)); // the read-back of step k has landed
			const auto w1 = now();
			const uint32_t frameNo = (uint32_t)k + 1; // steps are numbered from 1
			const double t = frameNo / 70.0;
			// The reference reads min(counter[0], maxBlobs) records (main.cpp:301); a compact slot holds at most --cap of them.  A step in which some
			// feed has more is read once more at full capacity (buf[s] still holds step k: the next launch into it is step k + 2) -- nothing is dropped.
			const uint8_t* slots = static_cast<const uint8_t*>(d.hSlots[s]);
			size_t stride = slotBytes;
			bool overflow = false;
			for(int i = 0; i < feeds && !overflow; i++) {
				int32_t head[4];
				memcpy(head, slots + (size_t)i * slotBytes, sizeof(head));
				overflow = head[3] < std::min(std::max(head[0], 0), mm);
			}
			if(overflow) {
				if(!d.dSlotsFull) {
					CK(vp_malloc_device(d.ctx, fullSlotBlock, (void**)&d.dSlotsFull));
					CK(vp_malloc_host(d.ctx, fullSlotBlock, &d.hSlotsFull));
				}
				CK(vp_records_compact(d.ctx, s, d.buf[s].counter, d.buf[s].matches, nullptr, feeds, mm, mm, d.dSlotsFull));
				CK(vp_copy_d2h(d.ctx, s, d.hSlotsFull, d.dSlotsFull, fullSlotBlock));
				CK(vp_sync(d.ctx, s));
				slots = static_cast<const uint8_t*>(d.hSlotsFull);
				stride = fullSlotBytes;
				d.overflowSteps++;
			}
			const int rc = vph_detect_batch_slots(d.det.data(), feeds, slots, stride, mm, frameNo, t, t, t + 0.002, threadsPerDev, static_cast<uint8_t*>(d.hPackets[s]), packetSlot);
			if(rc != VP_OK) {
				fprintf(stderr, "[FATAL] device %d step %d: the CPU stage refused the record slots: %s\n", g, k, vp_last_error());
				exit(1);
			}
			CK(vp_gather_wait(gather, s)); // the gather issued two steps ago has released this slot's block
			CK(vp_copy_h2d(d.ctx, s, vp_gather_send(gather, g, s), d.hPackets[s], packetBytes));
			if(timed) {
				d.waitMs += ms(w0, w1);
				d.stageMs += ms(w1, now());
				for(int i = 0; i < feeds; i++) {
					int32_t head[4], len;
					memcpy(head, slots + (size_t)i * stride, sizeof(head));
					memcpy(&len, static_cast<const uint8_t*>(d.hPackets[s]) + (size_t)i * packetSlot, 4);
					d.blobs += head[3];
					d.packetBytes += len > 0 ? len : 0;
				}
			}
		};
		for(int k = 0; k < total; k++) {
			if(k == o.warmup) { // everything issued so far completes, then all devices start the clock together
				if(o.detect && k > 0) {
					cpuStage(k - 1, false);
					gatherSlot((k - 1) & 1);
				}
				CK(vp_gather_wait(gather, 0));
				CK(vp_gather_wait(gather, 1));
				CK(vp_sync(d.ctx, -1));
				meet.arrive([&] { t0 = now(); });
			}
			const int s = k & 1;
			if(!o.detect) CK(vp_gather_wait(gather, s)); // the gather issued two steps ago has released this slot's block (vp_frame / the compaction write into it)
			launch(k);
			if(!o.detect) {
				gatherSlot(s);
			} else if(k > 0 && k != o.warmup) {
				cpuStage(k - 1, k - 1 >= o.warmup); // the previous step's CPU stage while the GPU works on this one
				gatherSlot((k - 1) & 1);
			}
		}
		if(o.detect) { // the last step's CPU stage has no following GPU step to hide behind
			cpuStage(total - 1, total - 1 >= o.warmup);
			gatherSlot((total - 1) & 1);
		}
		CK(vp_gather_wait(gather, 0)); // every step's gather has completed inside the timed region
		CK(vp_gather_wait(gather, 1));
		CK(vp_sync(d.ctx, -1));
		meet.arrive([&] { t1 = now(); });
	};
	{
		std::vector<std::thread> th;
		for(int g = 0; g < N; g++) th.emplace_back(driver, g);
		for(std::thread& t : th) t.join();
	}
	const double dt = std::chrono::duration<double>(t1 - t0).count();

	// what the last step gathered: every device holds every device's block, byte for byte what the owners produced
	const int last = (total - 1) & 1;
	int verified = -1;
	std::vector<uint8_t> own((size_t)N * blockBytes), got((size_t)N * blockBytes);
	for(int g = 0; g < N; g++) CK(vp_copy_d2h_sync(dev[g].ctx, 0, own.data() + (size_t)g * blockBytes, vp_gather_send(gather, g, last), blockBytes));
	if(o.verify) {
		verified = 1;
		for(int g = 0; g < N; g++) {
			CK(vp_copy_d2h_sync(dev[g].ctx, 0, got.data(), vp_gather_recv(gather, g, last), (size_t)N * blockBytes));
			if(memcmp(got.data(), own.data(), (size_t)N * blockBytes) != 0) {
				fprintf(stderr, "[FATAL] device %d received blocks that differ from what their owners produced\n", g);
				verified = 0;
			}
		}
	}
	if(o.dump) { // what the LAST device received in the last step: N blocks in rank order
		CK(vp_copy_d2h_sync(dev[N - 1].ctx, 0, got.data(), vp_gather_recv(gather, N - 1, last), (size_t)N * blockBytes));
		FILE* f = fopen(o.dump, "wb");
		if(!f || fwrite(got.data(), 1, (size_t)N * blockBytes, f) != (size_t)N * blockBytes) {
			fprintf(stderr, "[FATAL] cannot write %s\n", o.dump);
			return 1;
		}
		fclose(f);
	}

	long blobs = 0;
	printf("{\"program\": \"vp_feeds\", \"gpus\": %d, \"feeds_per_gpu\": %d, \"steps\": %d, \"frames_per_s\": %.1f, \"ms_per_step\": %.4f, \"record_bytes_per_gpu\": %zu, \"block\": \"%s\", "
		   "\"full_record_bytes_per_gpu\": %zu, \"gather\": \"%s, one grouped call per step, %d rank(s) in one process, one driver thread per device\", \"same_device\": %s, \"sat_variant\": \"%s\"",
			N, feeds, o.steps, (double)N * feeds * o.steps / dt, 1e3 * dt / o.steps, blockBytes, blockKind, fullBytes,
			vp_gather_is_loopback(gather) ? "TEST-ONLY loop-back (all ranks on device 0, device-to-device copies, no RCCL)" : "RCCL ncclAllGather", N, o.same_device ? "true" : "false",
			o.local ? "local" : "tiled");
	if(o.detect) {
		long robots = 0, balls = 0, pbytes = 0, overflowSteps = 0;
		double stage = 0, wait = 0;
		for(Dev& d : dev) {
			for(VphDetector* dt2 : d.det) {
				VphFrameHeader h;
				vph_frame_header(dt2, &h);
				robots += h.n_robots_yellow + h.n_robots_blue;
				balls += h.n_balls;
			}
			blobs += d.blobs;
			overflowSteps += d.overflowSteps;
			pbytes += d.packetBytes;
			stage += d.stageMs;
			wait += d.waitMs;
		}
		printf(", \"detect\": {\"host_threads_per_gpu\": %d, \"robots_per_frame\": %.3f, \"balls_per_frame\": %.3f, \"packet_bytes_per_frame\": %.1f, \"packet_slot_bytes\": %d, "
			   "\"host_ms_per_step\": {\"wait_for_gpu\": %.3f, \"cpu_stage\": %.3f}, \"steps_reread_at_full_capacity\": %ld, \"rendered\": {\"robots\": 16, \"balls\": 8}}, \"blobs_per_frame\": %.2f",
				threadsPerDev, robots / ((double)N * feeds), balls / ((double)N * feeds), pbytes / ((double)N * feeds * o.steps), packetSlot, wait / N / o.steps, stage / N / o.steps,
				overflowSteps, blobs / ((double)N * feeds * o.steps));
	} else {
		long truncated = 0; // compact slots of the LAST step that hold fewer records than min(counter[0], maxBlobs): --cap too small for that frame
		printf(", \"counters\": [");
		for(int g = 0; g < N; g++) {
			printf("%s[", g ? ", " : "");
			for(int i = 0; i < feeds; i++) {
				int32_t c[4];
				if(o.full) memcpy(c, own.data() + (size_t)g * blockBytes + offC + 12 * (size_t)i, 12);
				else memcpy(c, own.data() + (size_t)g * blockBytes + (size_t)i * slotBytes, 16);
				printf("%s[%d, %d, %d]", i ? ", " : "", c[0], c[1], c[2]);
				blobs += c[0] < mm ? c[0] : mm;
				if(!o.full && c[3] < std::min(std::max(c[0], 0), mm)) truncated++;
			}
			printf("]");
		}
		printf("], \"blobs_per_frame\": %.2f, \"truncated_slots_last_step\": %ld", blobs / ((double)N * feeds), truncated);
		if(truncated) fprintf(stderr, "[WARN] %ld compact record slot(s) of the last step hold fewer records than the host loop reads (main.cpp:301): raise --cap (%d)\n", truncated, o.cap);
	}
	printf(", \"placement\": {\"sat_write_gbs\": [");
	for(int g = 0; g < N; g++) printf("%s%.0f", g ? ", " : "", dev[g].place[0].write_gbs);
	printf("], \"candidates\": [");
	for(int g = 0; g < N; g++) printf("%s%d", g ? ", " : "", dev[g].place[0].candidates + 0);
	printf("], \"search\": [");
	int exhausted = 0;
	for(int g = 0; g < N; g++) {
		const int sel = dev[g].place[0].selected;
		exhausted += sel == 2 && dev[g].place[0].write_gbs < dev[g].place[0].fast_gbs;
		printf("%s\"%s\"", g ? ", " : "", sel == 1 ? "criterion met" : sel == 2 ? "budget exhausted" : "none");
	}
	printf("]");
	if(exhausted) {
		printf(", \"warning\": \"%d device(s) kept the best block seen although none reached the write rate of a fast block\"", exhausted);
		fprintf(stderr, "[WARN] placement search ran out of candidates, bytes or memory on %d device(s): expect the SAT pass 5-15 %% slower there\n", exhausted);
	}
	printf("}");
	if(o.verify) printf(", \"gather_verified\": %s", verified == 1 ? "true" : "false");
	printf("}\n");

	for(Dev& d : dev) {
		for(VphDetector* dt2 : d.det) vph_detector_destroy(dt2);
		for(int s = 0; s < 2; s++) {
			vp_frame_buffers_destroy(d.ctx, d.slab[s]);
			if(d.dSlots[s]) vp_free_device(d.ctx, d.dSlots[s]);
			if(s == 0 && d.dSlotsFull) vp_free_device(d.ctx, d.dSlotsFull);
			if(s == 0 && d.hSlotsFull) vp_free_host(d.ctx, d.hSlotsFull);
			if(d.hSlots[s]) vp_free_host(d.ctx, d.hSlots[s]);
			if(d.hPackets[s]) vp_free_host(d.ctx, d.hPackets[s]);
		}
		vp_free_device(d.ctx, d.dRaw);
	}
	vp_gather_destroy(gather);
	for(Dev& d : dev) vp_ctx_destroy(d.ctx);
	return verified == 0 ? 1 : 0;
}
