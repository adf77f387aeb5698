// Device memory with a checked placement (include/vp_hip.h: vp_probe_stream, vp_malloc_device_placed, vp_frame_buffers_*).
//
// Why this exists: the gradientDot + SAT pass is HBM-bound and its duration moves by up to 15 % with the physical pages its input and output
// happen to get from hipMalloc (profiles/r02_bench_spread.json).  The reference allocates every image once and keeps it in a
// pool for the life of the process (OpenCL::acquire, opencl.cpp:108-122; CLArray, opencl.cpp:137-159), so where an allocation lands is decided
// once -- and can be decided deliberately.
#include "vp_common.h"

#include <algorithm>
#include <cstring>

// ---- stream probes: what a buffer's placement gives a pure 16-byte-per-lane read or write stream --------------------------------------
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_probe_read(const v4u* __restrict__ in, size_t n16, unsigned* __restrict__ sink) {
	v4u acc = {0, 0, 0, 0};
	for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
		const v4u v = __builtin_nontemporal_load(in + i);
		acc ^= v;
	}
	if((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) *sink = acc.x; // practically never: keeps the loads alive
}
__global__ __launch_bounds__(256) void k_probe_write(v4u* __restrict__ out, size_t n16) {
	for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v4u{0u, 0u, 0u, 0u}, out + i);
}

// mean microseconds of `reps` launches of a pure read (write = 0) or zero-fill (write = 1: the buffer's contents are lost) stream over the buffer
int vp_probe_stream_impl(VpContext* ctx, int stream, void* ptr, size_t bytes, int write, int reps, float* us) {
	VP_REQUIRE(ptr && bytes >= 4096 && reps >= 1 && us && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0, "vp_probe_stream: bad arguments");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	hipStream_t s = ctx->streams[stream];
	void* sink = nullptr;
	VP_TRY(vp_workspace(ctx, stream, 0, 256, &sink));
	hipEvent_t e0, e1;
	VP_CHECK_HIP(hipEventCreate(&e0));
	if(hipEventCreate(&e1) != hipSuccess) {
		(void)hipEventDestroy(e0);
		vp_set_error("vp_probe_stream: hipEventCreate failed");
		return VP_ERR_HIP;
	}
	const unsigned grid = (unsigned)ctx->cu_count * 16;
	const size_t n16 = bytes / 16;
	for(int i = -1; i < reps; i++) {
		if(i == 0) (void)hipEventRecord(e0, s);
		if(write) k_probe_write<<<grid, 256, 0, s>>>(static_cast<v4u*>(ptr), n16);
		else k_probe_read<<<grid, 256, 0, s>>>(static_cast<const v4u*>(ptr), n16, static_cast<unsigned*>(sink));
	}
	(void)hipEventRecord(e1, s);
	hipError_t e = hipEventSynchronize(e1);
	float ms = 0.f;
	if(e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	VP_CHECK_HIP(e);
	*us = ms * 1e3f / (float)reps;
	return vp_launch_ok("stream probe");
}

extern "C" int vp_probe_stream(VpContext* ctx, int stream, void* ptr, size_t bytes, int write, int reps, float* us) {
	VP_TRY(vp_check_ctx(ctx, stream));
	return vp_probe_stream_impl(ctx, stream, ptr, bytes, write, reps, us);
}

// ---- placement-checked allocation -----------------------------------------------------------------------------------------------------
// What was measured (tools/placement_scan.py, profiles/r03_placement.json): the zero-fill stream of a 1.6 GB hipMalloc buffer runs at either
// 4.95-5.4 TB/s or 3.8-4.4 TB/s; the figure belongs to the allocation (identical on every later sweep, identical for a buffer re-allocated at
// once), not to its virtual address (the same address range is fast in one allocation and slow in the next), clusters in allocation order (22
// consecutive allocations = 33 GB all fast, ~30 % of 100 overall) and leaves read streams untouched (6.7-6.9 TB/s everywhere).  The SAT pass writes
// its table once per launch: 608 us into a fast allocation, 650-705 us into a slow one.  So: allocate candidates WITHOUT freeing (a freed block is
// handed out again by the next hipMalloc), probe each, keep the fastest, free the rest.
// When to stop: a block counts as fast when its zero-fill rate reaches 0.70 of the READ-stream rate measured on the first candidate (reads do not
// depend on the placement: 6.5-6.9 TB/s; fast blocks write at 0.74-0.81 of that, slow ones at 0.57-0.66) -- a ratio of two measurements of this device,
// not an MI355X constant.  The search ends once `count` blocks are fast, or once sixteen candidates lie within 6 % of each other (one population:
// nothing to choose).  [A rule relative to the best block seen was tried first and stopped inside the slow population, which is itself 15 % wide.]
// It never holds more than "placement_max_mib" (vp_ctx_tune; 0 = half of the free memory, at most a quarter of the device) at once and stops while an
// eighth of the device is still free: a second context or process on the same GPU keeps room.  VpPlacement.selected says how it ended: 1 = criterion met, 2 = budget (candidates or bytes) or memory ran out first and the best blocks seen
// were kept.
// need_fast: how many of the `count` blocks must be fast for the search to stop (0 = all of them)
int vp_malloc_device_placed_impl(VpContext* ctx, size_t bytes, int count, void** dptrs, VpPlacement* info, int need_fast = 0) {
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	for(int i = 0; i < count; i++) dptrs[i] = nullptr;
	const int budget = ctx->tune.placement_candidates;
	// below a few Infinity Caches (256 MiB) a write stream never reaches HBM at its own pace: nothing to measure, nothing to gain
	const bool select = budget > count && bytes >= (size_t(768) << 20);
	struct Cand {
		void* p;
		float gbs;
	};
	std::vector<Cand> cands;
	auto release_all = [&] {
		for(const Cand& c : cands) (void)hipFree(c.p);
		cands.clear();
	};
	size_t free_b = 0, total_b = 0;
	VP_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
	// What the search may hold at once: "placement_max_mib", else half of what was free when it started but never more than a quarter of the device
	// (two searches that start together on one GPU -- two contexts, two processes -- then hold half of it at most between them).  Independently of
	// that cap the loop below asks the device again before every candidate and stops once the memory free NOW, whoever took the rest, would fall
	// under a reserve of an eighth of the device: somebody else's hipMalloc must not fail because this search is shopping.
	// "placement_max_mib" = -1: the caller says the GPU is its own (bench.py, vp_feeds with one context per device): nine tenths of what is free, and a small
	// absolute reserve (1 GiB or 2 % of the device, whichever is more) for what the runtime, RCCL or a caching allocator on another thread may ask for meanwhile (ADVICE
	// r05).  NO OTHER THREAD SHOULD ALLOCATE LARGE BLOCKS ON THE DEVICE while vp_frame_buffers_create / vp_malloc_device_placed runs in this mode.
	size_t spend = std::min(free_b / 2, total_b / 4);
	if(ctx->tune.placement_max_mib > 0) spend = std::min(free_b / 2, (size_t)ctx->tune.placement_max_mib << 20);
	else if(ctx->tune.placement_max_mib < 0) spend = free_b - free_b / 10; // the GPU is the caller's own: nine tenths of what is free (at 256 cfg2 frames per step a candidate is 3.2 GB:
	                                                                        // half of the memory was 44 candidates and ran out on boxes whose fast blocks are scarce)
	const size_t reserve = ctx->tune.placement_max_mib < 0 ? std::max(size_t(1) << 30, total_b / 50) : total_b / 8;
	bool met = false;
	float fast_gbs = 0.f; // 0.70 x the read-stream rate of this device, measured on the first candidate
	int n_fast = 0;
	while((int)cands.size() < (select ? budget : count)) {
		if(select && (int)cands.size() >= count) {
			if((cands.size() + 1) * bytes > spend) break;
			size_t free_now = 0, t_ = 0;
			if(hipMemGetInfo(&free_now, &t_) != hipSuccess || free_now < bytes + reserve) break;
		}
		void* p = nullptr;
		if(hipMalloc(&p, bytes) != hipSuccess) {
			(void)hipGetLastError();
			if((int)cands.size() >= count) break; // enough to hand out: stop searching
			release_all();
			vp_set_error("hipMalloc(%zu) failed (buffer %d of %d)", bytes, (int)cands.size() + 1, count);
			return VP_ERR_NOMEM;
		}
		float us = 0.f;
		if(select) {
			const int rc = vp_probe_stream_impl(ctx, 0, p, bytes & ~size_t(15), 1, 3, &us);
			if(rc != VP_OK) {
				(void)hipFree(p);
				release_all();
				return rc;
			}
		}
		const float gbs = us > 0.f ? (float)((double)bytes / us * 1e-3) : 0.f;
		if(select && cands.empty()) {
			float rus = 0.f;
			const int rc = vp_probe_stream_impl(ctx, 0, p, bytes & ~size_t(15), 0, 3, &rus);
			if(rc != VP_OK) {
				(void)hipFree(p);
				return rc;
			}
			fast_gbs = rus > 0.f ? 0.70f * (float)((double)bytes / rus * 1e-3) : 0.f;
		}
		cands.push_back({p, gbs});
		if(select) {
			n_fast += fast_gbs > 0.f && gbs >= fast_gbs;
			if(n_fast >= (need_fast > 0 ? need_fast : count) && (int)cands.size() >= count) { // (enough fast ones AND enough blocks to hand out)
				met = true;
				break;
			}
			if(cands.size() >= 16) { // one population?
				float best = 0.f, worst = gbs;
				for(const Cand& c : cands) {
					best = std::max(best, c.gbs);
					worst = std::min(worst, c.gbs);
				}
				if(worst >= 0.94f * best) {
					met = true;
					break;
				}
			}
		}
	}
	if((int)cands.size() < count) { // (cannot happen: every exit above holds at least `count` blocks -- kept as a guard, a short list once handed out a null flat)
		release_all();
		vp_set_error("vp_malloc_device_placed: internal error, %d of %d blocks", (int)cands.size(), count);
		return VP_ERR_NOMEM;
	}
	std::stable_sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.gbs > b.gbs; });
	for(int i = 0; i < count; i++) {
		dptrs[i] = cands[i].p;
		if(info) info[i] = VpPlacement{cands[i].gbs, cands.front().gbs, cands.back().gbs, (int32_t)cands.size(), select ? (met ? 1 : 2) : 0, select ? fast_gbs : 0.f};
	}
	for(size_t i = count; i < cands.size(); i++) (void)hipFree(cands[i].p);
	return VP_OK;
}

extern "C" int vp_malloc_device_placed(VpContext* ctx, size_t bytes, int count, void** dptrs, VpPlacement* info) {
	VP_REQUIRE(ctx && dptrs && bytes > 0 && count >= 1 && count <= 64, "vp_malloc_device_placed: bad arguments");
	return vp_malloc_device_placed_impl(ctx, bytes, count, dptrs, info);
}

// ---- frame buffers: everything vp_frame writes for `batch` frames of one geometry, allocated once (the reference's image pool, opencl.cpp:108-122) ----
struct VpFrameSlab {
	std::vector<void*> blocks;
	VpFrameBuffers buffers;
	VpPlacement placement[2]; // [0] the table, [1] flat
};

extern "C" int vp_frame_buffers_create(VpContext* ctx, const VpFrameParams* p, int batch, int views, VpFrameBuffers* out, VpFrameSlab** slab_out) {
	VP_REQUIRE(ctx && p && out && slab_out && batch > 0 && batch <= 65535 && p->fw > 0 && p->fh > 0 && p->qw > 0 && p->qh > 0 && p->max_matches > 0, "vp_frame_buffers_create: bad arguments");
	VP_REQUIRE((views & ~(VP_VIEW_PLANES | VP_VIEW_GRADDOT | VP_VIEW_BLOBCENTER)) == 0, "vp_frame_buffers_create: unknown view bits %#x", views);
	const bool local = p->sat_variant == VP_SAT_LOCAL;
	VP_REQUIRE(!local || !(views & (VP_VIEW_GRADDOT | VP_VIEW_BLOBCENTER)), "vp_frame_buffers_create: VP_SAT_LOCAL has no gradDot / blobCenter view");
	VpFrameSlab* slab = new VpFrameSlab();
	memset(&slab->buffers, 0, sizeof(slab->buffers));
	memset(slab->placement, 0, sizeof(slab->placement));
	const size_t npx = (size_t)p->fw * p->fh * batch;
	int rc = VP_OK;
	auto plain = [&](size_t bytes, void** dst) {
		if(rc != VP_OK) return;
		rc = vp_malloc_device(ctx, bytes, dst);
		if(rc == VP_OK) slab->blocks.push_back(*dst);
	};
	// the one image-sized buffer whose placement matters is the TABLE (written once per frame by the HBM-bound 8 B/px pass): the fastest-writing candidate
	// becomes the table, the next one flat -- which is written by the resampling kernel (not HBM-bound) and only read by the SAT pass (reads do not depend
	// on the placement), so one fast block ends the search.  VP_SAT_LOCAL has no table: flat is a plain block.
	if(local) {
		plain(npx * 4, reinterpret_cast<void**>(&slab->buffers.flat));
	} else {
		void* big[2] = {nullptr, nullptr};
		rc = vp_malloc_device_placed_impl(ctx, npx * 4, 2, big, slab->placement, 1);
		if(rc == VP_OK) {
			slab->blocks.push_back(big[0]);
			slab->blocks.push_back(big[1]);
			slab->buffers.sat = static_cast<int32_t*>(big[0]);
			slab->buffers.flat = static_cast<uint8_t*>(big[1]);
		}
	}
	plain(sizeof(VpMatch) * (size_t)p->max_matches * batch, reinterpret_cast<void**>(&slab->buffers.matches));
	plain(sizeof(uint32_t) * (size_t)p->max_matches * batch, reinterpret_cast<void**>(&slab->buffers.idx));
	plain(sizeof(int32_t) * 3 * (size_t)batch, reinterpret_cast<void**>(&slab->buffers.counter));
	if(views & VP_VIEW_PLANES) plain((size_t)4 * p->qw * p->qh * batch, reinterpret_cast<void**>(&slab->buffers.planes));
	if(views & VP_VIEW_GRADDOT) plain(npx * 4, reinterpret_cast<void**>(&slab->buffers.graddot));
	if(views & VP_VIEW_BLOBCENTER) plain(npx * 4, reinterpret_cast<void**>(&slab->buffers.blobcenter));
	if(rc != VP_OK) {
		for(void* b : slab->blocks) (void)hipFree(b);
		delete slab;
		return rc;
	}
	*out = slab->buffers;
	*slab_out = slab;
	return VP_OK;
}

extern "C" int vp_frame_buffers_placement(const VpFrameSlab* slab, VpPlacement* sat, VpPlacement* flat) {
	VP_REQUIRE(slab, "null slab");
	if(sat) *sat = slab->placement[0];
	if(flat) *flat = slab->placement[1];
	return VP_OK;
}

extern "C" int vp_frame_buffers_destroy(VpContext* ctx, VpFrameSlab* slab) {
	VP_REQUIRE(ctx, "null context");
	if(!slab) return VP_OK;
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	for(int i = 0; i < ctx->n_streams; i++) VP_CHECK_HIP(hipStreamSynchronize(ctx->streams[i]));
	for(void* b : slab->blocks) (void)hipFree(b);
	delete slab;
	return VP_OK;
}
