// Multi-GPU end of the hot path (include/vp_multi.h): one process, one RCCL communicator per device, one all-gather of the
// per-step detection records.  Host code only (no kernels); built into libvp_multi.so, which links libvp_hip.so and librccl.so --
// the single-GPU library stays free of the RCCL dependency.
// Replaces the per-camera processes exchanging SSL_WrapperPackets over UDP multicast (src/udpsocket.cpp:94-100, 147-163).
#include "vp_common.h"
#include "vp_multi.h"

#include <rccl/rccl.h>

struct VpGather {
	int n = 0, slots = 0;
	bool loopback = false;                        // vp_gather_create_loopback: every rank on ONE device, device-to-device copies instead of RCCL (tests only)
	size_t bytes = 0;
	std::vector<int> dev;
	std::vector<ncclComm_t> comm;
	std::vector<hipStream_t> gs;                  // gather stream per device
	std::vector<std::vector<void*>> send, recv;   // [rank][slot]
	std::vector<std::vector<hipEvent_t>> ready, done;
	std::vector<char> started;                    // [slot]
};

#define VP_CHECK_NCCL(expr)                                                                                    \
	do {                                                                                                         \
		ncclResult_t r_ = (expr);                                                                                \
		if(r_ != ncclSuccess) {                                                                                  \
			vp_set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__);          \
			return VP_ERR_HIP;                                                                                   \
		}                                                                                                        \
	} while(0)

static size_t pad256(size_t v) { return (v + 255) & ~size_t(255); }

extern "C" size_t vp_record_layout(int batch, int max_matches, size_t* off_counter, size_t* off_matches, size_t* off_idx) {
	if(batch <= 0 || max_matches <= 0) return 0;
	const size_t c = pad256(sizeof(int32_t) * 3 * (size_t)batch), m = pad256(sizeof(VpMatch) * (size_t)max_matches * batch), i = pad256(sizeof(uint32_t) * (size_t)max_matches * batch);
	if(off_counter) *off_counter = 0;
	if(off_matches) *off_matches = c;
	if(off_idx) *off_idx = c + m;
	return c + m + i;
}

extern "C" size_t vp_packet_layout(int batch, int slot_bytes) {
	if(batch <= 0 || slot_bytes <= 4 || (slot_bytes & 3)) return 0;
	return pad256((size_t)batch * slot_bytes);
}

extern "C" void vp_gather_destroy(VpGather* g) {
	if(!g) return;
	for(int r = 0; r < g->n; r++) {
		(void)hipSetDevice(g->dev[r]);
		if(r < (int)g->gs.size() && g->gs[r]) (void)hipStreamSynchronize(g->gs[r]);
		if(r < (int)g->comm.size() && g->comm[r]) (void)ncclCommDestroy(g->comm[r]);
		for(int s = 0; s < g->slots; s++) {
			if(r < (int)g->send.size() && g->send[r][s]) (void)hipFree(g->send[r][s]);
			if(r < (int)g->recv.size() && g->recv[r][s]) (void)hipFree(g->recv[r][s]);
			if(r < (int)g->ready.size() && g->ready[r][s]) (void)hipEventDestroy(g->ready[r][s]);
			if(r < (int)g->done.size() && g->done[r][s]) (void)hipEventDestroy(g->done[r][s]);
		}
		if(r < (int)g->gs.size() && g->gs[r]) (void)hipStreamDestroy(g->gs[r]);
	}
	delete g;
}

static int gather_init(VpGather* g) {
	int have = 0;
	hipError_t e = hipGetDeviceCount(&have);
	if(e != hipSuccess || have <= 0) {
		vp_set_error("no HIP device available (%s): this library has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "0 devices");
		return VP_ERR_NODEVICE;
	}
	for(int r = 0; r < g->n; r++) {
		VP_REQUIRE(g->dev[r] >= 0 && g->dev[r] < have, "vp_gather_create: device %d out of range [0,%d)", g->dev[r], have);
		for(int q = 0; q < r && !g->loopback; q++) VP_REQUIRE(g->dev[q] != g->dev[r], "vp_gather_create: device %d listed twice", g->dev[r]);
	}
	g->gs.assign(g->n, nullptr);
	g->comm.assign(g->n, nullptr);
	g->send.assign(g->n, std::vector<void*>(g->slots, nullptr));
	g->recv.assign(g->n, std::vector<void*>(g->slots, nullptr));
	g->ready.assign(g->n, std::vector<hipEvent_t>(g->slots, nullptr));
	g->done.assign(g->n, std::vector<hipEvent_t>(g->slots, nullptr));
	g->started.assign(g->slots, 0);
	for(int r = 0; r < g->n; r++) {
		VP_CHECK_HIP(hipSetDevice(g->dev[r]));
		VP_CHECK_HIP(hipStreamCreateWithFlags(&g->gs[r], hipStreamNonBlocking));
		for(int s = 0; s < g->slots; s++) {
			if(hipMalloc(&g->send[r][s], g->bytes) != hipSuccess || hipMalloc(&g->recv[r][s], g->bytes * (size_t)g->n) != hipSuccess) {
				vp_set_error("vp_gather_create: hipMalloc of the record blocks failed on device %d", g->dev[r]);
				return VP_ERR_NOMEM;
			}
			VP_CHECK_HIP(hipMemset(g->send[r][s], 0, g->bytes));
			VP_CHECK_HIP(hipEventCreateWithFlags(&g->ready[r][s], hipEventDisableTiming));
			VP_CHECK_HIP(hipEventCreateWithFlags(&g->done[r][s], hipEventDisableTiming));
		}
	}
	if(!g->loopback) VP_CHECK_NCCL(ncclCommInitAll(g->comm.data(), g->n, g->dev.data()));
	return VP_OK;
}

static int gather_create(int n_devices, const int* devices, size_t bytes_per_device, int n_slots, bool loopback, VpGather** out) {
	VpGather* g = new VpGather();
	g->n = n_devices;
	g->slots = n_slots;
	g->bytes = bytes_per_device;
	g->loopback = loopback;
	g->dev.assign(devices, devices + n_devices);
	const int rc = gather_init(g);
	if(rc != VP_OK) {
		vp_gather_destroy(g);
		return rc;
	}
	*out = g;
	return VP_OK;
}

extern "C" int vp_gather_create(int n_devices, const int* devices, size_t bytes_per_device, int n_slots, VpGather** out) {
	VP_REQUIRE(out && devices && n_devices >= 1 && n_devices <= 64 && bytes_per_device > 0 && n_slots >= 1 && n_slots <= 4, "vp_gather_create: bad arguments");
	return gather_create(n_devices, devices, bytes_per_device, n_slots, false, out);
}

// Test-only: n ranks that all live on ONE device.  Never chosen implicitly and never a fallback of vp_gather_create; exists so that the N > 1 host
// pipeline (driver threads, rendezvous, slot reuse, end-of-run drain of host/feeds_multi.cpp) can execute on a one-GPU box.
extern "C" int vp_gather_create_loopback(int n_ranks, int device, size_t bytes_per_rank, int n_slots, VpGather** out) {
	VP_REQUIRE(out && n_ranks >= 1 && n_ranks <= 64 && bytes_per_rank > 0 && n_slots >= 1 && n_slots <= 4, "vp_gather_create_loopback: bad arguments");
	const std::vector<int> devs(n_ranks, device);
	return gather_create(n_ranks, devs.data(), bytes_per_rank, n_slots, true, out);
}

extern "C" int vp_gather_is_loopback(const VpGather* g) { return g && g->loopback ? 1 : 0; }

extern "C" int vp_gather_devices(const VpGather* g) { return g ? g->n : 0; }
extern "C" void* vp_gather_send(VpGather* g, int rank, int slot) { return (g && rank >= 0 && rank < g->n && slot >= 0 && slot < g->slots) ? g->send[rank][slot] : nullptr; }
extern "C" void* vp_gather_recv(VpGather* g, int rank, int slot) { return (g && rank >= 0 && rank < g->n && slot >= 0 && slot < g->slots) ? g->recv[rank][slot] : nullptr; }

extern "C" int vp_gather_start(VpGather* g, int slot, VpContext* const* ctx, const int* stream) {
	VP_REQUIRE(g && ctx && stream && slot >= 0 && slot < g->slots, "vp_gather_start: bad arguments");
	for(int r = 0; r < g->n; r++) {
		VP_TRY(vp_check_ctx(ctx[r], stream[r]));
		VP_REQUIRE(ctx[r]->device == g->dev[r], "vp_gather_start: context %d lives on device %d, the communicator's rank %d on device %d", r, ctx[r]->device, r, g->dev[r]);
		VP_CHECK_HIP(hipSetDevice(g->dev[r]));
		VP_CHECK_HIP(hipEventRecord(g->ready[r][slot], ctx[r]->streams[stream[r]]));
		VP_CHECK_HIP(hipStreamWaitEvent(g->gs[r], g->ready[r][slot], 0));
	}
	if(g->loopback) { // the all-gather as n x n device-to-device copies: receiver r's gather stream waits for EVERY owner's stream, then copies every block
		VP_CHECK_HIP(hipSetDevice(g->dev[0]));
		for(int r = 0; r < g->n; r++) {
			for(int q = 0; q < g->n; q++) {
				if(q != r) VP_CHECK_HIP(hipStreamWaitEvent(g->gs[r], g->ready[q][slot], 0));
				VP_CHECK_HIP(hipMemcpyAsync(static_cast<uint8_t*>(g->recv[r][slot]) + (size_t)q * g->bytes, g->send[q][slot], g->bytes, hipMemcpyDeviceToDevice, g->gs[r]));
			}
			VP_CHECK_HIP(hipEventRecord(g->done[r][slot], g->gs[r]));
		}
		g->started[slot] = 1;
		return VP_OK;
	}
	VP_CHECK_NCCL(ncclGroupStart());
	for(int r = 0; r < g->n; r++) {
		const ncclResult_t rc = ncclAllGather(g->send[r][slot], g->recv[r][slot], g->bytes, ncclUint8, g->comm[r], g->gs[r]);
		if(rc != ncclSuccess) {
			(void)ncclGroupEnd();
			vp_set_error("ncclAllGather failed on rank %d: %s", r, ncclGetErrorString(rc));
			return VP_ERR_HIP;
		}
	}
	VP_CHECK_NCCL(ncclGroupEnd());
	for(int r = 0; r < g->n; r++) {
		VP_CHECK_HIP(hipSetDevice(g->dev[r]));
		VP_CHECK_HIP(hipEventRecord(g->done[r][slot], g->gs[r]));
	}
	g->started[slot] = 1;
	return VP_OK;
}

extern "C" int vp_gather_wait(VpGather* g, int slot) {
	VP_REQUIRE(g && slot >= 0 && slot < g->slots, "vp_gather_wait: bad arguments");
	if(!g->started[slot]) return VP_OK;
	for(int r = 0; r < g->n; r++) {
		VP_CHECK_HIP(hipSetDevice(g->dev[r]));
		VP_CHECK_HIP(hipEventSynchronize(g->done[r][slot]));
	}
	return VP_OK;
}
