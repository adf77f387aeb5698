// Context / runtime layer of the C ABI: replaces the reference's OpenCL wrapper (src/opencl.{h,cpp}).
//   OpenCL::OpenCL (device pick, one in-order profiling queue)   opencl.cpp:34-50   -> vp_ctx_create
//   OpenCL::wait                                                   opencl.cpp:87-92   -> vp_sync
//   OpenCL::printRuntimes / clearEvents                            opencl.cpp:94-105  -> vp_event_times / vp_events_clear
//   clAlloc / CLArray / CLImage allocation, map/unmap              opencl.cpp:137-159 -> vp_malloc_* / vp_copy_*
#include "vp_common.h"

#include <cstring>

static thread_local char g_err[512] = "";

void vp_set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

void vp_clear_error() { g_err[0] = 0; }
extern "C" const char* vp_last_error(void) { return g_err; }

extern "C" int vp_device_count(void) {
	int n = 0;
	if(hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

static int ctx_init(int device, VpContext* ctx) {
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if(e != hipSuccess || n <= 0) {
		vp_set_error("no HIP device available (%s): this library has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "0 devices");
		return VP_ERR_NODEVICE;
	}
	VP_REQUIRE(device >= 0 && device < n, "device %d out of range [0,%d)", device, n);
	VP_CHECK_HIP(hipSetDevice(device));
	hipDeviceProp_t prop;
	VP_CHECK_HIP(hipGetDeviceProperties(&prop, device));
	ctx->device = device;
	ctx->device_name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
	ctx->cu_count = prop.multiProcessorCount;
	return VP_OK;
}

extern "C" int vp_ctx_create(int device, int n_streams, VpContext** out) {
	VP_REQUIRE(out, "out is null");
	VP_REQUIRE(n_streams >= 1 && n_streams <= VP_MAX_STREAMS, "n_streams %d out of range [1,%d]", n_streams, VP_MAX_STREAMS);
	VpContext* ctx = new VpContext();
	int rc = ctx_init(device, ctx);
	if(rc != VP_OK) {
		delete ctx;
		return rc;
	}
	ctx->n_streams = n_streams;
	for(int i = 0; i < n_streams; i++) {
		if(hipStreamCreateWithFlags(&ctx->streams[i], hipStreamNonBlocking) != hipSuccess) {
			vp_set_error("hipStreamCreate failed");
			delete ctx;
			return VP_ERR_HIP;
		}
	}
	*out = ctx;
	return VP_OK;
}

extern "C" int vp_ctx_adopt_stream(int device, void* hip_stream, VpContext** out) {
	VP_REQUIRE(out, "out is null");
	VpContext* ctx = new VpContext();
	int rc = ctx_init(device, ctx);
	if(rc != VP_OK) {
		delete ctx;
		return rc;
	}
	ctx->n_streams = 1;
	ctx->own_streams = false;
	ctx->streams[0] = (hipStream_t)hip_stream;
	*out = ctx;
	return VP_OK;
}

extern "C" void vp_ctx_destroy(VpContext* ctx) {
	if(!ctx) return;
	(void)hipSetDevice(ctx->device);
	for(int i = 0; i < ctx->n_streams; i++) {
		(void)hipStreamSynchronize(ctx->streams[i]);
		for(int k = 0; k < VP_WS_SLOTS; k++)
			if(ctx->ws[k][i].ptr) (void)hipFree(ctx->ws[k][i].ptr);
		if(ctx->host_flag[i]) (void)hipHostFree(ctx->host_flag[i]);
		if(ctx->compact_arrived[i]) (void)hipFree(ctx->compact_arrived[i]);
		if(ctx->own_streams) (void)hipStreamDestroy(ctx->streams[i]);
	}
	for(auto& e : ctx->events) {
		(void)hipEventDestroy(e.start);
		(void)hipEventDestroy(e.stop);
	}
	vp_frame_graphs_clear(ctx);
	for(VpImage* im : ctx->image_pool) {
		if(im->buf.dev) (void)hipFree(im->buf.dev);
		if(im->buf.host) (void)hipHostFree(im->buf.host);
		delete im;
	}
	delete ctx;
}

extern "C" int vp_abi_version(void) { return VP_ABI_VERSION; }
extern "C" size_t vp_abi_sizeof(const char* name) {
	if(!name) return 0;
	const std::string n(name);
	if(n == "VpPlacement") return sizeof(VpPlacement);
	if(n == "VpFrameParams") return sizeof(VpFrameParams);
	if(n == "VpFrameBuffers") return sizeof(VpFrameBuffers);
	if(n == "VpCameraModel") return sizeof(VpCameraModel);
	if(n == "VpMatch") return sizeof(VpMatch);
	if(n == "VpStageTime") return sizeof(VpStageTime);
	return 0;
}

extern "C" const char* vp_device_name(VpContext* ctx) { return ctx ? ctx->device_name.c_str() : ""; }

extern "C" int vp_ctx_tune(VpContext* ctx, const char* key, int value) {
	VP_REQUIRE(ctx && key, "vp_ctx_tune: null argument");
	const std::string k(key);
	if(k == "sat_workgroups_per_cu") {
		VP_REQUIRE(value == 0 || (value >= 2 && value <= 8), "vp_ctx_tune: sat_workgroups_per_cu %d: 0 (automatic) or 2..8", value);
		ctx->tune.sat_workgroups_per_cu = value;
	} else if(k == "resample_xcd_bands") {
		VP_REQUIRE(value >= -1 && value <= 1, "vp_ctx_tune: resample_xcd_bands takes -1 (automatic), 0 or 1");
		ctx->tune.resample_xcd_bands = value;
	} else if(k == "resample_groups_per_wg") {
		VP_REQUIRE(value >= 0 && value <= 64, "vp_ctx_tune: resample_groups_per_wg %d: 0 (automatic) or 1..64", value);
		ctx->tune.resample_groups_per_wg = value;
	} else if(k == "resample_float_stage") {
		VP_REQUIRE(value >= 0 && value <= 2, "vp_ctx_tune: resample_float_stage %d: 0 (bytes), 1 (floats) or 2 (floats, two frames side by side)", value);
		ctx->tune.resample_float_stage = value;
	} else if(k == "resample_bgr_staged") {
		ctx->tune.resample_bgr_staged = value != 0;
	} else if(k == "candidate_capacity") {
		VP_REQUIRE(value >= 0, "vp_ctx_tune: candidate_capacity must be >= 0 (0 = default)");
		ctx->tune.candidate_capacity = value;
	} else if(k == "sat_halo_strips") {
		VP_REQUIRE(value == 0 || value == 1, "vp_ctx_tune: sat_halo_strips is 0 or 1");
		ctx->tune.sat_halo_strips = value;
	} else if(k == "frame_graph") {
		VP_REQUIRE(value >= 0 && value <= 2, "vp_ctx_tune: frame_graph is 0, 1 or 2");
		ctx->tune.frame_graph = value;
		if(!value) vp_frame_graphs_clear(ctx);
	} else if(k == "placement_max_mib") {
		VP_REQUIRE(value >= -1, "vp_ctx_tune: placement_max_mib: MiB, 0 (default rule) or -1 (this process owns the GPU)");
		ctx->tune.placement_max_mib = value;
	} else if(k == "placement_candidates") {
		VP_REQUIRE(value >= 0 && value <= 256, "vp_ctx_tune: placement_candidates out of range [0,256]");
		ctx->tune.placement_candidates = value;
	} else if(k == "tail_emit_kernel") {
		VP_REQUIRE(value == 0 || value == 1, "vp_ctx_tune: tail_emit_kernel is 0 or 1");
		ctx->tune.tail_emit_kernel = value;
	} else if(k == "sat_spin_limit") {
		VP_REQUIRE(value >= 0, "vp_ctx_tune: sat_spin_limit: polls per look-back wait, 0 = default (2^22)");
		ctx->tune.sat_spin_limit = value;
	} else if(k == "sat_debug_drop_tile") {
		VP_REQUIRE(value >= -1, "vp_ctx_tune: sat_debug_drop_tile: a tile index of the launch, or -1 (off)");
		ctx->tune.sat_debug_drop_tile = value;
	} else if(k == "tail_segment_rows") {
		VP_REQUIRE(value >= 0 && value <= 4096, "vp_ctx_tune: tail_segment_rows out of range [0,4096]");
		ctx->tune.tail_segment_rows = value;
	} else {
		vp_set_error("vp_ctx_tune: unknown key '%s'", key);
		return VP_ERR_INVALID;
	}
	return VP_OK;
}

// ---- ordering against other streams: the cl::Event of OpenCL::run (opencl.h:76-86) as a device-side dependency -------------------------------------
// The context's own streams are non-blocking: nothing orders them against a stream the caller fills its buffers on, or reads the results on.  Both
// directions are one event: recorded on the producing stream, waited for by the consuming one; the host never blocks.
static int stream_edge(VpContext* ctx, hipStream_t producer, int producer_device, hipStream_t consumer, const char* what) {
	VP_REQUIRE(!(producer == consumer), "%s: both sides are the same stream (already ordered)", what);
	VP_CHECK_HIP(hipSetDevice(producer_device)); // an event is recorded on a stream of the device it was created on
	hipEvent_t ev = nullptr;
	hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
	if(e == hipSuccess) {
		e = hipEventRecord(ev, producer);
		if(e == hipSuccess) e = hipStreamWaitEvent(consumer, ev, 0);
		(void)hipEventDestroy(ev); // released by the runtime once the recorded work has completed
	}
	(void)hipSetDevice(ctx->device); // whatever happened: the calling thread is left on the context's device
	if(e != hipSuccess) {
		vp_set_error("%s failed: %s", what, hipGetErrorString(e));
		return VP_ERR_HIP;
	}
	return VP_OK;
}
static int foreign_stream_device(VpContext* ctx, hipStream_t s, int* dev) {
	*dev = ctx->device; // the legacy default stream (NULL) belongs to the calling thread's current device: callers pass a real stream for another GPU
	if(s) VP_CHECK_HIP(hipStreamGetDevice(s, dev));
	return VP_OK;
}

extern "C" int vp_stream_wait_for(VpContext* ctx, int stream, void* producer_stream) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(!ctx->capturing[stream], "vp_stream_wait_for: the stream is being captured");
	if((hipStream_t)producer_stream == ctx->streams[stream]) return VP_OK; // an adopted stream is ordered with itself
	int dev = 0;
	VP_TRY(foreign_stream_device(ctx, (hipStream_t)producer_stream, &dev));
	return stream_edge(ctx, (hipStream_t)producer_stream, dev, ctx->streams[stream], "vp_stream_wait_for");
}

extern "C" int vp_stream_release_to(VpContext* ctx, int stream, void* consumer_stream) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(!ctx->capturing[stream], "vp_stream_release_to: the stream is being captured");
	if((hipStream_t)consumer_stream == ctx->streams[stream]) return VP_OK;
	return stream_edge(ctx, ctx->streams[stream], ctx->device, (hipStream_t)consumer_stream, "vp_stream_release_to");
}

// a raised flag means a bounded look-back wait of the single-pass SAT expired: the results of that launch are invalid
int vp_check_timeout(VpContext* ctx, int stream) {
	int* f = ctx->host_flag[stream];
	if(f && *reinterpret_cast<volatile int*>(f) != 0) {
		vp_set_error("single-pass SAT: a tile carry never arrived (bounded look-back wait expired) on stream %d", stream);
		return VP_ERR_TIMEOUT;
	}
	return VP_OK;
}

extern "C" int vp_sync(VpContext* ctx, int stream) {
	VP_REQUIRE(ctx, "null context");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	if(stream < 0) {
		for(int i = 0; i < ctx->n_streams; i++) VP_CHECK_HIP(hipStreamSynchronize(ctx->streams[i]));
		for(int i = 0; i < ctx->n_streams; i++) VP_TRY(vp_check_timeout(ctx, i));
		return VP_OK;
	}
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_CHECK_HIP(hipStreamSynchronize(ctx->streams[stream]));
	return vp_check_timeout(ctx, stream);
}

extern "C" int vp_flush(VpContext* ctx, int stream) {
	VP_REQUIRE(ctx, "null context");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	for(int i = 0; i < ctx->n_streams; i++) {
		if(stream >= 0 && i != stream) continue;
		const hipError_t e = hipStreamQuery(ctx->streams[i]);
		if(e != hipSuccess && e != hipErrorNotReady) {
			vp_set_error("hipStreamQuery failed: %s", hipGetErrorString(e));
			return VP_ERR_HIP;
		}
		(void)hipGetLastError(); // hipErrorNotReady is the expected answer, not an error to leave behind
	}
	if(stream >= ctx->n_streams) return vp_check_ctx(ctx, stream);
	return VP_OK;
}

int vp_workspace(VpContext* ctx, int stream, int slot, size_t bytes, void** out, bool* grew) {
	VpWorkspace& w = ctx->ws[slot][stream];
	if(grew) *grew = false;
	if(w.bytes < bytes) {
		if(ctx->capturing[stream]) {
			vp_set_error("a workspace would have to grow inside a graph capture");
			return VP_ERR_INVALID;
		}
		ctx->ws_generation[stream]++;
		if(grew) *grew = true;
		VP_CHECK_HIP(hipSetDevice(ctx->device)); // hipMalloc / hipFree act on the current device, which a multi-context caller may have moved
		VP_CHECK_HIP(hipStreamSynchronize(ctx->streams[stream]));
		if(w.ptr) VP_CHECK_HIP(hipFree(w.ptr));
		w.ptr = nullptr;
		w.bytes = 0;
		const size_t want = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
		hipError_t e = hipMalloc(&w.ptr, want);
		if(e != hipSuccess) {
			vp_set_error("workspace hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
			return VP_ERR_NOMEM;
		}
		w.bytes = want;
		// a workspace on another device would be read over xGMI on every launch, silently: a caller that drives several contexts from one thread
		// (host/feeds_multi.cpp in round 2) is exactly how that happens, and a one-GPU box cannot show it -- so check where the block landed
		hipPointerAttribute_t attr;
		if(hipPointerGetAttributes(&attr, w.ptr) == hipSuccess && attr.device != ctx->device) {
			vp_set_error("workspace of a context on device %d was allocated on device %d", ctx->device, attr.device);
			(void)hipFree(w.ptr); // do not keep (and later hand out) the misplaced, uninitialised block
			w.ptr = nullptr;
			w.bytes = 0;
			return VP_ERR_HIP;
		}
		VP_CHECK_HIP(hipMemsetAsync(w.ptr, 0, want, ctx->streams[stream]));
	}
	*out = w.ptr;
	return VP_OK;
}

// ---- events ---------------------------------------------------------------------------------------------------------
extern "C" int vp_events_enable(VpContext* ctx, int on) {
	VP_REQUIRE(ctx, "null context");
	ctx->events_on = on != 0;
	return VP_OK;
}

extern "C" int vp_event_times(VpContext* ctx, VpStageTime* out, int cap) {
	VP_REQUIRE(ctx && (out || cap == 0), "bad arguments");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	for(int i = 0; i < ctx->n_streams; i++) VP_CHECK_HIP(hipStreamSynchronize(ctx->streams[i]));
	std::lock_guard<std::mutex> g(ctx->mu);
	int n = 0;
	for(auto& e : ctx->events) {
		if(n >= cap) break;
		float ms = 0.f;
		VP_CHECK_HIP(hipEventElapsedTime(&ms, e.start, e.stop));
		out[n].stage = e.stage;
		out[n].ms = ms;
		n++;
	}
	return n;
}

extern "C" int vp_events_clear(VpContext* ctx) {
	VP_REQUIRE(ctx, "null context");
	std::lock_guard<std::mutex> g(ctx->mu);
	for(auto& e : ctx->events) {
		(void)hipEventDestroy(e.start);
		(void)hipEventDestroy(e.stop);
	}
	ctx->events.clear();
	return VP_OK;
}

// ---- memory ---------------------------------------------------------------------------------------------------------
extern "C" int vp_malloc_device(VpContext* ctx, size_t bytes, void** dptr) {
	VP_REQUIRE(ctx && dptr, "bad arguments");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
	if(e != hipSuccess) {
		vp_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
		return VP_ERR_NOMEM;
	}
	return VP_OK;
}

extern "C" int vp_free_device(VpContext* ctx, void* dptr) {
	VP_REQUIRE(ctx, "null context");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	if(dptr) VP_CHECK_HIP(hipFree(dptr));
	return VP_OK;
}

extern "C" int vp_malloc_host(VpContext* ctx, size_t bytes, void** hptr) {
	VP_REQUIRE(ctx && hptr, "bad arguments");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault);
	if(e != hipSuccess) {
		vp_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
		return VP_ERR_NOMEM;
	}
	return VP_OK;
}

extern "C" int vp_free_host(VpContext* ctx, void* hptr) {
	VP_REQUIRE(ctx, "null context");
	if(hptr) VP_CHECK_HIP(hipHostFree(hptr));
	return VP_OK;
}

// ---- buffer and image objects: CLArray / RawImage / CLImage and their blocking maps (opencl.h:115-262, opencl.cpp:108-159) --------
static int buffer_alloc(VpContext* ctx, VpBuffer& b, size_t bytes) {
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	b.bytes = bytes;
	hipError_t e = hipMalloc(&b.dev, bytes ? bytes : 1);
	if(e != hipSuccess) {
		vp_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
		return VP_ERR_NOMEM;
	}
	return VP_OK;
}
static int buffer_map(VpContext* ctx, int stream, VpBuffer& b, int read_back, void** host) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(host, "null pointer");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	std::lock_guard<std::mutex> g(b.mu);
	if(!b.host) {
		hipError_t e = hipHostMalloc(&b.host, b.bytes ? b.bytes : 1, hipHostMallocDefault);
		if(e != hipSuccess) {
			vp_set_error("hipHostMalloc(%zu) failed: %s", b.bytes, hipGetErrorString(e));
			return VP_ERR_NOMEM;
		}
	}
	if(read_back) VP_CHECK_HIP(hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, ctx->streams[stream]));
	VP_CHECK_HIP(hipStreamSynchronize(ctx->streams[stream])); // blocking, like clEnqueueMapBuffer(..., CL_TRUE, ...)
	*host = b.host;
	return VP_OK;
}
static int buffer_unmap(VpContext* ctx, int stream, VpBuffer& b, int write_back) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	std::lock_guard<std::mutex> g(b.mu);
	if(write_back) {
		VP_REQUIRE(b.host, "unmap of a buffer that was never mapped");
		VP_CHECK_HIP(hipMemcpyAsync(b.dev, b.host, b.bytes, hipMemcpyHostToDevice, ctx->streams[stream]));
		VP_CHECK_HIP(hipStreamSynchronize(ctx->streams[stream]));
	}
	return VP_OK;
}

extern "C" int vp_buffer_create(VpContext* ctx, size_t bytes, const void* init, VpBuffer** out) {
	VP_REQUIRE(ctx && out, "bad arguments");
	VpBuffer* b = new VpBuffer();
	int rc = buffer_alloc(ctx, *b, bytes);
	if(rc == VP_OK && init && bytes) { // CL_MEM_COPY_HOST_PTR (opencl.cpp:147)
		if(hipMemcpy(b->dev, init, bytes, hipMemcpyHostToDevice) != hipSuccess) {
			vp_set_error("initial copy of %zu bytes failed", bytes);
			rc = VP_ERR_HIP;
		}
	}
	if(rc != VP_OK) {
		if(b->dev) (void)hipFree(b->dev);
		delete b;
		return rc;
	}
	*out = b;
	return VP_OK;
}
extern "C" void* vp_buffer_device(const VpBuffer* b) { return b ? b->dev : nullptr; }
extern "C" size_t vp_buffer_size(const VpBuffer* b) { return b ? b->bytes : 0; }
extern "C" int vp_buffer_map(VpContext* ctx, int stream, VpBuffer* b, int read_back, void** host) {
	VP_REQUIRE(b, "null buffer");
	return buffer_map(ctx, stream, *b, read_back, host);
}
extern "C" int vp_buffer_unmap(VpContext* ctx, int stream, VpBuffer* b, int write_back) {
	VP_REQUIRE(b, "null buffer");
	return buffer_unmap(ctx, stream, *b, write_back);
}
extern "C" int vp_buffer_destroy(VpContext* ctx, VpBuffer* b) {
	VP_REQUIRE(ctx, "null context");
	if(!b) return VP_OK;
	(void)hipSetDevice(ctx->device);
	if(b->dev) (void)hipFree(b->dev);
	if(b->host) (void)hipHostFree(b->host);
	delete b;
	return VP_OK;
}

static size_t pixel_bytes(int format) { // PixelFormat::pixelSize() (opencl.cpp:24-27); NV12 is over-allocated like the reference's
	return format == VP_FMT_U8 ? 1 : (format == VP_FMT_NV12 ? 2 : 4);
}
extern "C" int vp_image_acquire(VpContext* ctx, int format, int w, int h, VpImage** out) {
	VP_REQUIRE(ctx && out && w > 0 && h > 0 && format >= VP_FMT_RGBA8 && format <= VP_FMT_NV12, "vp_image_acquire: bad arguments");
	std::lock_guard<std::mutex> g(ctx->mu);
	for(VpImage* im : ctx->image_pool)
		if(!im->in_use && im->format == format && im->w == w && im->h == h) {
			im->in_use = true;
			*out = im;
			return VP_OK;
		}
	VpImage* im = new VpImage();
	const int rc = buffer_alloc(ctx, im->buf, (size_t)w * h * pixel_bytes(format));
	if(rc != VP_OK) {
		delete im;
		return rc;
	}
	im->format = format;
	im->w = w;
	im->h = h;
	im->in_use = true;
	ctx->image_pool.push_back(im);
	*out = im;
	return VP_OK;
}
extern "C" int vp_image_release(VpContext* ctx, VpImage* im) {
	VP_REQUIRE(ctx, "null context");
	if(im) {
		std::lock_guard<std::mutex> g(ctx->mu);
		im->in_use = false;
	}
	return VP_OK;
}
extern "C" void* vp_image_device(const VpImage* im) { return im ? im->buf.dev : nullptr; }
extern "C" int vp_image_info(const VpImage* im, int* format, int* w, int* h, size_t* bytes) {
	VP_REQUIRE(im, "null image");
	if(format) *format = im->format;
	if(w) *w = im->w;
	if(h) *h = im->h;
	if(bytes) *bytes = im->buf.bytes;
	return VP_OK;
}
extern "C" int vp_image_map(VpContext* ctx, int stream, VpImage* im, int read_back, void** host) {
	VP_REQUIRE(im, "null image");
	return buffer_map(ctx, stream, im->buf, read_back, host);
}
extern "C" int vp_image_unmap(VpContext* ctx, int stream, VpImage* im, int write_back) {
	VP_REQUIRE(im, "null image");
	return buffer_unmap(ctx, stream, im->buf, write_back);
}

extern "C" int vp_copy_h2d(VpContext* ctx, int stream, void* dst_d, const void* src_h, size_t bytes) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VP_CHECK_HIP(hipMemcpyAsync(dst_d, src_h, bytes, hipMemcpyHostToDevice, ctx->streams[stream]));
	return VP_OK;
}

extern "C" int vp_copy_d2h(VpContext* ctx, int stream, void* dst_h, const void* src_d, size_t bytes) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VP_CHECK_HIP(hipMemcpyAsync(dst_h, src_d, bytes, hipMemcpyDeviceToHost, ctx->streams[stream]));
	return VP_OK;
}

extern "C" int vp_copy_d2h_sync(VpContext* ctx, int stream, void* dst_h, const void* src_d, size_t bytes) {
	VP_TRY(vp_copy_d2h(ctx, stream, dst_h, src_d, bytes));
	VP_CHECK_HIP(hipStreamSynchronize(ctx->streams[stream]));
	return VP_OK;
}

extern "C" int vp_memset_device(VpContext* ctx, int stream, void* dst_d, int value, size_t bytes) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VP_CHECK_HIP(hipMemsetAsync(dst_d, value, bytes, ctx->streams[stream]));
	return VP_OK;
}
