// See opencl.h.  Implements the facade over the C ABI of libvp_hip.so (include/vp_hip.h).
#include "opencl.h"
#include "vp_host.h"

const PixelFormat PixelFormat::RGBA8(4, 1, true, -1, "");
const PixelFormat PixelFormat::U8(1, 1, false, -1, "");
const PixelFormat PixelFormat::F32(4, 1, false, -1, "");
const PixelFormat PixelFormat::NV12(1, 2, true, -1, "");
const PixelFormat PixelFormat::RGGB8(2, 2, true, VP_LAYOUT_RGGB, "-DRGGB");
const PixelFormat PixelFormat::GRBG8(2, 2, true, VP_LAYOUT_GRBG, "-DGRBG");
const PixelFormat PixelFormat::BGR8(3, 1, true, VP_LAYOUT_BGR, "-DBGR");

VpContext* OpenCL::ctx = nullptr;

OpenCL::OpenCL() { // opencl.cpp:34-50: first GPU, one in-order queue with profiling
	if(ctx) FATAL("only one OpenCL object per process (the reference installs a process-wide default context, opencl.cpp:46-49)");
	VP_OR_FATAL(vp_ctx_create(0, 1, &ctx));
	VP_OR_FATAL(vp_events_enable(ctx, 1));
	LOG("Using device: HIP » " << vp_device_name(ctx));
}

OpenCL::~OpenCL() {
	nv12pool.clear();
	if(satTable) vp_free_device(ctx, satTable);
	vp_ctx_destroy(ctx);
	ctx = nullptr;
}

cl::Kernel OpenCL::compile(const char* code, const std::string& options) { // opencl.cpp:67-85
	cl::Kernel k;
	const std::string src(code ? code : "");
	auto has = [&](const char* s) { return src.find(s) != std::string::npos; };
	if(has("quad2rgba")) k.id = cl::KernelId::quad2rgba; // debug / stream colour kernels (Resources.cpp:127-130)
	else if(has("quad2nv12")) k.id = cl::KernelId::quad2nv12;
	else if(has("rgb2nv12")) k.id = cl::KernelId::rgb2nv12;
	else if(has("f2nv12")) k.id = cl::KernelId::f2nv12;
	else if(has("raw2quad")) k.id = cl::KernelId::raw2quad;
	else if(has("resampling")) k.id = cl::KernelId::resampling;
	else if(has("gradient_dotproduct")) k.id = cl::KernelId::gradientDot;
	else if(has("sat_horizontal")) k.id = cl::KernelId::satHorizontal;
	else if(has("sat_vertical")) k.id = cl::KernelId::satVertical;
	else if(has("circularize")) k.id = cl::KernelId::blobCenter;
	else if(has("vp:blobScore") || (has("matches") && has("image2d_t score"))) k.id = cl::KernelId::blobScore;
	else if(has("vp:blobList") || has("matches")) k.id = cl::KernelId::blobList;
	else if(has("circle")) k.id = cl::KernelId::satBlobCenter;
	else FATAL("Kernel missing: " << src.substr(0, 60));
	if(options.find("-DGRBG") != std::string::npos) k.layout = VP_LAYOUT_GRBG;
	else if(options.find("-DBGR") != std::string::npos) k.layout = VP_LAYOUT_BGR;
	return k;
}

void OpenCL::wait(const cl::Event& event) { VP_OR_FATAL(vp_sync(event.ctx ? event.ctx : ctx, event.stream)); }

void OpenCL::printRuntimes() { // opencl.cpp:94-101
	VpStageTime t[64];
	const int n = vp_event_times(ctx, t, 64);
	if(n < 0) FATAL("vp_event_times: " << vp_last_error());
	std::cout << std::fixed;
	std::cout.precision(2);
	for(int i = 0; i < n; i++) std::cout << t[i].ms << "ms ";
	std::cout << std::endl;
}

void OpenCL::clearEvents() { VP_OR_FATAL(vp_events_clear(ctx)); }

std::shared_ptr<CLImage> OpenCL::acquire(const PixelFormat* format, int width, int height, const std::string& name) { // opencl.cpp:108-122
	return std::make_shared<CLImage>(format, width, height, name); // the pool is vp_image_acquire's: an idle image of this format and size is reused
}

std::shared_ptr<RawImage> OpenCL::acquireNV12(int width, int height) { // opencl.cpp:124-135
	for(auto& i : nv12pool)
		if(i.use_count() == 1 && i->width == width && i->height == height) return i;
	auto array = std::make_shared<RawImage>(&PixelFormat::NV12, width, height);
	nv12pool.push_back(array);
	return array;
}

// The reference launches gradientDot, satHorizontal, satVertical, circle and matches one by one (Resources.cpp:160-163,
// main.cpp:289).  Here gradientDot+SAT is one fused launch into a private int32 table, issued when satVertical arrives; the
// F32 gradDot / blobCenter images stay materialised because the caller gets them back (debug views, blob_benchmark.cpp:162).
void OpenCL::dispatch(const cl::Kernel& k, const cl::EnqueueArgs& e, const std::vector<Arg>& a) {
	auto need = [&](size_t n) { if(a.size() != n) FATAL("Enqueue kernel error: expected " << n << " arguments, got " << a.size()); };
	switch(k.id) {
	case cl::KernelId::raw2quad: { // (buffer img, 4 x U8 image) over NDRange(qw, qh): Resources.cpp:142
		need(5);
		VP_OR_FATAL(vp_raw2quad4(ctx, 0, (const uint8_t*)a[0].p, k.layout, (int)e.global.d[0], (int)e.global.d[1], (uint8_t*)a[1].p, (uint8_t*)a[2].p, (uint8_t*)a[3].p, (uint8_t*)a[4].p));
		break;
	}
	case cl::KernelId::resampling: { // Resources.cpp:159
		need(10);
		VP_OR_FATAL(vp_resample4(ctx, 0, (const uint8_t*)a[0].p, (const uint8_t*)a[1].p, (const uint8_t*)a[2].p, (const uint8_t*)a[3].p, k.layout, a[0].w, a[0].h, &a[5].m,
				a[6].f, a[7].f, a[8].f, a[9].f, samplerMode, (int)e.global.d[0], (int)e.global.d[1], (uint8_t*)a[4].p));
		break;
	}
	case cl::KernelId::gradientDot: // Resources.cpp:160 -- remembered, launched fused with the SAT
		need(3);
		pendingGrad.flat = a[0].p;
		pendingGrad.graddot = a[1].p;
		pendingGrad.w = a[0].w;
		pendingGrad.h = a[0].h;
		pendingGrad.offset = a[2].i;
		break;
	case cl::KernelId::satHorizontal: // Resources.cpp:161 -- first half of the SAT: nothing to do yet
		need(2);
		break;
	case cl::KernelId::satVertical: { // Resources.cpp:162 -- gradientDot + both SAT passes, one launch
		need(2);
		if(!pendingGrad.flat) FATAL("sat_vertical without a preceding gradient_dotproduct");
		const size_t bytes = sizeof(int32_t) * (size_t)pendingGrad.w * pendingGrad.h;
		if(satBytes < bytes) {
			if(satTable) VP_OR_FATAL(vp_free_device(ctx, satTable));
			VP_OR_FATAL(vp_malloc_device(ctx, bytes, &satTable));
			satBytes = bytes;
		}
		VP_OR_FATAL(vp_graddot_sat(ctx, 0, (const uint8_t*)pendingGrad.flat, pendingGrad.w, pendingGrad.h, pendingGrad.offset, 1, satVariant, (int32_t*)satTable, (float*)pendingGrad.graddot));
		break;
	}
	case cl::KernelId::satBlobCenter: // Resources.cpp:163 -- reads the private table, whatever image the caller passed as `sat`
		need(3);
		VP_OR_FATAL(vp_blobcenter(ctx, 0, (const int32_t*)satTable, a[1].w, a[1].h, a[2].i, 1, (float*)a[1].p));
		break;
	case cl::KernelId::blobList: // main.cpp:289
		need(8);
		VP_OR_FATAL(vp_bloblist(ctx, 0, (const uint8_t*)a[0].p, (const float*)a[1].p, a[0].w, a[0].h, 1, a[4].f, a[5].f, a[6].i, a[7].i, listOrder, (VpMatch*)a[2].p, nullptr, (int32_t*)a[3].p));
		break;
	case cl::KernelId::blobScore: // blob_benchmark.cpp:154-155
		need(5);
		VP_OR_FATAL(vp_blobscore(ctx, 0, (const uint8_t*)a[0].p, (const float*)a[1].p, a[0].w, a[0].h, 1, a[3].f, a[4].i, (float*)a[2].p));
		break;
	case cl::KernelId::blobCenter:
		need(4);
		VP_OR_FATAL(vp_circularize(ctx, 0, (const float*)a[0].p, a[0].w, a[0].h, 1, a[2].i, a[3].i, (float*)a[1].p));
		break;
	case cl::KernelId::quad2rgba: // Resources.cpp:147 -- (4 x U8 image, RGBA8 image) over NDRange(qw, qh)
		need(5);
		VP_OR_FATAL(vp_quad2rgba4(ctx, 0, (const uint8_t*)a[0].p, (const uint8_t*)a[1].p, (const uint8_t*)a[2].p, (const uint8_t*)a[3].p, k.layout, a[0].w, a[0].h, samplerMode, (uint8_t*)a[4].p));
		break;
	case cl::KernelId::quad2nv12: // Resources.cpp:168 -- (4 x U8 image, NV12 buffer)
		need(5);
		VP_OR_FATAL(vp_quad2nv124(ctx, 0, (const uint8_t*)a[0].p, (const uint8_t*)a[1].p, (const uint8_t*)a[2].p, (const uint8_t*)a[3].p, k.layout, a[0].w, a[0].h, samplerMode, (uint8_t*)a[4].p));
		break;
	case cl::KernelId::rgb2nv12: // Resources.cpp:184 -- (RGBA8 image, NV12 buffer)
		need(2);
		VP_OR_FATAL(vp_rgba2nv12(ctx, 0, (const uint8_t*)a[0].p, a[0].w, a[0].h, 1, (uint8_t*)a[1].p));
		break;
	case cl::KernelId::f2nv12: // Resources.cpp:184 -- (F32 image, NV12 buffer)
		need(2);
		VP_OR_FATAL(vp_f2nv12(ctx, 0, (const float*)a[0].p, a[0].w, a[0].h, 1, (uint8_t*)a[1].p));
		break;
	default:
		FATAL("Enqueue kernel error: kernel not on the hot path");
	}
}

// ---- memory objects ------------------------------------------------------------------------------------------------------
static std::shared_ptr<VpBuffer> bufferCreate(size_t bytes, const void* init) {
	VpBuffer* b = nullptr;
	if(vp_buffer_create(OpenCL::context(), bytes, init, &b) != VP_OK) FATAL("Error during buffer allocation: " << vp_last_error());
	return std::shared_ptr<VpBuffer>(b, [](VpBuffer* q) { if(OpenCL::context()) vp_buffer_destroy(OpenCL::context(), q); });
}

CLArray::CLArray(int size): buffer{nullptr}, size(size), object(bufferCreate((size_t)size, nullptr)) { const_cast<cl::Buffer&>(buffer).dptr = vp_buffer_device(object.get()); }
CLArray::CLArray(void* data, int size): buffer{nullptr}, size(size), object(bufferCreate((size_t)size, data)) { // CL_MEM_COPY_HOST_PTR (opencl.cpp:147)
	const_cast<cl::Buffer&>(buffer).dptr = vp_buffer_device(object.get());
}

// opencl.cpp:161-179.  RGBA8 -> the three colour channels (the reference converts RGBA to BGR for cv::imwrite, which stores RGB); F32 -> grey through
// factor / offset; anything else (U8 planes) -> grey as it is.
void CLImage::save(const std::string& suffix, float factor, float offset) const {
	const std::string path = "img/" + name + suffix;
	const size_t n = (size_t)width * height;
	std::vector<uint8_t> px;
	int channels = 1;
	if(format == &PixelFormat::F32) {
		CLImageMap<float> map = read<float>();
		px.resize(n);
		vph_f32_to_grey(*map, n, factor, offset, px.data());
	} else if(format == &PixelFormat::RGBA8) {
		CLImageMap<uint8_t> map = read<uint8_t>();
		channels = 3;
		px.resize(3 * n);
		for(size_t i = 0; i < n; i++)
			for(int c = 0; c < 3; c++) px[3 * i + c] = map[4 * i + c];
	} else {
		CLImageMap<uint8_t> map = read<uint8_t>();
		px.assign(*map, *map + n); // U8 planes (the only other image format the pool hands out)
	}
	if(vph_write_png(path.c_str(), width, height, channels, px.data()) != VP_OK) WARN("could not write " << path);
}

CLImage::CLImage(const PixelFormat* format): format(format), width(0), height(0) {}
CLImage::CLImage(const PixelFormat* format, int width, int height, std::string name): format(format), width(width), height(height), name(std::move(name)) {
	VpImage* im = nullptr;
	if(vp_image_acquire(OpenCL::context(), format->imageFormat(), width, height, &im) != VP_OK) FATAL("Error during image allocation: " << vp_last_error());
	pooled = std::shared_ptr<VpImage>(im, [](VpImage* q) { if(OpenCL::context()) vp_image_release(OpenCL::context(), q); });
	image.dptr = vp_image_device(im);
	image.width = width;
	image.height = height;
}
