// HIP-backed host facade with the call surface of the reference's src/opencl.h (classes OpenCL, CLArray, CLMap, RawImage,
// CLImage, CLImageMap, PixelFormat) so that the per-frame loop of src/main.cpp:262-317 and src/Resources.cpp:138-164 can be
// written against it unchanged.  Nothing here compiles or ships OpenCL: a `cl::Kernel` is a tag naming one entry point of
// the C ABI (include/vp_hip.h), `OpenCL::run` forwards the argument pack to it, images and buffers are plain HIP memory.
//
// Differences a maintainer will notice (all deliberate):
//   * CLImageMap has no `cv::Mat cv` member (OpenCV is not a dependency of the hot path); rowPitch == width.  It is the ONE member of the
//     reference's opencl.h this header lacks: tests/test_reference_callsites.py compiles the reference's own text of Resources.cpp:138-164 and
//     the GPU statements of main.cpp:253-317 against this header.
//   * map()/unmap() are a device<->pinned-host copy on the context stream (the reference's blocking clEnqueueMapBuffer).
//   * the intermediate SAT is an int32 table owned by Resources::rgba2blobCenter, never exposed (the reference keeps its
//     two F32 SAT images private to that function as well, Resources.cpp:155-156).
// Errors follow the reference: any failure is FATAL = message on stderr + exit(1) (src/log.h:21).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "vp_hip.h"

#define LOG(msg) do { std::cout << "[LOG] " << msg << std::endl; } while(0)
#define WARN(msg) do { std::cerr << "[WARN] " << __FILE__ << ":" << __LINE__ << " " << msg << std::endl; } while(0)
#define FATAL(msg) do { std::cerr << "[FATAL] " << __FILE__ << ":" << __LINE__ << " " << msg << std::endl; exit(1); } while(0)
#define VP_OR_FATAL(expr) do { if((expr) != VP_OK) FATAL(#expr << ": " << vp_last_error()); } while(0)

typedef VpCameraModel CLCameraModel; // Perspective.h:22-29

// ---- the sliver of the cl:: namespace the call sites use ------------------------------------------------------------------
namespace cl {
enum class KernelId { none, raw2quad, resampling, gradientDot, satHorizontal, satVertical, satBlobCenter, blobList, blobScore, blobCenter, quad2rgba, quad2nv12, rgb2nv12, f2nv12 };
struct Kernel {
	KernelId id = KernelId::none;
	int layout = VP_LAYOUT_RGGB; // from the -DRGGB / -DGRBG / -DBGR compile option (opencl.cpp:29-31)
};
struct NDRange {
	size_t d[2] = {1, 1};
	NDRange() = default;
	explicit NDRange(size_t x) { d[0] = x; }
	NDRange(size_t x, size_t y) { d[0] = x; d[1] = y; }
};
struct Event {
	VpContext* ctx = nullptr;
	int stream = 0;
};
struct EnqueueArgs {
	NDRange global;
	explicit EnqueueArgs(NDRange g): global(g) {}
	EnqueueArgs(const Event&, NDRange g): global(g) {} // one in-order stream: the explicit event chain is redundant (opencl.cpp:48)
};
struct Buffer { void* dptr = nullptr; };
struct Image2D { void* dptr = nullptr; int width = 0, height = 0; };
} // namespace cl

class PixelFormat { // opencl.h:30-56
public:
	static const PixelFormat RGBA8, U8, F32, NV12, RGGB8, GRBG8, BGR8;
	int pixelSize() const { return stride * rowStride; }
	const int stride;
	const int rowStride;
	const bool color;
	const int layout; // VpLayout for raw formats, -1 otherwise
	int imageFormat() const { return this == &U8 ? VP_FMT_U8 : (this == &F32 ? VP_FMT_F32 : (this == &NV12 ? VP_FMT_NV12 : VP_FMT_RGBA8)); }
	const char* kernelOptions;
private:
	PixelFormat(int stride, int rowStride, bool color, int layout, const char* options): stride(stride), rowStride(rowStride), color(color), layout(layout), kernelOptions(options) {}
};

class CLImage;
class RawImage;

class OpenCL { // opencl.h:70-112
public:
	OpenCL();
	~OpenCL();

	cl::Kernel compile(const char* code, const std::string& options = "");

	template<typename... Ts>
	cl::Event run(cl::Kernel kernel, const cl::EnqueueArgs& args, Ts... ts) {
		std::vector<Arg> packed;
		(packed.push_back(Arg(ts)), ...);
		dispatch(kernel, args, packed);
		return cl::Event{ctx, 0};
	}
	template<typename... Ts>
	void await(cl::Kernel kernel, const cl::EnqueueArgs& args, Ts... ts) { wait(run(kernel, args, std::forward<Ts>(ts)...)); }

	static void wait(const cl::Event& event);
	void printRuntimes();
	void clearEvents();

	std::shared_ptr<CLImage> acquire(const PixelFormat* format, int width, int height, const std::string& name);
	std::shared_ptr<RawImage> acquireNV12(int width, int height);

	// extensions the hot path uses
	void setSamplerMode(int mode) { samplerMode = mode; }
	void setSatVariant(int v) { satVariant = v; }
	void setListOrder(int o) { listOrder = o; }
	static VpContext* context() { return ctx; }

private:
	struct Arg {
		enum Kind { BUF, IMG, I32, F32, MODEL } kind;
		void* p = nullptr;
		int w = 0, h = 0;
		int i = 0;
		float f = 0.f;
		CLCameraModel m{};
		Arg(const cl::Buffer& b): kind(BUF), p(b.dptr) {}
		Arg(const cl::Image2D& im): kind(IMG), p(im.dptr), w(im.width), h(im.height) {}
		Arg(int v): kind(I32), i(v) {}
		Arg(float v): kind(F32), f(v) {}
		Arg(const CLCameraModel& v): kind(MODEL), m(v) {}
	};
	void dispatch(const cl::Kernel& kernel, const cl::EnqueueArgs& args, const std::vector<Arg>& a);

	static VpContext* ctx;
	int samplerMode = VP_SAMPLER_BILINEAR_RTZ, satVariant = VP_SAT_TILED, listOrder = VP_ORDER_RASTER;
	std::vector<std::shared_ptr<RawImage>> nv12pool; // (the CLImage pool lives behind vp_image_acquire / vp_image_release)
	// state handed from one launch to the next inside rgba2blobCenter: the reference's satHorizontal/satVertical pair becomes one
	// fused launch, so gradientDot only records its arguments and satHorizontal is a no-op (see dispatch())
	struct { void* flat = nullptr; void* graddot = nullptr; int w = 0, h = 0, offset = 0; } pendingGrad;
	void* satTable = nullptr;
	size_t satBytes = 0;
};

template<typename T>
class CLMap { // opencl.h:114-152: blocking map in the constructor, blocking unmap in the destructor, move-only
public:
	CLMap(VpBuffer* buffer, bool readBack, bool writeBack): buffer(buffer), writeBack(writeBack) {
		void* h = nullptr;
		VP_OR_FATAL(vp_buffer_map(OpenCL::context(), 0, buffer, readBack, &h));
		map = (T*)h;
	}
	~CLMap() {
		if(unmoved) VP_OR_FATAL(vp_buffer_unmap(OpenCL::context(), 0, buffer, writeBack));
	}
	CLMap(CLMap&& other) noexcept: buffer(other.buffer), map(other.map), writeBack(other.writeBack) { other.unmoved = false; }
	CLMap(const CLMap&) = delete;
	CLMap& operator=(const CLMap&) = delete;
	T*& operator*() { return map; }
	T* operator->() { return map; }
	T& operator[](int i) { return map[i]; }
	const T* const& operator*() const { return reinterpret_cast<const T* const&>(map); } // opencl.h:139-141 (`const T*` and `T*` are similar types: the same object, no temporary)
	const T* operator->() const { return map; }
	const T& operator[](int i) const { return map[i]; }

private:
	VpBuffer* buffer;
	T* map;
	bool writeBack;
	bool unmoved = true;
};

class CLArray { // opencl.h:154-165; opencl.cpp:146-147
public:
	explicit CLArray(int size);
	CLArray(void* data, int size);

	template<typename T> CLMap<T> read() const { return CLMap<T>(object.get(), true, false); }
	template<typename T> CLMap<T> write() { return CLMap<T>(object.get(), false, true); }
	template<typename T> CLMap<T> readWrite() { return CLMap<T>(object.get(), true, true); }

	const cl::Buffer buffer;
	const int size;

private:
	std::shared_ptr<VpBuffer> object; // vp_buffer_create / vp_buffer_destroy: released when the last copy of the array goes away
};

class RawImage: public CLArray { // opencl.h:168-188 -- for Bayer formats width/height count 2x2 cells
public:
	RawImage(const RawImage& other) = default; // opencl.h:170: shares the device buffer (here: the shared VpBuffer), as copying a cl::Buffer handle does
	RawImage(const PixelFormat* format, int width, int height): CLArray(width * height * format->pixelSize()), format(format), width(width), height(height), name() {}
	RawImage(CLArray array, const PixelFormat* format, int width, int height, std::string name): CLArray(std::move(array)), format(format), width(width), height(height), name(std::move(name)) {} // opencl.h:172
	RawImage(const PixelFormat* format, int width, int height, std::string name): CLArray(width * height * format->pixelSize()), format(format), width(width), height(height), name(std::move(name)) {}
	RawImage(const PixelFormat* format, int width, int height, double timestamp): CLArray(width * height * format->pixelSize()), format(format), width(width), height(height), timestamp(timestamp), name() {}
	RawImage(const PixelFormat* format, int width, int height, unsigned char* data): CLArray(data, width * height * format->pixelSize()), format(format), width(width), height(height) {}
	RawImage(const PixelFormat* format, int width, int height, double timestamp, unsigned char* data): CLArray(data, width * height * format->pixelSize()), format(format), width(width), height(height), timestamp(timestamp) {}
	virtual ~RawImage() = default;

	const PixelFormat* format;
	const int width;
	const int height;
	double timestamp = 0;
	const std::string name;
};

template<typename T> class CLImageMap;

class CLImage { // opencl.h:195-212
public:
	explicit CLImage(const PixelFormat* format);
	CLImage(const PixelFormat* format, int width, int height, std::string name);

	template<typename T> CLImageMap<T> read() const { return CLImageMap<T>(*this, true, false); }
	template<typename T> CLImageMap<T> write() { return CLImageMap<T>(*this, false, true); }
	template<typename T> CLImageMap<T> readWrite() { return CLImageMap<T>(*this, true, true); }

	// opencl.h:204, opencl.cpp:161-179: "img/" + name + suffix as an 8-bit PNG -- F32: saturate(round(factor * v + offset)) grey, RGBA8: RGB, else grey.
	// Written by vph_write_png (include/vp_host.h), no OpenCV; a path that cannot be opened is a WARN (cv::imwrite returns false there).
	void save(const std::string& suffix, float factor = 1.0f, float offset = 0.0f) const;

	cl::Image2D image;
	const PixelFormat* format;
	int width;
	int height;
	std::string name;

	size_t bytes() const { return (size_t)width * height * format->pixelSize(); }
	VpImage* object() const { return pooled.get(); }

private:
	std::shared_ptr<VpImage> pooled; // vp_image_acquire; handed back to the pool (vp_image_release) when the last copy goes away
};

template<typename T>
class CLImageMap { // opencl.h:215-262 (without the cv::Mat view)
public:
	CLImageMap(const CLImage& image, bool readBack, bool writeBack): object(image.object()), writeBack(writeBack) {
		void* h = nullptr;
		VP_OR_FATAL(vp_image_map(OpenCL::context(), 0, object, readBack, &h));
		map = (T*)h;
		bytePitch = (size_t)image.width * image.format->pixelSize();
		rowPitch = bytePitch / sizeof(T);
	}
	~CLImageMap() {
		if(unmoved) VP_OR_FATAL(vp_image_unmap(OpenCL::context(), 0, object, writeBack));
	}
	CLImageMap(CLImageMap&& other) noexcept: bytePitch(other.bytePitch), rowPitch(other.rowPitch), object(other.object), map(other.map), writeBack(other.writeBack) { other.unmoved = false; }
	CLImageMap(const CLImageMap&) = delete;
	CLImageMap& operator=(const CLImageMap&) = delete;
	T*& operator*() { return map; }
	T* operator->() { return map; }
	T& operator[](int i) { return map[i]; }
	T& operator()(int x, int y) { return map[x + y * rowPitch]; }
	const T& operator()(int x, int y) const { return map[x + y * rowPitch]; }
	const T* const& operator*() const { return reinterpret_cast<const T* const&>(map); } // opencl.h:247-249
	const T* operator->() const { return map; }
	const T& operator[](int i) const { return map[i]; }

	size_t bytePitch;
	size_t rowPitch;

private:
	VpImage* object;
	T* map;
	bool writeBack;
	bool unmoved = true;
};
