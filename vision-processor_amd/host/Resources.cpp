// CALLER SIDE of the drop-in, not product logic.  raw2quad(), quad2rgba(), rgba2blobCenter(), streamQuad() and streamImage() below are the
// reference's call sequences (src/Resources.cpp:138-186) PARAPHRASED against the facade of host/opencl.h (the repeated expressions get names), so that
// the facade demo builds and runs on a box without the reference tree (tests/test_gpu_facade.py).  That the reference's REAL text of those functions
// compiles against this facade is checked where the reference exists: tests/test_reference_callsites.py.  Nothing here computes anything; the work
// happens behind OpenCL::run / await in libvp_hip.so.
#include "Resources.h"

#include <vector>

std::shared_ptr<RawImage> SyntheticCamera::readImage() {
	if(frame >= frames) return nullptr;
	const bool bgr = fmt->layout == VP_LAYOUT_BGR;
	std::vector<unsigned char> raw((size_t)qw * qh * fmt->pixelSize());
	if(vp_synth_frame(&model, fmt->layout, qw, qh, planeHeight, fieldLen, fieldWid, seed + (uint32_t)frame, robots, balls, noise, raw.data()) != VP_OK) FATAL("synthetic frame generation failed");
	(void)bgr;
	frame++;
	return std::make_shared<RawImage>(fmt, qw, qh, getTime(), raw.data());
}

Resources::Resources(std::unique_ptr<SyntheticCamera> cam, int camId_, float geometryTolerance_): camera(std::move(cam)) {
	camId = camId_;
	geometryTolerance = geometryTolerance_;
	openCl = std::make_shared<OpenCL>();
	perspective = std::make_shared<vp::Perspective>(camId, geometryTolerance);
	const char* layout = camera->format().kernelOptions; // Resources.cpp:121-126: the format-dependent kernels get -DRGGB/-DGRBG/-DBGR
	raw2quadKernel = openCl->compile(kernel_raw2quad_cl, layout);
	resampling = openCl->compile(kernel_resampling_cl, layout);
	gradientDot = openCl->compile(kernel_gradientDot_cl);
	satHorizontal = openCl->compile(kernel_satHorizontal_cl);
	satVertical = openCl->compile(kernel_satVertical_cl);
	satBlobCenter = openCl->compile(kernel_satBlobCenter_cl);
	quad2rgbaKernel = openCl->compile(kernel_quad2rgba_cl, layout);
	quad2nv12 = openCl->compile(kernel_quad2nv12_cl, layout);
	rgba2nv12 = openCl->compile(kernel_rgba2nv12_cl);
	f2nv12 = openCl->compile(kernel_f2nv12_cl);
}

void Resources::raw2quad(const RawImage& img, std::shared_ptr<CLImage>* channels) {
	for(int c = 0; c < 4; c++) channels[c] = openCl->acquire(&PixelFormat::U8, img.width, img.height, img.name);
	openCl->await(raw2quadKernel, cl::EnqueueArgs(cl::NDRange(img.width, img.height)), img.buffer, channels[0]->image, channels[1]->image, channels[2]->image, channels[3]->image);
}

void Resources::rgba2blobCenter(const std::shared_ptr<CLImage>* channels, std::shared_ptr<CLImage>& flat, std::shared_ptr<CLImage>& gradDot, std::shared_ptr<CLImage>& blobCenter) {
	const vp::Perspective& p = *perspective;
	const int fw = p.reprojectedFieldSize.x, fh = p.reprojectedFieldSize.y;
	const std::string& name = channels[0]->name;
	const cl::NDRange field(fw, fh);
	flat = openCl->acquire(&PixelFormat::RGBA8, fw, fh, name);
	gradDot = openCl->acquire(&PixelFormat::F32, fw, fh, name);
	// the reference's two SAT images: same (format, size) pool traffic, contents unused here (the table is int32 and private)
	std::shared_ptr<CLImage> gradDotHor = openCl->acquire(&PixelFormat::F32, fw, fh, name);
	std::shared_ptr<CLImage> gradDotSat = openCl->acquire(&PixelFormat::F32, fw, fh, name);
	blobCenter = openCl->acquire(&PixelFormat::F32, fw, fh, name);

	cl::Event e1 = openCl->run(resampling, cl::EnqueueArgs(field), channels[0]->image, channels[1]->image, channels[2]->image, channels[3]->image, flat->image, p.getCLCameraModel(),
			(float)gcSocket->maxBotHeight, p.fieldScale, p.visibleFieldExtent[0], p.visibleFieldExtent[2]);
	cl::Event e2 = openCl->run(gradientDot, cl::EnqueueArgs(e1, field), flat->image, gradDot->image, p.gradOffset());
	cl::Event e3 = openCl->run(satHorizontal, cl::EnqueueArgs(e2, cl::NDRange(fh)), gradDot->image, gradDotHor->image);
	cl::Event e4 = openCl->run(satVertical, cl::EnqueueArgs(e3, cl::NDRange(fw)), gradDotHor->image, gradDotSat->image);
	openCl->await(satBlobCenter, cl::EnqueueArgs(e4, field), gradDotSat->image, blobCenter->image, p.circleRadius());
}

std::shared_ptr<CLImage> Resources::quad2rgba(std::shared_ptr<CLImage>* channels) {
	std::shared_ptr<CLImage> rgba = openCl->acquire(&PixelFormat::RGBA8, channels[0]->width, channels[0]->height, channels[0]->name);
	openCl->await(quad2rgbaKernel, cl::EnqueueArgs(cl::NDRange(channels[0]->width, channels[0]->height)), channels[0]->image, channels[1]->image, channels[2]->image, channels[3]->image, rgba->image);
	return rgba;
}

std::shared_ptr<RawImage> Resources::streamQuad(std::shared_ptr<CLImage>* channels) {
	std::shared_ptr<RawImage> nv12 = openCl->acquireNV12(channels[0]->width, channels[0]->height);
	openCl->await(quad2nv12, cl::EnqueueArgs(cl::NDRange(channels[0]->width, channels[0]->height)), channels[0]->image, channels[1]->image, channels[2]->image, channels[3]->image, nv12->buffer);
	return nv12;
}

std::shared_ptr<RawImage> Resources::streamImage(CLImage& img) {
	cl::Kernel kernel;
	if(img.format == &PixelFormat::RGBA8) {
		kernel = rgba2nv12;
	} else if(img.format == &PixelFormat::F32) {
		kernel = f2nv12;
	} else {
		WARN("Unimplemented pixel format submitted for streaming.");
		return nullptr;
	}
	std::shared_ptr<RawImage> nv12 = openCl->acquireNV12(img.width, img.height);
	openCl->await(kernel, cl::EnqueueArgs(cl::NDRange(img.width, img.height)), img.image, nv12->buffer);
	return nv12;
}
