// The launch half of the reference's pipeline facade (src/Resources.{h,cpp}): kernel handles plus the two per-frame launch
// sequences raw2quad() and rgba2blobCenter().  Configuration parsing, sockets, streaming and calibration are control plane
// and out of scope; the fields the frame loop reads keep their names.
#pragma once

#include <memory>
#include <string>

#include "CameraModel.h"
#include "blobs/detection.h"
#include "cl_kernels.h"
#include "opencl.h"

struct __attribute__((packed)) RGB { // Resources.h:30-36
	uint8_t r, g, b;
};

struct __attribute__((packed)) CLMatch { // main.cpp:33-41
	float x, y;
	RGB color;
	RGB center;
	float circ;
	float score;
};
static_assert(sizeof(CLMatch) == 22 && sizeof(CLMatch) == sizeof(VpMatch), "CLMatch layout");

// Synthetic stand-in for src/driver/*: same contract as CameraDriver (cameradriver.h:35-47): readImage() hands out a RawImage
// whose width/height count 2x2 cells for Bayer formats, nullptr when the source is exhausted.
class SyntheticCamera {
public:
	SyntheticCamera(const PixelFormat* format, int qw, int qh, const VpCameraModel& model, float planeHeight, float fieldLen, float fieldWid, int frames, uint32_t seed,
			int robots, int balls, float noise): fmt(format), qw(qw), qh(qh), model(model), planeHeight(planeHeight), fieldLen(fieldLen), fieldWid(fieldWid), frames(frames), seed(seed),
			robots(robots), balls(balls), noise(noise) {}
	std::shared_ptr<RawImage> readImage();
	const PixelFormat& format() const { return *fmt; }
	double getTime() const { return frame / 30.0; }

private:
	const PixelFormat* fmt;
	int qw, qh;
	VpCameraModel model;
	float planeHeight, fieldLen, fieldWid;
	int frames;
	uint32_t seed;
	int robots, balls;
	float noise;
	int frame = 0;
};

// vp::BlobResources carries the members the CPU hypothesis stage reads (camId, minScore, minConfidence, the colour model,
// maxBotHeight, perspective, tracked objects ...: blobs/detection.h); the launch half adds the rest.
class Resources: public vp::BlobResources {
public:
	Resources(std::unique_ptr<SyntheticCamera> camera, int camId, float geometryTolerance);

	std::unique_ptr<SyntheticCamera> camera;
	double minCircularity = 15.0; // thresholds.circularity (Resources.cpp:190)
	// minScore (BlobResources) is applied on the CPU after the list stage (main.cpp:342); the kernel always gets 0.0f
	int maxBlobs = 2000;          // thresholds.blobs (Resources.cpp:84)
	float resamplingFactor = 1.0f;
	bool debugImages = false;

	std::shared_ptr<OpenCL> openCl;

	cl::Kernel raw2quadKernel, resampling, gradientDot, satHorizontal, satVertical, satBlobCenter;
	cl::Kernel quad2rgbaKernel, quad2nv12, rgba2nv12, f2nv12; // Resources.cpp:127-130

	void raw2quad(const RawImage& img, std::shared_ptr<CLImage>* channels); // Resources.cpp:138-143
	void rgba2blobCenter(const std::shared_ptr<CLImage>* channels, std::shared_ptr<CLImage>& flat, std::shared_ptr<CLImage>& gradDot, std::shared_ptr<CLImage>& blobCenter); // :151-164
	std::shared_ptr<CLImage> quad2rgba(std::shared_ptr<CLImage>* channels); // :145-149
	// :166-186 -- the reference hands the NV12 frame to rtpStreamer->sendFrame (RTP/libav: out of scope); here the frame is returned
	std::shared_ptr<RawImage> streamQuad(std::shared_ptr<CLImage>* channels);
	std::shared_ptr<RawImage> streamImage(CLImage& img);
};
