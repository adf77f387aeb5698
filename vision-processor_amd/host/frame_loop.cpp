// The per-frame detection loop of the reference (src/main.cpp:251-372) written against the HIP facade: readImage ->
// geometryCheck -> raw2quad -> rgba2blobCenter -> zero counters -> blobList -> read back -> flat2field -> hypotheses ->
// SSL_DetectionFrame -> SSL_WrapperPacket bytes -> (own packet looped back into the tracker, as the multicast group does).
// Sockets and RTP streaming themselves are I/O and out of scope: the packet is written to the dump file instead of sent.
//
// usage: vp_frame_loop <layout 0|1|2> <qw> <qh> <camHeight> <visibleLen> <distortion> <fieldLen> <fieldWid> <frames> <seed> <robots> <balls>
//                      <mode: direct|calib> [direct: fieldScale offX offY fw fh] [dump.bin]
// direct: kernel scalars assigned as in SURVEY 8(d) (Perspective's public members); calib: derived by Perspective::geometryCheck.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "Resources.h"

using vp::Match; // src/blobs/match.h: what the CPU stages consume

int main(int argc, char* argv[]) {
	if(argc < 14) {
		fprintf(stderr, "usage: see frame_loop.cpp\n");
		return 2;
	}
	int ai = 1;
	const int layout = atoi(argv[ai++]), qw = atoi(argv[ai++]), qh = atoi(argv[ai++]);
	const float camHeight = (float)atof(argv[ai++]), visibleLen = (float)atof(argv[ai++]), distortion = (float)atof(argv[ai++]);
	const float fieldLen = (float)atof(argv[ai++]), fieldWid = (float)atof(argv[ai++]);
	const int frames = atoi(argv[ai++]);
	const uint32_t seed = (uint32_t)strtoul(argv[ai++], nullptr, 0);
	const int robots = atoi(argv[ai++]), balls = atoi(argv[ai++]);
	const std::string mode = argv[ai++];
	const PixelFormat* fmt = layout == VP_LAYOUT_RGGB ? &PixelFormat::RGGB8 : (layout == VP_LAYOUT_GRBG ? &PixelFormat::GRBG8 : &PixelFormat::BGR8);

	VpCameraModel cam;
	vp_synth_camera(qw, qh, camHeight, 180.f, visibleLen, distortion, &cam);
	Resources r(std::make_unique<SyntheticCamera>(fmt, qw, qh, cam, 180.f, fieldLen, fieldWid, frames, seed, robots, balls, 3.0f), 0, 10.0f);
	cl::Kernel blobList = r.openCl->compile(kernel_blobList_cl); // main.cpp:253

	vp::Perspective& p = *r.perspective;
	if(mode == "direct") {
		if(argc < ai + 5) FATAL("direct mode needs fieldScale offX offY fw fh");
		p.model.size = {qw, qh};
		p.model.focalLength = cam.f;
		p.model.principalPoint = {cam.p[0], cam.p[1]};
		p.model.distortionK2 = cam.d;
		p.model.pos = {cam.c[0], cam.c[1], cam.c[2]};
		p.model.updateDerived(); // nadir orientation is the CameraModel default
		p.fieldScale = (float)atof(argv[ai++]);
		p.visibleFieldExtent[0] = (float)atof(argv[ai++]);
		p.visibleFieldExtent[2] = (float)atof(argv[ai++]);
		p.reprojectedFieldSize = {atoi(argv[ai]), atoi(argv[ai + 1])};
		ai += 2;
		p.geometryVersion = 1;
	} else {
		vp::FieldSize field;
		field.field_length = (int)fieldLen;
		field.field_width = (int)fieldWid;
		field.boundary_width = 300;
		vp::CameraModel cm;
		cm.size = {qw, qh};
		cm.focalLength = cam.f;
		cm.principalPoint = {cam.p[0], cam.p[1]};
		cm.distortionK2 = cam.d;
		cm.pos = {cam.c[0], cam.c[1], cam.c[2]};
		cm.updateDerived();
		const vp::CameraCalibration calib = cm.getProto(0);
		p.setGeometry(field, &calib, 1);
	}
	FILE* dump = argc > ai ? fopen(argv[ai], "wb") : nullptr;
	FILE* debugDump = getenv("VP_DEBUG_DUMP") ? fopen(getenv("VP_DEBUG_DUMP"), "wb") : nullptr; // debugImages (main.cpp:291)
	r.debugImages = getenv("VP_DEBUG_IMAGES") != nullptr; // debug.images of the reference's config.yml (Resources.cpp:201): img/<name><suffix> PNGs of the first frame

	CLArray matchArray((int)sizeof(CLMatch) * r.maxBlobs); // main.cpp:257-258
	CLArray counter((int)sizeof(int) * 3);
	uint32_t frameId = 0;
	while(true) { // main.cpp:262
		frameId++;
		std::shared_ptr<RawImage> img = r.camera->readImage();
		if(img == nullptr) break;
		const double startTime = r.camera->getTime(); // main.cpp:270
		if(mode != "direct") p.geometryCheck(img->width, img->height, r.gcSocket->maxBotHeight, r.resamplingFactor);
		std::shared_ptr<CLImage> channels[4];
		r.raw2quad(*img, channels);
		if(!p.geometryVersion) continue; // main.cpp:277
		std::shared_ptr<CLImage> flat, gradDot, blobCenter;
		r.rgba2blobCenter(channels, flat, gradDot, blobCenter);
		if(debugDump && frameId == 1) { // main.cpp:291,377-403: the debug / stream views of the first frame
			auto put = [&](const void* p, size_t bytes) {
				const uint64_t n = bytes;
				fwrite(&n, sizeof(n), 1, debugDump);
				fwrite(p, 1, bytes, debugDump);
			};
			std::shared_ptr<CLImage> rgba = r.quad2rgba(channels);
			{
				CLImageMap<uint8_t> m = rgba->read<uint8_t>();
				put(*m, rgba->bytes());
			}
			const size_t quadNv12 = vp_nv12_bytes(img->width, img->height), flatNv12 = vp_nv12_bytes(flat->width, flat->height);
			for(const std::shared_ptr<RawImage>& nv : {r.streamQuad(channels), r.streamImage(*flat), r.streamImage(*blobCenter)}) {
				CLMap<uint8_t> m = nv->read<uint8_t>();
				put(*m, nv->width == img->width && nv->height == img->height ? quadNv12 : flatNv12);
			}
			fclose(debugDump);
			debugDump = nullptr;
		}
		if(r.debugImages && frameId == 1) { // main.cpp:291-295 (the reference saves after the blobList launch; the images are complete here already)
			flat->save(".flat." + std::to_string(frameId) + ".png");
			gradDot->save(".gradDot." + std::to_string(frameId) + ".png", 0.25f, 128.f);
			blobCenter->save(".blob." + std::to_string(frameId) + ".png");
		}
		{
			CLMap<int> counterMap = counter.write<int>(); // main.cpp:283-288
			counterMap[0] = counterMap[1] = counterMap[2] = 0;
		}
		r.openCl->await(blobList, cl::EnqueueArgs(cl::NDRange(p.reprojectedFieldSize.x, p.reprojectedFieldSize.y)), flat->image, blobCenter->image, matchArray.buffer, counter.buffer,
				(float)r.minCircularity, 0.0f, p.listRadius(), r.maxBlobs);

		std::vector<Match> matches; // main.cpp:297-317
		std::vector<uint8_t> packet;
		int counts[3];
		int nBalls = 0, nBots = 0;
		{
			CLMap<int> counterMap = counter.read<int>();
			CLMap<CLMatch> matchMap = matchArray.read<CLMatch>();
			const int matchAmount = std::min(r.maxBlobs, counterMap[0]);
			matches.reserve(matchAmount);
			for(int i = 0; i < matchAmount; i++) {
				const CLMatch& m = matchMap[i];
				matches.push_back({p.flat2field({m.x, m.y}), {m.color.r, m.color.g, m.color.b}, {m.center.r, m.center.g, m.center.b}, m.circ, m.score});
			}
			if(counterMap[0] > r.maxBlobs) WARN("max blob amount reached: " << counterMap[0] << "/" << r.maxBlobs);
			for(int i = 0; i < 3; i++) counts[i] = counterMap[i];
			// main.cpp:319-370: hypotheses -> detection frame -> wrapper packet; t_sent is the camera clock (main.cpp:367)
			vp::DetectionFrame detection = vp::detectFrame(r, matches, frameId, startTime, img->timestamp);
			detection.t_sent = r.camera->getTime();
			packet = vp::encodeWrapperPacket(detection); // r.socket->send(wrapper)
			{ // the multicast group delivers the packet to every camera process including this one: VisionSocket::parse (udpsocket.cpp:146-154)
				vp::DetectionFrame received;
				bool hasDetection = false;
				int source = 0;
				if(!vp::decodeWrapperPacket(packet.data(), packet.size(), received, hasDetection, source) || !hasDetection) FATAL("own SSL_WrapperPacket does not parse");
				vp::detectionTracking(r, received);
			}
			nBalls = (int)detection.balls.size();
			nBots = (int)(detection.robots_yellow.size() + detection.robots_blue.size());
			if(dump) {
				const int hdr[8] = {(int)frameId, matchAmount, counts[0], counts[1], counts[2], p.reprojectedFieldSize.x, p.reprojectedFieldSize.y, (int)packet.size()};
				fwrite(hdr, sizeof(hdr), 1, dump);
				const float sc[4] = {p.fieldScale, p.visibleFieldExtent[0], p.visibleFieldExtent[2], 0.f};
				fwrite(sc, sizeof(sc), 1, dump);
				fwrite(*matchMap, sizeof(CLMatch), (size_t)matchAmount, dump);
				for(const Match& m : matches) fwrite(&m.pos, sizeof(float), 2, dump);
				fwrite(packet.data(), 1, packet.size(), dump);
			}
		}
		printf("frame %u: %zu blobs %d balls %d bots, counter {%d, %d, %d}, flat %dx%d, fieldScale %.6f\n", frameId, matches.size(), nBalls, nBots, counts[0], counts[1], counts[2],
				p.reprojectedFieldSize.x, p.reprojectedFieldSize.y, p.fieldScale); // main.cpp:365
		r.openCl->printRuntimes();
		r.openCl->clearEvents(); // main.cpp:372
	}
	if(dump) fclose(dump);
	return 0;
}
