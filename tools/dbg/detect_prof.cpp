// Where does the CPU hypothesis stage spend its time?  (VERDICT r05 next #6.)  One cfg2 blob list (tools/data/cfg2_blobs_seed1.npy: 135 CLMatch records) through
// vph_detect_batch on ONE thread, many times; built with the host sources and -pg for gprof, or plain for wall-clock.  Tools only, no GPU.
//   g++ -O2 -pg -std=c++17 -Iinclude -ffp-contract=off -o /tmp/prof/detect_prof tools/dbg/detect_prof.cpp vision-processor_amd/host/synth.cpp \
//       vision-processor_amd/host/geometry.cpp vision-processor_amd/host/blobs/*.cpp -pthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vp_hip.h"
#include "vp_host.h"

int main(int argc, char** argv) {
	const char* path = argc > 1 ? argv[1] : "tools/data/cfg2_blobs_seed1.npy";
	const int reps = argc > 2 ? atoi(argv[2]) : 2000;
	FILE* f = fopen(path, "rb");
	if(!f) return 2;
	std::vector<unsigned char> file(1 << 20);
	const size_t got = fread(file.data(), 1, file.size(), f);
	fclose(f);
	const size_t hdr = 10 + (size_t)(file[8] | (file[9] << 8)); // .npy v1: magic(6) version(2) header_len(2) header
	const int n = (int)((got - hdr) / 22);
	const int mm = 2000;
	std::vector<unsigned char> recs((size_t)22 * mm, 0);
	memcpy(recs.data(), file.data() + hdr, (size_t)22 * n);
	VphConfig cfg;
	vph_config_defaults(&cfg);
	VphDetector* d = vph_detector_create(&cfg);
	VpCameraModel cam;
	vp_synth_camera(1024, 768, 5000.f, 180.f, 9600.f, 0.08f, &cam); // cfg2
	const float calib[11] = {cam.f, cam.p[0], cam.p[1], cam.d, -1.f, 0.f, 0.f, 0.f, -cam.c[0], cam.c[1], cam.c[2]};
	const int32_t field[5] = {9000, 6000, 300, -1, 10};
	if(vph_detector_set_geometry_direct(d, calib, 1024, 768, 4.6875f, -4800.f, -3600.f, 2048, 1536, field, 21.5f, 90.f, nullptr, 0, nullptr, 0) != 0) return 3;
	int32_t counter[3] = {n, 0, 0};
	std::vector<uint8_t> packet(8192);
	int32_t len = 0;
	const auto t0 = std::chrono::steady_clock::now();
	for(int k = 1; k <= reps; k++) {
		VphDetector* ds[1] = {d};
		if(vph_detect_batch(ds, 1, reinterpret_cast<const VpMatch*>(recs.data()), counter, mm, (uint32_t)k, k / 70.0, k / 70.0, k / 70.0 + 0.002, 1, packet.data(), (int)packet.size(), &len) != 0) return 4;
	}
	const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
	printf("%d records, %d reps: %.1f us per frame, packet %d bytes\n", n, reps, us, len);
	vph_detector_destroy(d);
	return 0;
}
