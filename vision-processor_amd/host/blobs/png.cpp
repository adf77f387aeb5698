// CLImage::save without OpenCV (opencl.h:204, opencl.cpp:161-179: cv::imwrite of a debug image): a PNG writer for 8-bit grey / RGB / RGBA pixels whose
// image data are STORED deflate blocks (valid for every decoder, no zlib), and the F32 -> grey conversion of the save's first branch.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "vp_hip.h"
#include "vp_host.h"

namespace {

uint32_t crc32_update(const uint8_t* p, size_t n, uint32_t crc) {
	static const std::vector<uint32_t> table = [] {
		std::vector<uint32_t> t(256);
		for(uint32_t i = 0; i < 256; i++) {
			uint32_t c = i;
			for(int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
			t[i] = c;
		}
		return t;
	}();
	for(size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
	return crc;
}

void be32(std::vector<uint8_t>& v, uint32_t x) {
	for(int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s));
}

bool chunk(FILE* f, const char type[4], const std::vector<uint8_t>& data) {
	std::vector<uint8_t> head, tail;
	be32(head, (uint32_t)data.size());
	head.insert(head.end(), type, type + 4);
	uint32_t crc = crc32_update(reinterpret_cast<const uint8_t*>(type), 4, 0xFFFFFFFFu);
	crc = crc32_update(data.data(), data.size(), crc) ^ 0xFFFFFFFFu;
	be32(tail, crc);
	return fwrite(head.data(), 1, head.size(), f) == head.size() && (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) && fwrite(tail.data(), 1, 4, f) == 4;
}

} // namespace

extern "C" int vph_write_png(const char* path, int w, int h, int channels, const uint8_t* pixels) {
	if(!path || !pixels || w <= 0 || h <= 0 || (channels != 1 && channels != 3 && channels != 4) || (size_t)w * channels > 0x7fffffffu / (size_t)h) return VP_ERR_INVALID;
	FILE* f = fopen(path, "wb");
	if(!f) return VP_ERR_INVALID;
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
	bool ok = fwrite(sig, 1, 8, f) == 8;
	std::vector<uint8_t> ihdr;
	be32(ihdr, (uint32_t)w);
	be32(ihdr, (uint32_t)h);
	const uint8_t rest[5] = {8, (uint8_t)(channels == 1 ? 0 : (channels == 3 ? 2 : 6)), 0, 0, 0}; // bit depth, colour type, deflate, adaptive filtering, no interlace
	ihdr.insert(ihdr.end(), rest, rest + 5);
	ok = ok && chunk(f, "IHDR", ihdr);
	// zlib stream: header, stored blocks of at most 65535 bytes over the filtered scanlines (filter byte 0 + row), Adler-32 of those bytes
	const size_t row = (size_t)w * channels, total = (row + 1) * (size_t)h;
	std::vector<uint8_t> z;
	z.reserve(total + (total / 65535 + 1) * 5 + 8);
	z.push_back(0x78);
	z.push_back(0x01);
	uint32_t a = 1, b = 0;
	size_t produced = 0, in_block = 0;
	auto put = [&](uint8_t v) {
		if(in_block == 0) {
			const size_t len = std::min<size_t>(65535, total - produced);
			z.push_back(produced + len == total ? 1 : 0); // BFINAL, BTYPE = 00 (stored)
			z.push_back((uint8_t)(len & 0xFF));
			z.push_back((uint8_t)(len >> 8));
			z.push_back((uint8_t)(~len & 0xFF));
			z.push_back((uint8_t)((~len >> 8) & 0xFF));
			in_block = len;
		}
		z.push_back(v);
		a = (a + v) % 65521u;
		b = (b + a) % 65521u;
		produced++;
		in_block--;
	};
	for(int y = 0; y < h; y++) {
		put(0);
		const uint8_t* r = pixels + (size_t)y * row;
		for(size_t i = 0; i < row; i++) put(r[i]);
	}
	be32(z, (b << 16) | a);
	ok = ok && chunk(f, "IDAT", z) && chunk(f, "IEND", {});
	ok = (fclose(f) == 0) && ok;
	return ok ? VP_OK : VP_ERR_INVALID;
}

extern "C" void vph_f32_to_grey(const float* in, size_t n, float factor, float offset, uint8_t* out) {
	for(size_t i = 0; i < n; i++) {
		const float v = factor * in[i] + offset;
		// cv::saturate_cast<uchar>(float) = saturate(cvRound(v)); cvRound = lrintf in the default rounding mode (ties to even), and a value no int holds
		// (NaN, +-inf, |v| >= 2^31) comes back from cvtss2si as INT_MIN, which saturates to 0
		const long r = (v != v || v < -2147483648.0f || v >= 2147483648.0f) ? (long)INT_MIN : lrintf(v);
		out[i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
	}
}
