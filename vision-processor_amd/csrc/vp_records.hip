// Transport form of a step's blob lists (include/vp_hip.h: vp_record_layout_compact, vp_records_compact).
// vp_frame writes fixed-capacity arrays (max_matches = maxBlobs records per frame, main.cpp:257: 2000 x 22 B + indices = 52 KB per frame) of which a frame
// fills a few per cent; what leaves the device -- the read-back of main.cpp:299-300, or the gather between the GPUs of a node -- only needs
// counter[3] and the min(counter[0], maxBlobs) records the host loop reads (main.cpp:301).  One slot per frame:
//     [counter 3 x int32 | n int32 | n_max x VpMatch (22 B) | pad to 4 | n_max x uint32 pixel index], padded to 16 bytes
// with n = min(counter[0], max_matches, cap) records present (n < min(counter[0], max_matches) says the slot was too small for this frame).
#include "vp_common.h"

#include <chrono>

static size_t slot_bytes_of(int cap) {
	const size_t m = (16 + sizeof(VpMatch) * (size_t)cap + 3) & ~size_t(3);
	return (m + 4 * (size_t)cap + 15) & ~size_t(15);
}

extern "C" size_t vp_record_layout_compact(int batch, int cap, size_t* slot_bytes, size_t* off_matches, size_t* off_idx) {
	if(batch <= 0 || cap <= 0) return 0;
	if(slot_bytes) *slot_bytes = slot_bytes_of(cap);
	if(off_matches) *off_matches = 16;
	if(off_idx) *off_idx = (16 + sizeof(VpMatch) * (size_t)cap + 3) & ~size_t(3);
	return (slot_bytes_of(cap) * (size_t)batch + 255) & ~size_t(255);
}

// one workgroup per frame: header, then the records as 16-bit words (VpMatch arrays are 2-byte aligned: 22 B records) and the indices
__global__ __launch_bounds__(256) void k_records_compact(const int32_t* __restrict__ counter, const uint16_t* __restrict__ matches, const uint32_t* __restrict__ idx, int max_matches, int cap,
		uint32_t slot_bytes, uint32_t off_idx, uint8_t* __restrict__ out, uint32_t* __restrict__ arrived, uint32_t* done_word, uint32_t done_value) {
	const int f = blockIdx.x;
	const int c0 = counter[3 * f];
	const int n = min(min(max(c0, 0), max_matches), cap);
	uint8_t* slot = out + (size_t)f * slot_bytes;
	if(threadIdx.x < 4) reinterpret_cast<int32_t*>(slot)[threadIdx.x] = threadIdx.x < 3 ? counter[3 * f + threadIdx.x] : n;
	const uint16_t* src = matches + (size_t)f * max_matches * 11;
	uint16_t* dst = reinterpret_cast<uint16_t*>(slot + 16);
	for(int i = threadIdx.x; i < n * 11; i += 256) dst[i] = src[i];
	if(idx) {
		const uint32_t* isrc = idx + (size_t)f * max_matches;
		uint32_t* idst = reinterpret_cast<uint32_t*>(slot + off_idx);
		for(int i = threadIdx.x; i < n; i += 256) idst[i] = isrc[i];
	}
	if(done_word) {
		// Completion WORD (vp_records_compact_signal): the last workgroup to finish stores done_value into *done_word -- pinned host memory the caller polls, so that
		// "the step's records are readable" reaches the host without a marker packet, an interrupt or any other round trip through the device's command processor.
		// Every workgroup makes its slot visible at system scope before it arrives; the arrival counter cleans itself for the next launch.
		__threadfence_system();
		__syncthreads();
		if(threadIdx.x == 0) {
			const uint32_t before = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
			if(before + 1u == gridDim.x) {
				__hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__threadfence_system();
				__hip_atomic_store(done_word, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
	}
}

static int records_compact(VpContext* ctx, int stream, const int32_t* counter, const VpMatch* matches, const uint32_t* idx, int batch, int max_matches, int cap, uint8_t* out,
		uint32_t* done_word, uint32_t done_value);

extern "C" int vp_records_compact(VpContext* ctx, int stream, const int32_t* counter, const VpMatch* matches, const uint32_t* idx, int batch, int max_matches, int cap, uint8_t* out) {
	return records_compact(ctx, stream, counter, matches, idx, batch, max_matches, cap, out, nullptr, 0);
}

extern "C" int vp_records_compact_signal(VpContext* ctx, int stream, const int32_t* counter, const VpMatch* matches, const uint32_t* idx, int batch, int max_matches, int cap, uint8_t* out,
		uint32_t* done_word, uint32_t done_value) {
	VP_REQUIRE(done_word && (reinterpret_cast<uintptr_t>(done_word) & 3) == 0, "vp_records_compact_signal: done_word must be a 4-byte aligned, device-visible address");
	return records_compact(ctx, stream, counter, matches, idx, batch, max_matches, cap, out, done_word, done_value);
}

// host side of the completion word: spin (pause) until *word == value; no device interaction at all
extern "C" int vp_wait_word(const volatile uint32_t* word, uint32_t value, double timeout_s) {
	VP_REQUIRE(word, "vp_wait_word: null pointer");
	const auto t0 = std::chrono::steady_clock::now();
	for(unsigned spins = 0;; spins++) {
		if(__atomic_load_n(word, __ATOMIC_ACQUIRE) == value) return VP_OK;
		__builtin_ia32_pause();
		if((spins & 1023u) == 1023u && timeout_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
			vp_set_error("vp_wait_word: the word did not take the value %u within %.3f s", value, timeout_s);
			return VP_ERR_TIMEOUT;
		}
	}
}

static int records_compact(VpContext* ctx, int stream, const int32_t* counter, const VpMatch* matches, const uint32_t* idx, int batch, int max_matches, int cap, uint8_t* out,
		uint32_t* done_word, uint32_t done_value) {
	VP_TRY(vp_check_ctx(ctx, stream));
	VP_REQUIRE(counter && matches && out && batch > 0 && batch <= 65535 && max_matches > 0 && cap > 0, "vp_records_compact: bad arguments");
	VP_REQUIRE((reinterpret_cast<uintptr_t>(matches) & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "vp_records_compact: matches must be 2-byte and out 16-byte aligned");
	VP_CHECK_HIP(hipSetDevice(ctx->device));
	VpStageTimer t(ctx, stream, "records compact");
	size_t slot = 0, off_idx = 0;
	vp_record_layout_compact(batch, cap, &slot, nullptr, &off_idx);
	uint32_t* arrived = nullptr;
	if(done_word) { // one self-cleaning arrival counter per stream (zero between launches)
		if(!ctx->compact_arrived[stream]) {
			VP_REQUIRE(!ctx->capturing[stream], "first vp_records_compact_signal on this stream inside a graph capture");
			VP_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->compact_arrived[stream]), 256));
			VP_CHECK_HIP(hipMemsetAsync(ctx->compact_arrived[stream], 0, 256, ctx->streams[stream])); // on the launch's own stream: ordered before the kernel that counts on it
		}
		arrived = ctx->compact_arrived[stream];
	}
	k_records_compact<<<batch, 256, 0, ctx->streams[stream]>>>(counter, reinterpret_cast<const uint16_t*>(matches), idx, max_matches, cap, (uint32_t)slot, (uint32_t)off_idx, out, arrived,
			done_word, done_value);
	return vp_launch_ok("records compact");
}
