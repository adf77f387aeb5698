/*
 * vp_multi.h -- C ABI of the multi-GPU end of the hot path (libvp_multi.so): ONE process, one VpContext per MI355X, camera
 * feeds sharded over the devices, and the only exchange the path has -- the gather of the fixed-size detection records of a
 * step -- as one RCCL all-gather per step on per-device streams, overlapped with the next step's kernels.
 *
 * Replaces: the reference runs one OS process per camera and exchanges SSL_WrapperPackets over UDP multicast
 * (src/udpsocket.cpp:94-100 send, :147-163 receive; README.md:68-71 camera_amount / cam_id tiling).  On one node the same
 * information is the blob records each device's vp_frame produced; every device receives every device's records
 * (ncclCommInitAll + ncclAllGather on uint8, equal counts, latency-bound).
 *
 * What a device contributes per step -- three forms, all fixed-size so that ONE all-gather with equal counts carries them:
 *   full records    vp_record_layout: [counter: batch x 3 int32, padded to 256 B][matches: batch x max_matches x 22 B, padded][idx: batch x
 *                   max_matches x uint32, padded]; vp_frame writes straight into it, nothing is packed (6.7 MB at 128 feeds x 2000 blobs);
 *   compact records vp_record_layout_compact / vp_records_compact (vp_hip.h): per feed counter[3], n and the n records a host would read
 *                   (main.cpp:299-301), `cap` records per slot (0.66 MB at 128 feeds, cap 192);
 *   packets         vp_packet_layout: per feed one slot [int32 length | SSL_WrapperPacket bytes]: what the reference's cameras actually exchange
 *                   (udpsocket.cpp:94-100), produced by the CPU stage (vph_detect_batch) of each device's feeds (0.5 MB at 128 feeds x 4 KB).
 *
 * Errors: VpStatus + vp_last_error() as in vp_hip.h.  No CPU fallback: without RCCL and a HIP device creation fails.
 */
#ifndef VP_MULTI_H
#define VP_MULTI_H

#include "vp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct VpGather VpGather;

/* Byte offsets of the three arrays inside one device's record block; returns the block size (a multiple of 256). */
size_t vp_record_layout(int batch, int max_matches, size_t* off_counter, size_t* off_matches, size_t* off_idx);

/* Packet block: `batch` slots of slot_bytes (a multiple of 4, > 4), slot i = [int32 length | payload of at most slot_bytes - 4 bytes]; length <= 0 = no
 * packet.  Returns the block size (a multiple of 256), 0 for bad arguments. */
size_t vp_packet_layout(int batch, int slot_bytes);

/* One communicator per device (ncclCommInitAll over devices[0..n)), one gather stream per device, n_slots (1..4) send / receive
 * buffer pairs per device so that the gather of step k runs beside the kernels of step k + 1. */
int vp_gather_create(int n_devices, const int* devices, size_t bytes_per_device, int n_slots, VpGather** out);
/* TEST-ONLY, UNSTABLE (not part of the supported ABI: a deployment never calls it, and it may change or go away without notice) constructor: n_ranks ranks that all live on ONE device, each with its own gather stream, send / receive blocks and events;
 * vp_gather_start then performs the all-gather as n x n device-to-device copies on the gather streams behind the same ready / done events.  No RCCL.
 * Never selected implicitly and never a fallback of vp_gather_create (which refuses a device listed twice): it exists so that the N > 1 host pipeline
 * (vp_feeds --gpus N --same-device: driver threads, rendezvous, last-arriver gather, slot reuse, end-of-run drain) can be executed on a one-GPU box. */
int vp_gather_create_loopback(int n_ranks, int device, size_t bytes_per_rank, int n_slots, VpGather** out);
int vp_gather_is_loopback(const VpGather* g);
void vp_gather_destroy(VpGather* g);
int vp_gather_devices(const VpGather* g);

/* Device pointers (memory of devices[rank]): the block this rank contributes, and the n_devices blocks it receives (rank-major). */
void* vp_gather_send(VpGather* g, int rank, int slot);
void* vp_gather_recv(VpGather* g, int rank, int slot);

/* Start the all-gather of `slot`: on every device it begins once the work queued so far on ctx[rank] / stream[rank] has finished
 * (event dependency, no host wait) and runs on that device's gather stream.  One grouped RCCL call over all devices. */
int vp_gather_start(VpGather* g, int slot, VpContext* const* ctx, const int* stream);

/* Block the host until the gather of `slot` has completed on every device (its send block may then be overwritten, its receive
 * blocks read).  A slot that was never started returns at once. */
int vp_gather_wait(VpGather* g, int slot);

#ifdef __cplusplus
}
#endif
#endif /* VP_MULTI_H */
