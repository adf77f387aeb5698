/* vp_host.h -- C ABI of the CPU stage that consumes the GPU hot path's blob list (libvp_host.so; no GPU, no HIP).
 *
 * blob list -> robot / ball hypotheses -> SSL_DetectionFrame -> SSL_WrapperPacket bytes.
 * Each entry names the reference code it replaces (paths relative to the reference tree).  The C++ implementation with the
 * reference's own function names lives in vision-processor_amd/host/blobs/detection.h; this header is the flat binding a
 * non-C++ caller (the ctypes tests, a cgo / JNI stub) would use.  Parity of this stage is UNPINNED (see detection.h).
 */
#ifndef VP_HOST_H
#define VP_HOST_H

#include <stdint.h>

#include "vp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct VphDetector VphDetector; /* the stage's slice of `Resources` + the last SSL_DetectionFrame it produced / decoded */

/* src/blobs/match.h:22-30 -- position in field mm (after Perspective::flat2field), colours widened to int */
typedef struct VphMatch {
	float x, y;
	int32_t color[3];
	int32_t center[3];
	float circ, score;
} VphMatch;

/* The tunables of Resources.cpp:189-209 that this stage reads, plus the GCSocket heights (udpsocket.h:122-128). */
typedef struct VphConfig {
	int32_t cam_id;
	double min_tracking_radius, max_bot_acceleration /* mm/s^2 */, min_score, min_cam_edge_distance;
	float min_confidence, clipping_tolerance, geometry_tolerance, reference_force, history_force;
	int32_t reference_colors[6][3]; /* orange, field, yellow, blue, green, pink */
	double max_bot_height, default_bot_height, yellow_bot_height, blue_bot_height;
} VphConfig;

typedef struct VphTracked { /* udpsocket.h:57-64 */
	int32_t id;
	double timestamp;
	float x, y, z, w, vx, vy, vz, vw, confidence;
	int32_t age;
} VphTracked;

typedef struct VphBall { /* SSL_DetectionBall, the fields hypothesis.cpp:69-80 sets */
	float confidence, x, y, pixel_x, pixel_y;
} VphBall;

typedef struct VphRobot { /* SSL_DetectionRobot, hypothesis.cpp:140-153 */
	float confidence;
	uint32_t robot_id;
	float x, y, orientation, pixel_x, pixel_y, height;
} VphRobot;

typedef struct VphFrameHeader { /* SSL_DetectionFrame scalars, main.cpp:345-351,364-368 */
	uint32_t frame_number, camera_id;
	double t_capture, t_sent, t_capture_camera;
	int32_t has_t_capture_camera;
	int32_t n_balls, n_robots_yellow, n_robots_blue, n_offsets;
} VphFrameHeader;

void vph_config_defaults(VphConfig* cfg); /* Resources.cpp:84-85,189-209 defaults, shipped robot-heights.yml */
VphDetector* vph_detector_create(const VphConfig* cfg);
void vph_detector_destroy(VphDetector* d);

/* Perspective::geometryCheck (Perspective.cpp:35-125) for one camera calibration + the field description.
 * calib = {focal, ppx, ppy, distortion, q0, q1, q2, q3, tx, ty, tz}; field = {length, width, boundary, boundary_goal_line or -1,
 * line_thickness}; lines = n_lines x {p1x, p1y, p2x, p2y}; arcs = n_arcs x {cx, cy, radius, a1, a2}. */
int vph_detector_set_geometry(VphDetector* d, const float calib[11], int calib_w, int calib_h, int img_w, int img_h, float resampling_factor, const int32_t field[5],
		float ball_radius, float max_robot_radius, const float* lines, int n_lines, const float* arcs, int n_arcs);

/* The same with the kernel scalars assigned directly instead of derived (fieldScale, extent[0], extent[2] and the
 * flat size are what the GPU path was launched with); skips the O(image) fieldScale reduction of geometryCheck. */
int vph_detector_set_geometry_direct(VphDetector* d, const float calib[11], int img_w, int img_h, float field_scale, float off_x, float off_y, int fw, int fh,
		const int32_t field[5], float ball_radius, float max_robot_radius, const float* lines, int n_lines, const float* arcs, int n_arcs);

/* main.cpp:303-311: CLMatch (22 bytes packed) -> Match with Perspective::flat2field applied */
int vph_matches_from_cl(const VphDetector* d, const VpMatch* in, int n, VphMatch* out);

/* VisionSocket::getTrackedObjects / getReceivedOffsets (udpsocket.h:77,81) as settable state */
int vph_detector_set_tracked(VphDetector* d, uint32_t cam, const VphTracked* objs, int n);
int vph_detector_get_tracked(const VphDetector* d, uint32_t cam, VphTracked* out, int cap); /* returns the count */
int vph_detector_set_offsets(VphDetector* d, const float* offsets, int n);
/* the adaptive colours r.orange, field, yellow, blue, green, pink, fieldLineColor (Resources.h:59-65): set != 0 writes them */
int vph_detector_colors(VphDetector* d, int32_t colors[7][3], int set);

/* main.cpp:319-362: hypotheses, filters, colour update, detection frame.  The frame stays in the detector. */
int vph_detect(VphDetector* d, const VphMatch* matches, int n, uint32_t frame_id, double start_time, double img_timestamp);
int vph_frame_header(const VphDetector* d, VphFrameHeader* out);
int vph_frame_set_t_sent(VphDetector* d, double t_sent); /* main.cpp:364-368 */
int vph_frame_balls(const VphDetector* d, VphBall* out, int cap);
int vph_frame_robots(const VphDetector* d, int blue, VphRobot* out, int cap);
int vph_frame_offsets(const VphDetector* d, float* out, int cap);

/* VisionSocket::detectionTracking (udpsocket.cpp:204-256) applied to the detector's current frame */
int vph_track(VphDetector* d);

/* SSL_WrapperPacket{detection, source = SSL_SOURCE_VISION_PROCESSOR} (socket->send(wrapper), main.cpp:370): returns the length,
 * or -needed when cap is too small. */
int vph_encode(const VphDetector* d, uint8_t* buf, int cap);
/* VisionSocket::parse (udpsocket.cpp:146-154): 1 = packet carried a detection (now the current frame), 0 = none, <0 malformed */
int vph_decode(VphDetector* d, const uint8_t* buf, int len, int* source);

/* VisionSocket::parse + geometryCheck for a geometry packet (udpsocket.cpp:113-130,156-162; Perspective.cpp:35-125): when the packet
 * carries SSL_GeometryData with a calibration for this detector's camera id, the geometry is installed and the kernel scalars are
 * derived for a sensor of img_w x img_h quad pixels.  Returns 1 = applied, 0 = no geometry / no calibration for this camera, < 0 =
 * malformed.  out_f / out_i / out_model as vp_geometry_check (may be NULL). */
int vph_apply_geometry_packet(VphDetector* d, const uint8_t* buf, int len, int img_w, int img_h, float resampling_factor, float* out_f, int32_t* out_i, VpCameraModel* out_model);

/* VisionSocket (udpsocket.h:67-113): attach a datagram socket to the detector.  vph_udp_send emits the current frame's packet to
 * (ip, port) (a multicast group such as 224.5.23.2:10006, or a unicast address); a receiver thread parses every datagram that
 * arrives there -- the detector's own included -- and tracks its detections per camera id.  vph_udp_sync waits until at least
 * `at_least` detection packets have been received since the socket was opened (returns the count, or -1 on timeout) and installs
 * the socket's tracked objects as the detector's (what main.cpp:86 reads through r.socket->getTrackedObjects()). */
int vph_udp_open(VphDetector* d, const char* ip, int port);
int vph_udp_send(VphDetector* d);
int vph_udp_sync(VphDetector* d, int at_least, int timeout_ms);
/* VisionSocket::geometryCheck + Perspective::geometryCheck (udpsocket.cpp:113-130, Perspective.cpp:35-125): waits up to timeout_ms for a
 * geometry newer than the one last applied to arrive on the socket and derives the kernel scalars from it.  1 = applied, 0 = nothing
 * new (or no calibration for this camera), < 0 = error.  Outputs as vph_apply_geometry_packet. */
int vph_udp_apply_geometry(VphDetector* d, int img_w, int img_h, float resampling_factor, int timeout_ms, float* out_f, int32_t* out_i, VpCameraModel* out_model);
/* VisionSocket::updateTime + getReceivedOffsets (udpsocket.cpp:139-144,259-282; main.cpp:361-362,371): steers the socket's clock by
 * half the mean offset asymmetry against the other cameras, installs the received offsets as the t_offsets of the detector's next
 * frames, returns the clock (system time + accumulated offset, seconds) in *now. */
int vph_udp_update_time(VphDetector* d, double* now);
int vph_udp_close(VphDetector* d);

/* GCSocket (udpsocket.cpp:304-330): robot heights per team colour from game-controller Referee packets.  names / heights_mm: the
 * height table (the reference ships robot-heights.yml); the detector's maxBotHeight / defaultBotHeight / yellow / blue heights are
 * set from it at once, and vph_gc_sync (waits for at_least packets, -1 on timeout) installs the heights of the teams the referee
 * named.  NB maxBotHeight also enters the GPU path (resampling plane, Resources.cpp:159): re-derive the geometry after it changes. */
int vph_gc_open(VphDetector* d, const char* ip, int port, const char* const* names, const double* heights_mm, int n);
int vph_gc_sync(VphDetector* d, int at_least, int timeout_ms, double heights_out[4] /* max, default, yellow, blue */);
int vph_gc_close(VphDetector* d);

/* One step of `n` independent camera feeds (one detector each, e.g. the frames of one GPU launch group), on `threads` host
 * threads: per feed the read-back of main.cpp:297-317 (min(max_matches, counter[0]) records of its slice, flat2field), vph_detect,
 * t_sent, the loop-back of its own packet into its tracker, and the encoded SSL_WrapperPacket into packets + i * packet_cap
 * (lengths[i] = bytes written, or -needed when packet_cap is too small: nothing written for that feed).
 * cl_matches = n x max_matches records, counters = n x 3. */
int vph_detect_batch(VphDetector* const* ds, int n, const VpMatch* cl_matches, const int32_t* counters, int max_matches, uint32_t frame_id, double start_time,
		double img_timestamp, double t_sent, int threads, uint8_t* packets, int packet_cap, int32_t* lengths);

/* The same step with the transport forms on both sides: feed i reads slot i of a compact record block (vp_records_compact, vp_hip.h: [counter 3 x int32 |
 * n | records], slot_bytes apart) and writes slot i of a packet block (vp_packet_layout, vp_multi.h: [int32 length | payload], packet_slot_bytes
 * apart; length = -needed when the slot is too small).  A record slot that holds fewer records than min(max_matches, counter[0]) would drop blobs:
 * VP_ERR_INVALID, nothing is run. */
int vph_detect_batch_slots(VphDetector* const* ds, int n, const uint8_t* slots, size_t slot_bytes, int max_matches, uint32_t frame_id, double start_time, double img_timestamp,
		double t_sent, int threads, uint8_t* packet_block, int packet_slot_bytes);

/* CLImage::save (opencl.h:204, opencl.cpp:161-179: cv::imwrite of a debug image as PNG): 8-bit greyscale (channels 1), RGB (3) or RGBA (4) pixels,
 * tightly packed rows, as a PNG whose image data are stored (uncompressed) deflate blocks -- every decoder reads it, no zlib / OpenCV needed.
 * VP_ERR_INVALID: bad arguments or the path cannot be opened. */
int vph_write_png(const char* path, int w, int h, int channels, const uint8_t* pixels);
/* the F32 branch of CLImage::save: out[i] = cv::saturate_cast<uint8_t>(factor * in[i] + offset) -- round to nearest even, saturate, NaN -> 0 */
void vph_f32_to_grey(const float* in, size_t n, float factor, float offset, uint8_t* out);

/* building blocks, exported for the unit tests */
float vph_atan2_fast(float y, float x);                                                              /* hypothesis.cpp:35-58 */
int vph_kmeans(const int32_t contrast[3], const int32_t* values, int n, int32_t c1[3], int32_t c2[3]); /* kmeans.cpp:19-90 */
int vph_kdtree_range(const float* points_xy, int n, float px, float py, float radius, int32_t* out_idx, int cap); /* kdtree.cpp */

#ifdef __cplusplus
}
#endif
#endif /* VP_HOST_H */
