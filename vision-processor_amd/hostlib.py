"""ctypes binding of libvp_host.so: the C++ host side that needs no GPU (synthetic frames, geometry producer)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_SO = os.path.join(HERE, "libvp_host.so")


class VpCameraModel(C.Structure):
    """include/vp_hip.h VpCameraModel == Perspective.h:22-29 (72 bytes, packed)."""

    _pack_ = 1
    _fields_ = [("shape", C.c_int32 * 2), ("f", C.c_float), ("p", C.c_float * 2), ("d", C.c_float), ("r", C.c_float * 9), ("c", C.c_float * 3)]


assert C.sizeof(VpCameraModel) == 72

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_SO):
            raise FileNotFoundError(f"{HOST_SO} missing: run `make host` (or __graft_entry__.build())")
        _lib = C.CDLL(HOST_SO)
        _lib.vp_synth_frame.restype = C.c_int
        _lib.vp_synth_camera.restype = None
        _lib.vp_geometry_check.restype = C.c_int
        _lib.vp_synth_truth.restype = C.c_int
    return _lib


def synth_camera(qw, qh, cam_height, plane_height, visible_len, distortion) -> VpCameraModel:
    m = VpCameraModel()
    lib().vp_synth_camera(int(qw), int(qh), C.c_float(cam_height), C.c_float(plane_height), C.c_float(visible_len), C.c_float(distortion), C.byref(m))
    return m


def synth_frame(model, layout, qw, qh, plane_height, field_len, field_wid, seed, n_robots, n_balls, noise_sigma) -> np.ndarray:
    """Returns the raw sensor buffer: Bayer (2qh, 2qw) uint8 or BGR (qh, qw, 3) uint8."""
    raw = np.zeros((qh, qw, 3), np.uint8) if layout == 2 else np.zeros((2 * qh, 2 * qw), np.uint8)
    rc = lib().vp_synth_frame(C.byref(model), int(layout), int(qw), int(qh), C.c_float(plane_height), C.c_float(field_len), C.c_float(field_wid),
                              C.c_uint32(seed & 0xFFFFFFFF), int(n_robots), int(n_balls), C.c_float(noise_sigma), raw.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc != 0:
        raise RuntimeError(f"vp_synth_frame failed: {rc}")
    return raw


def synth_truth(seed, n_robots, n_balls, field_len, field_wid) -> np.ndarray:
    """Ground-truth discs of synth_frame(seed, ...): (n, 6) float32 rows {x, y, radius, r, g, b} in field mm."""
    n = 5 * int(n_robots) + int(n_balls)
    out = np.zeros((max(n, 1), 6), np.float32)
    got = lib().vp_synth_truth(C.c_uint32(seed & 0xFFFFFFFF), int(n_robots), int(n_balls), C.c_float(field_len), C.c_float(field_wid), out.ctypes.data_as(C.c_void_p), n)
    assert got == n, (got, n)
    return out[:n]


def geometry_check(calib, img_w, img_h, calib_w, calib_h, field, ball_radius=21.5, geometry_tolerance=10.0, max_bot_height=180.0, resampling_factor=1.0):
    """Perspective::geometryCheck (Perspective.cpp:35-125) through the Eigen-free restatement.

    calib = (focal, ppx, ppy, distortion, q0, q1, q2, q3, tx, ty, tz); field = (length, width, boundary, boundary_goal_line or -1)."""
    c = (C.c_float * 11)(*[float(v) for v in calib])
    f = (C.c_int * 4)(*[int(v) for v in field])
    of = (C.c_float * 7)()
    oi = (C.c_int * 6)()
    m = VpCameraModel()
    lib().vp_geometry_check(c, int(img_w), int(img_h), int(calib_w), int(calib_h), f, C.c_float(ball_radius), C.c_float(geometry_tolerance),
                            C.c_double(max_bot_height), C.c_float(resampling_factor), of, oi, C.byref(m))
    return {
        "field_scale": of[0], "extent": [of[1], of[2], of[3], of[4]], "min_blob_radius": of[5], "max_blob_radius": of[6],
        "fw": oi[0], "fh": oi[1], "grad_offset": oi[2], "circle_radius": oi[3], "list_radius": oi[4], "geometry_version": oi[5], "model": m,
    }


# ---- blob list -> detection frame (include/vp_host.h) --------------------------------------------
class VphConfig(C.Structure):
    _fields_ = [("cam_id", C.c_int32), ("min_tracking_radius", C.c_double), ("max_bot_acceleration", C.c_double), ("min_score", C.c_double),
                ("min_cam_edge_distance", C.c_double), ("min_confidence", C.c_float), ("clipping_tolerance", C.c_float), ("geometry_tolerance", C.c_float),
                ("reference_force", C.c_float), ("history_force", C.c_float), ("reference_colors", (C.c_int32 * 3) * 6), ("max_bot_height", C.c_double),
                ("default_bot_height", C.c_double), ("yellow_bot_height", C.c_double), ("blue_bot_height", C.c_double)]


class VphFrameHeader(C.Structure):
    _fields_ = [("frame_number", C.c_uint32), ("camera_id", C.c_uint32), ("t_capture", C.c_double), ("t_sent", C.c_double), ("t_capture_camera", C.c_double),
                ("has_t_capture_camera", C.c_int32), ("n_balls", C.c_int32), ("n_robots_yellow", C.c_int32), ("n_robots_blue", C.c_int32), ("n_offsets", C.c_int32)]


VPH_MATCH = np.dtype([("x", "<f4"), ("y", "<f4"), ("color", "<i4", 3), ("center", "<i4", 3), ("circ", "<f4"), ("score", "<f4")])
VPH_TRACKED = np.dtype([("id", "<i4"), ("timestamp", "<f8"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"), ("vx", "<f4"), ("vy", "<f4"), ("vz", "<f4"),
                        ("vw", "<f4"), ("confidence", "<f4"), ("age", "<i4")], align=True)
VPH_BALL = np.dtype([("confidence", "<f4"), ("x", "<f4"), ("y", "<f4"), ("pixel_x", "<f4"), ("pixel_y", "<f4")])
VPH_ROBOT = np.dtype([("confidence", "<f4"), ("robot_id", "<u4"), ("x", "<f4"), ("y", "<f4"), ("orientation", "<f4"), ("pixel_x", "<f4"), ("pixel_y", "<f4"), ("height", "<f4")])
assert VPH_MATCH.itemsize == 40 and VPH_TRACKED.itemsize == 56 and VPH_BALL.itemsize == 20 and VPH_ROBOT.itemsize == 32


def _vph():
    l = lib()
    if not getattr(l, "_vph_ready", False):
        l.vph_detector_create.restype = C.c_void_p
        l.vph_detector_destroy.argtypes = [C.c_void_p]
        l.vph_detector_destroy.restype = None
        l.vph_atan2_fast.restype = C.c_float
        l.vph_atan2_fast.argtypes = [C.c_float, C.c_float]
        l._vph_ready = True
    return l


def config_defaults() -> VphConfig:
    cfg = VphConfig()
    _vph().vph_config_defaults(C.byref(cfg))
    return cfg


class Detector:
    """One camera's blob -> SSL_DetectionFrame stage (main.cpp:319-362) behind the flat C binding."""

    def __init__(self, cfg: VphConfig | None = None):
        self.l = _vph()
        self.cfg = cfg or config_defaults()
        self.h = C.c_void_p(self.l.vph_detector_create(C.byref(self.cfg)))
        if not self.h:
            raise RuntimeError("vph_detector_create failed")

    def close(self):
        if self.h:
            self.l.vph_detector_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p)

    def set_geometry(self, calib, calib_size, img_size, field, ball_radius=21.5, max_robot_radius=90.0, lines=(), arcs=(), resampling_factor=1.0):
        """field = (length, width, boundary, boundary_goal_line or -1, line_thickness)."""
        c = (C.c_float * 11)(*[float(v) for v in calib])
        f = (C.c_int32 * 5)(*[int(v) for v in field])
        ln = np.ascontiguousarray(np.asarray(lines, np.float32).reshape(-1, 4))
        ar = np.ascontiguousarray(np.asarray(arcs, np.float32).reshape(-1, 5))
        rc = self.l.vph_detector_set_geometry(self.h, c, int(calib_size[0]), int(calib_size[1]), int(img_size[0]), int(img_size[1]), C.c_float(resampling_factor), f,
                                              C.c_float(ball_radius), C.c_float(max_robot_radius), self._p(ln), len(ln), self._p(ar), len(ar))
        if rc != 0:
            raise RuntimeError(f"vph_detector_set_geometry failed: {rc}")

    def set_geometry_direct(self, calib, img_size, field_scale, off_x, off_y, fw, fh, field=(9000, 6000, 300, -1, 0), ball_radius=21.5, max_robot_radius=90.0, lines=(), arcs=()):
        """Kernel scalars assigned as the GPU path was launched with instead of derived by geometryCheck."""
        c = (C.c_float * 11)(*[float(v) for v in calib])
        f = (C.c_int32 * 5)(*[int(v) for v in field])
        ln = np.ascontiguousarray(np.asarray(lines, np.float32).reshape(-1, 4))
        ar = np.ascontiguousarray(np.asarray(arcs, np.float32).reshape(-1, 5))
        rc = self.l.vph_detector_set_geometry_direct(self.h, c, int(img_size[0]), int(img_size[1]), C.c_float(field_scale), C.c_float(off_x), C.c_float(off_y), int(fw), int(fh), f,
                                                     C.c_float(ball_radius), C.c_float(max_robot_radius), self._p(ln), len(ln), self._p(ar), len(ar))
        if rc != 0:
            raise RuntimeError(f"vph_detector_set_geometry_direct failed: {rc}")

    def matches_from_cl(self, cl_matches: np.ndarray) -> np.ndarray:
        """22-byte packed CLMatch records (capi.MATCH_DTYPE) -> VPH_MATCH in field coordinates (main.cpp:303-311)."""
        cl_matches = np.ascontiguousarray(cl_matches)
        assert cl_matches.dtype.itemsize == 22
        out = np.zeros(len(cl_matches), VPH_MATCH)
        rc = self.l.vph_matches_from_cl(self.h, self._p(cl_matches), len(cl_matches), self._p(out))
        if rc != 0:
            raise RuntimeError(f"vph_matches_from_cl failed: {rc}")
        return out

    def set_tracked(self, cam: int, objs: np.ndarray):
        objs = np.ascontiguousarray(objs, VPH_TRACKED)
        assert self.l.vph_detector_set_tracked(self.h, C.c_uint32(cam), self._p(objs), len(objs)) == 0

    def get_tracked(self, cam: int) -> np.ndarray:
        n = self.l.vph_detector_get_tracked(self.h, C.c_uint32(cam), None, 0)
        out = np.zeros(n, VPH_TRACKED)
        self.l.vph_detector_get_tracked(self.h, C.c_uint32(cam), self._p(out), n)
        return out

    def set_offsets(self, offsets):
        o = np.ascontiguousarray(offsets, np.float32)
        assert self.l.vph_detector_set_offsets(self.h, self._p(o), len(o)) == 0

    def colors(self) -> np.ndarray:
        """(7, 3) int32: orange, field, yellow, blue, green, pink, fieldLineColor."""
        out = np.zeros((7, 3), np.int32)
        assert self.l.vph_detector_colors(self.h, self._p(out), 0) == 0
        return out

    def set_colors(self, colors):
        c = np.ascontiguousarray(colors, np.int32).reshape(7, 3)
        assert self.l.vph_detector_colors(self.h, self._p(c), 1) == 0

    def detect(self, matches: np.ndarray, frame_id: int, start_time: float, img_timestamp: float = 0.0) -> dict:
        matches = np.ascontiguousarray(matches, VPH_MATCH)
        rc = self.l.vph_detect(self.h, self._p(matches), len(matches), C.c_uint32(frame_id), C.c_double(start_time), C.c_double(img_timestamp))
        if rc != 0:
            raise RuntimeError(f"vph_detect failed: {rc}")
        return self.frame()

    def frame(self) -> dict:
        hd = VphFrameHeader()
        assert self.l.vph_frame_header(self.h, C.byref(hd)) == 0
        balls = np.zeros(hd.n_balls, VPH_BALL)
        yellow = np.zeros(hd.n_robots_yellow, VPH_ROBOT)
        blue = np.zeros(hd.n_robots_blue, VPH_ROBOT)
        offs = np.zeros(hd.n_offsets, np.float32)
        self.l.vph_frame_balls(self.h, self._p(balls), len(balls))
        self.l.vph_frame_robots(self.h, 0, self._p(yellow), len(yellow))
        self.l.vph_frame_robots(self.h, 1, self._p(blue), len(blue))
        self.l.vph_frame_offsets(self.h, self._p(offs), len(offs))
        return {"frame_number": hd.frame_number, "camera_id": hd.camera_id, "t_capture": hd.t_capture, "t_sent": hd.t_sent,
                "t_capture_camera": hd.t_capture_camera if hd.has_t_capture_camera else None, "balls": balls, "robots_yellow": yellow, "robots_blue": blue, "t_offsets": offs}

    def set_t_sent(self, t: float):
        assert self.l.vph_frame_set_t_sent(self.h, C.c_double(t)) == 0

    def track(self):
        assert self.l.vph_track(self.h) == 0

    def apply_geometry_packet(self, data: bytes, img_size, resampling_factor=1.0):
        """VisionSocket::parse + Perspective::geometryCheck for an SSL_WrapperPacket carrying geometry.  Returns None when the packet has
        no geometry (or no calibration for this camera), else the kernel scalars as geometry_check() returns them."""
        buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data.ljust(1, b"\0"))
        of, oi, m = (C.c_float * 7)(), (C.c_int32 * 6)(), VpCameraModel()
        rc = self.l.vph_apply_geometry_packet(self.h, buf, len(data), int(img_size[0]), int(img_size[1]), C.c_float(resampling_factor), of, oi, C.byref(m))
        if rc < 0:
            raise ValueError("malformed SSL_WrapperPacket")
        if rc == 0:
            return None
        return {"field_scale": of[0], "extent": [of[1], of[2], of[3], of[4]], "min_blob_radius": of[5], "max_blob_radius": of[6],
                "fw": oi[0], "fh": oi[1], "grad_offset": oi[2], "circle_radius": oi[3], "list_radius": oi[4], "geometry_version": oi[5], "model": m}

    def udp_open(self, ip: str, port: int):
        rc = self.l.vph_udp_open(self.h, ip.encode(), int(port))
        if rc != 0:
            raise OSError(f"vph_udp_open({ip}:{port}) failed: {rc}")

    def udp_send(self):
        assert self.l.vph_udp_send(self.h) == 0

    def udp_sync(self, at_least: int, timeout_ms: int = 2000) -> int:
        """Waits for `at_least` received detection packets; installs the socket's tracked objects; returns the count (-1 = timeout)."""
        return int(self.l.vph_udp_sync(self.h, int(at_least), int(timeout_ms)))

    def udp_apply_geometry(self, img_size, resampling_factor=1.0, timeout_ms=2000):
        """Waits for a geometry newer than the last applied one on the socket and derives the kernel scalars from it (None = nothing new)."""
        of, oi, m = (C.c_float * 7)(), (C.c_int32 * 6)(), VpCameraModel()
        rc = self.l.vph_udp_apply_geometry(self.h, int(img_size[0]), int(img_size[1]), C.c_float(resampling_factor), int(timeout_ms), of, oi, C.byref(m))
        if rc < 0:
            raise OSError(f"vph_udp_apply_geometry failed: {rc}")
        if rc == 0:
            return None
        return {"field_scale": of[0], "extent": [of[1], of[2], of[3], of[4]], "min_blob_radius": of[5], "max_blob_radius": of[6],
                "fw": oi[0], "fh": oi[1], "grad_offset": oi[2], "circle_radius": oi[3], "list_radius": oi[4], "geometry_version": oi[5], "model": m}

    def udp_update_time(self) -> float:
        """VisionSocket::updateTime; the received clock offsets become the t_offsets of the next frames.  Returns the socket's clock."""
        now = C.c_double(0)
        assert self.l.vph_udp_update_time(self.h, C.byref(now)) == 0
        return now.value

    def gc_open(self, ip: str, port: int, bot_heights: dict):
        """GCSocket: bot_heights = {team name: robot height in mm} (the reference's robot-heights.yml)."""
        names = [k.encode() for k in bot_heights]
        arr = (C.c_char_p * len(names))(*names)
        hs = (C.c_double * len(names))(*[float(v) for v in bot_heights.values()])
        rc = self.l.vph_gc_open(self.h, ip.encode(), int(port), arr, hs, len(names))
        if rc != 0:
            raise OSError(f"vph_gc_open({ip}:{port}) failed: {rc}")

    def gc_sync(self, at_least: int, timeout_ms: int = 2000):
        """Returns (packets received or -1 on timeout, {max, default, yellow, blue} heights now in force)."""
        out = (C.c_double * 4)()
        n = int(self.l.vph_gc_sync(self.h, int(at_least), int(timeout_ms), out))
        return n, {"max": out[0], "default": out[1], "yellow": out[2], "blue": out[3]}

    def gc_close(self):
        self.l.vph_gc_close(self.h)

    def udp_close(self):
        self.l.vph_udp_close(self.h)

    def encode(self) -> bytes:
        n = -self.l.vph_encode(self.h, None, 0)
        buf = (C.c_uint8 * max(n, 1))()
        got = self.l.vph_encode(self.h, buf, n)
        assert got == n, (got, n)
        return bytes(buf[:n])

    def decode(self, data: bytes):
        """Returns (has_detection, source); a detection becomes the current frame."""
        buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data.ljust(1, b"\0"))
        src = C.c_int(0)
        rc = self.l.vph_decode(self.h, buf, len(data), C.byref(src))
        if rc < 0:
            raise ValueError("malformed SSL_WrapperPacket")
        return bool(rc), src.value


def atan2_fast(y, x) -> np.float32:
    return np.float32(_vph().vph_atan2_fast(C.c_float(y), C.c_float(x)))


def kmeans(contrast, values, c1, c2):
    v = np.ascontiguousarray(values, np.int32).reshape(-1, 3)
    a, b, c = (np.ascontiguousarray(t, np.int32) for t in (contrast, c1, c2))
    ok = _vph().vph_kmeans(a.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), len(v), b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
    return bool(ok), tuple(int(t) for t in b), tuple(int(t) for t in c)


def kdtree_range(points, point, radius):
    p = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
    out = np.zeros(len(p), np.int32)
    n = _vph().vph_kdtree_range(p.ctypes.data_as(C.c_void_p), len(p), C.c_float(point[0]), C.c_float(point[1]), C.c_float(radius), out.ctypes.data_as(C.c_void_p), len(out))
    return [int(i) for i in out[:n]]


def nadir_calib(model: VpCameraModel):
    """The 11 calibration floats of a nadir camera described by a VpCameraModel (vp_synth_camera): q = (x -1, y 0, z 0, w 0),
    t = R (-pos) = (-cx, cy, cz) for R = diag(1, -1, -1)."""
    return (model.f, model.p[0], model.p[1], model.d, -1.0, 0.0, 0.0, 0.0, -model.c[0], model.c[1], model.c[2])


def detect_batch(detectors, cl_matches: np.ndarray, counters: np.ndarray, max_matches: int, frame_id: int, start_time: float, img_timestamp: float, t_sent: float,
                 threads: int, packets: np.ndarray, lengths: np.ndarray):
    """vph_detect_batch: one step of len(detectors) independent feeds on `threads` host threads (the call releases the GIL).
    cl_matches: (n, 22 * max_matches) uint8 (or a structured 22-byte array), counters (n, 3) int32; packets (n, cap) uint8 and
    lengths (n,) int32 are filled in place."""
    n = len(detectors)
    assert cl_matches.flags.c_contiguous and counters.flags.c_contiguous and packets.flags.c_contiguous and lengths.flags.c_contiguous
    assert cl_matches.nbytes >= n * 22 * max_matches and counters.dtype == np.int32 and counters.size >= 3 * n
    assert packets.dtype == np.uint8 and packets.shape[0] >= n and lengths.dtype == np.int32 and lengths.size >= n
    hs = (C.c_void_p * n)(*[d.h for d in detectors])
    rc = _vph().vph_detect_batch(hs, n, cl_matches.ctypes.data_as(C.c_void_p), counters.ctypes.data_as(C.c_void_p), int(max_matches), C.c_uint32(frame_id),
                                 C.c_double(start_time), C.c_double(img_timestamp), C.c_double(t_sent), int(threads), packets.ctypes.data_as(C.c_void_p),
                                 int(packets.shape[1]), lengths.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(f"vph_detect_batch failed: {rc}")


def detect_batch_slots(detectors, slots: np.ndarray, slot_bytes: int, max_matches: int, frame_id: int, start_time: float, img_timestamp: float, t_sent: float,
                       threads: int, packet_block: np.ndarray, packet_slot_bytes: int):
    """vph_detect_batch_slots: the same step with the transport forms on both sides -- compact record slots in (vp_records_compact), packet slots
    [int32 length | bytes] out.  Returns the library's status (0 = ok; negative when a record slot was too small for its frame)."""
    n = len(detectors)
    assert slots.dtype == np.uint8 and slots.flags.c_contiguous and slots.size >= n * slot_bytes
    assert packet_block.dtype == np.uint8 and packet_block.flags.c_contiguous and packet_block.size >= n * packet_slot_bytes
    hs = (C.c_void_p * n)(*[d.h for d in detectors])
    return int(_vph().vph_detect_batch_slots(hs, n, slots.ctypes.data_as(C.c_void_p), C.c_size_t(slot_bytes), int(max_matches), C.c_uint32(frame_id), C.c_double(start_time),
                                             C.c_double(img_timestamp), C.c_double(t_sent), int(threads), packet_block.ctypes.data_as(C.c_void_p), int(packet_slot_bytes)))
