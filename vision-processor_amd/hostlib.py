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
