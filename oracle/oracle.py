"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes bindings for the two CPU checkers:

* ``Port``  -- oracle/libvporacle.so, this repo's plain-C restatement (vp_oracle.c).
* ``Ref``   -- oracle/_ref/libvpref.so, the reference's own kernel/*.cl compiled unmodified (Makefile `ref`).

Both expose the same numpy-level API so tests can run one against the other.  Only tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module; the product path
(vision-processor_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "libvporacle.so")
REF_SO = os.path.join(HERE, "_ref", "libvpref.so")

RGGB, GRBG, BGR = 0, 1, 2
LAYOUTS = {"RGGB": RGGB, "GRBG": GRBG, "BGR": BGR}
SAMPLER_NEAREST, SAMPLER_BILINEAR_RTZ = 0, 1

# blobList.cl:26-32 / main.cpp:33-41 -- 22 bytes, packed
MATCH_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("color", "u1", (3,)), ("center", "u1", (3,)), ("circ", "<f4"), ("score", "<f4")])
assert MATCH_DTYPE.itemsize == 22


class CameraModel(C.Structure):
    """resampling.cl:20-27 == Perspective.h:22-29 (72 bytes, packed)."""

    _pack_ = 1
    _fields_ = [("shape", C.c_int32 * 2), ("f", C.c_float), ("p", C.c_float * 2), ("d", C.c_float), ("r", C.c_float * 9), ("c", C.c_float * 3)]

    @classmethod
    def make(cls, shape, f, p, d, r, c):
        m = cls()
        m.shape[:] = [int(shape[0]), int(shape[1])]
        m.f = float(f)
        m.p[:] = [float(p[0]), float(p[1])]
        m.d = float(d)
        m.r[:] = [float(v) for v in np.asarray(r, dtype=np.float32).reshape(9)]
        m.c[:] = [float(v) for v in c]
        return m

    def to_array(self) -> np.ndarray:
        return np.frombuffer(bytes(self), dtype=np.uint8).copy()

    @classmethod
    def from_array(cls, a) -> "CameraModel":
        return cls.from_buffer_copy(np.asarray(a, dtype=np.uint8).tobytes())


assert C.sizeof(CameraModel) == 72


def build(ref: bool = True) -> None:
    """Compile the checkers (gcc for the port; ROCm clang for the reference kernels when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", HERE, "oracle"] + (["ref"] if ref else []), check=True)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _u8(a):
    return _p(a, C.c_uint8)


def _f32(a):
    return _p(a, C.c_float)


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class _Base:
    name = "?"

    def sat(self, graddot):
        """sat_horizontal then sat_vertical (Resources.cpp:161-162)."""
        return self.sat_vertical(self.sat_horizontal(graddot))

    def chain_from_flat(self, flat, o, r, R, circ_threshold, min_score=0.0, max_matches=2000):
        g = self.graddot(flat, o)
        s = self.sat(g)
        bc = self.blobcenter(s, r)
        m, idx, cnt = self.bloblist(flat, bc, circ_threshold, min_score, R, max_matches)
        return {"graddot": g, "sat": s, "blobcenter": bc, "matches": m, "idx": idx, "counter": cnt}


class Port(_Base):
    """This repo's C restatement."""

    name = "port"

    def __init__(self, threads: int = 1):
        if not os.path.exists(PORT_SO):
            build(ref=False)
        self.lib = C.CDLL(PORT_SO)
        self.threads = int(threads)
        self.lib.vpo_sat_i32.restype = C.c_int64
        self.lib.vpo_frame.restype = C.c_int

    def raw2quad(self, raw, layout, qw, qh):
        raw = _c(raw, np.uint8).reshape(-1)
        planes = np.zeros((4, qh, qw), np.uint8)
        self.lib.vpo_raw2quad(_u8(raw), layout, qw, qh, _u8(planes[0]), _u8(planes[1]), _u8(planes[2]), _u8(planes[3]))
        return planes

    def resample(self, planes, layout, model, max_robot_height, field_scale, off_x, off_y, sampler_mode, fw, fh):
        planes = _c(planes, np.uint8)
        qh, qw = planes.shape[1:]
        flat = np.zeros((fh, fw, 4), np.uint8)
        self.lib.vpo_resample(_u8(planes[0]), _u8(planes[1]), _u8(planes[2]), _u8(planes[3]), qw, qh, layout, C.byref(model),
                              C.c_float(max_robot_height), C.c_float(field_scale), C.c_float(off_x), C.c_float(off_y), sampler_mode, _u8(flat), fw, fh, self.threads)
        return flat

    def graddot(self, flat, o):
        flat = _c(flat, np.uint8)
        h, w = flat.shape[:2]
        out = np.zeros((h, w), np.float32)
        self.lib.vpo_graddot(_u8(flat), w, h, int(o), _f32(out), self.threads)
        return out

    def sat_horizontal(self, g):
        g = _c(g, np.float32)
        h, w = g.shape
        out = np.zeros_like(g)
        self.lib.vpo_sat_horizontal(_f32(g), w, h, _f32(out), self.threads)
        return out

    def sat_vertical(self, g):
        g = _c(g, np.float32)
        h, w = g.shape
        out = np.zeros_like(g)
        self.lib.vpo_sat_vertical(_f32(g), w, h, _f32(out), self.threads)
        return out

    def sat_i32(self, g):
        """Exact int32 (wrapping) SAT; returns (sat, max|true SAT|)."""
        g = _c(g, np.float32)
        h, w = g.shape
        out = np.zeros((h, w), np.int32)
        mx = self.lib.vpo_sat_i32(_f32(g), w, h, _p(out, C.c_int32))
        return out, int(mx)

    def blobcenter(self, sat, r):
        sat = _c(sat, np.float32)
        h, w = sat.shape
        out = np.zeros_like(sat)
        self.lib.vpo_blobcenter(_f32(sat), w, h, int(r), _f32(out), self.threads)
        return out

    def blobcenter_i32(self, sat, r):
        sat = _c(sat, np.int32)
        h, w = sat.shape
        out = np.zeros((h, w), np.float32)
        self.lib.vpo_blobcenter_i32(_p(sat, C.c_int32), w, h, int(r), _f32(out), self.threads)
        return out

    def bloblist(self, flat, circ, circ_threshold, min_score, R, max_matches):
        flat = _c(flat, np.uint8)
        circ = _c(circ, np.float32)
        h, w = circ.shape
        m = np.zeros(max(max_matches, 1), MATCH_DTYPE)
        idx = np.zeros(max(max_matches, 1), np.uint32)
        cnt = np.zeros(3, np.int32)
        self.lib.vpo_bloblist(_u8(flat), _f32(circ), w, h, C.c_float(circ_threshold), C.c_float(min_score), int(R), int(max_matches),
                              m.ctypes.data_as(C.c_void_p), _p(idx, C.c_uint32), _p(cnt, C.c_int32))
        n = min(int(cnt[0]), max_matches)
        return m[:n].copy(), idx[:n].copy(), cnt

    def blobscore(self, flat, circ, circ_threshold, R):
        flat = _c(flat, np.uint8)
        circ = _c(circ, np.float32)
        h, w = circ.shape
        out = np.zeros_like(circ)
        self.lib.vpo_blobscore(_u8(flat), _f32(circ), w, h, C.c_float(circ_threshold), int(R), _f32(out), self.threads)
        return out

    def circularize(self, g, min_r, max_r):
        g = _c(g, np.float32)
        h, w = g.shape
        out = np.zeros_like(g)
        self.lib.vpo_circularize(_f32(g), w, h, int(min_r), int(max_r), _f32(out), self.threads)
        return out

    def frame(self, raw, layout, qw, qh, model, max_robot_height, field_scale, off_x, off_y, sampler_mode, fw, fh, o, r, R,
              circ_threshold, min_score=0.0, max_matches=2000, sat_mode=0, scratch=None):
        """Whole chain (vpo_frame).  Returns (matches, idx, counter).  `scratch` lets a timing loop reuse buffers."""
        raw = _c(raw, np.uint8).reshape(-1)
        if scratch is None:
            scratch = self.frame_scratch(qw, qh, fw, fh, max_matches)
        s = scratch
        self.lib.vpo_frame(_u8(raw), layout, qw, qh, C.byref(model), C.c_float(max_robot_height), C.c_float(field_scale), C.c_float(off_x), C.c_float(off_y),
                           sampler_mode, fw, fh, int(o), int(r), int(R), C.c_float(circ_threshold), C.c_float(min_score), int(max_matches), int(sat_mode), self.threads,
                           _u8(s["planes"]), _u8(s["flat"]), _f32(s["graddot"]), _f32(s["tmp"]), _f32(s["sat"]), _f32(s["blobcenter"]),
                           s["matches"].ctypes.data_as(C.c_void_p), _p(s["idx"], C.c_uint32), _p(s["counter"], C.c_int32))
        n = min(int(s["counter"][0]), max_matches)
        return s["matches"][:n].copy(), s["idx"][:n].copy(), s["counter"].copy()

    @staticmethod
    def frame_scratch(qw, qh, fw, fh, max_matches):
        return {
            "planes": np.zeros((4, qh, qw), np.uint8), "flat": np.zeros((fh, fw, 4), np.uint8), "graddot": np.zeros((fh, fw), np.float32),
            "tmp": np.zeros((fh, fw), np.float32), "sat": np.zeros((fh, fw), np.float32), "blobcenter": np.zeros((fh, fw), np.float32),
            "matches": np.zeros(max(max_matches, 1), MATCH_DTYPE), "idx": np.zeros(max(max_matches, 1), np.uint32), "counter": np.zeros(3, np.int32),
        }

    # ---- debug / stream colour kernels
    @staticmethod
    def nv12_bytes(w, h):
        return w * h + w * ((h + 1) // 2)

    def quad2rgba(self, planes, layout, sampler_mode):
        planes = _c(planes, np.uint8)
        qh, qw = planes.shape[1:]
        out = np.zeros((qh, qw, 4), np.uint8)
        self.lib.vpo_quad2rgba(_u8(planes[0]), _u8(planes[1]), _u8(planes[2]), _u8(planes[3]), qw, qh, layout, int(sampler_mode), _u8(out))
        return out

    def quad2nv12(self, planes, layout, sampler_mode):
        planes = _c(planes, np.uint8)
        qh, qw = planes.shape[1:]
        out = np.zeros(self.nv12_bytes(qw, qh), np.uint8)
        self.lib.vpo_quad2nv12(_u8(planes[0]), _u8(planes[1]), _u8(planes[2]), _u8(planes[3]), qw, qh, layout, int(sampler_mode), _u8(out))
        return out

    def rgba2nv12(self, rgba):
        rgba = _c(rgba, np.uint8)
        h, w = rgba.shape[:2]
        out = np.zeros(self.nv12_bytes(w, h), np.uint8)
        self.lib.vpo_rgba2nv12(_u8(rgba), w, h, _u8(out))
        return out

    def f2nv12(self, img):
        img = _c(img, np.float32)
        h, w = img.shape
        out = np.zeros(self.nv12_bytes(w, h), np.uint8)
        self.lib.vpo_f2nv12(_f32(img), w, h, _u8(out))
        return out


class Ref(_Base):
    """The reference's own kernels (compiled from /root/reference/kernel/*.cl) behind the ref_shim.c NDRange loops."""

    name = "reference"

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (run `make -C oracle ref` where /root/reference exists)")
        self.lib = C.CDLL(REF_SO)

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def raw2quad(self, raw, layout, qw, qh):
        raw = _c(raw, np.uint8).reshape(-1)
        planes = np.zeros((4, qh, qw), np.uint8)
        self.lib.ref_raw2quad(_u8(raw), layout, qw, qh, _u8(planes[0]), _u8(planes[1]), _u8(planes[2]), _u8(planes[3]))
        return planes

    def resample(self, planes, layout, model, max_robot_height, field_scale, off_x, off_y, sampler_mode, fw, fh):
        planes = _c(planes, np.uint8)
        qh, qw = planes.shape[1:]
        flat = np.zeros((fh, fw, 4), np.uint8)
        self.lib.ref_set_sampler_mode(int(sampler_mode))
        self.lib.ref_resampling(_u8(planes[0]), _u8(planes[1]), _u8(planes[2]), _u8(planes[3]), qw, qh, layout, C.byref(model),
                                C.c_float(max_robot_height), C.c_float(field_scale), C.c_float(off_x), C.c_float(off_y), _u8(flat), fw, fh)
        return flat

    def graddot(self, flat, o):
        flat = _c(flat, np.uint8)
        h, w = flat.shape[:2]
        out = np.zeros((h, w), np.float32)
        self.lib.ref_gradient_dotproduct(_u8(flat), w, h, int(o), _f32(out))
        return out

    def sat_horizontal(self, g):
        g = _c(g, np.float32)
        h, w = g.shape
        out = np.zeros_like(g)
        self.lib.ref_sat_horizontal(_f32(g), w, h, _f32(out))
        return out

    def sat_vertical(self, g):
        g = _c(g, np.float32)
        h, w = g.shape
        out = np.zeros_like(g)
        self.lib.ref_sat_vertical(_f32(g), w, h, _f32(out))
        return out

    def blobcenter(self, sat, r):
        sat = _c(sat, np.float32)
        h, w = sat.shape
        out = np.zeros_like(sat)
        self.lib.ref_circle(_f32(sat), w, h, int(r), _f32(out))
        return out

    def bloblist(self, flat, circ, circ_threshold, min_score, R, max_matches):
        flat = _c(flat, np.uint8)
        circ = _c(circ, np.float32)
        h, w = circ.shape
        m = np.zeros(max(max_matches, 1), MATCH_DTYPE)
        idx = np.zeros(max(max_matches, 1), np.uint32)
        cnt = np.zeros(3, np.int32)
        self.lib.ref_matches(_u8(flat), _f32(circ), w, h, m.ctypes.data_as(C.c_void_p), _p(cnt, C.c_int32), C.c_float(circ_threshold), C.c_float(min_score),
                             int(R), int(max_matches), _p(idx, C.c_uint32))
        n = min(int(cnt[0]), max_matches)
        return m[:n].copy(), idx[:n].copy(), cnt

    def frame_mt(self, raw, layout, qw, qh, model, max_robot_height, field_scale, off_x, off_y, sampler_mode, fw, fh, o, r, R, circ_threshold, min_score, max_matches, threads,
                 scratch=None):
        """The whole chain through the reference's compiled kernels with every NDRange spread over `threads` host threads (timing
        baseline; blob order is the threads').  Returns (matches, counter)."""
        raw = _c(raw, np.uint8).reshape(-1)
        if scratch is None:
            scratch = (np.zeros(4 * qw * qh, np.uint8), np.zeros(4 * fw * fh, np.uint8), np.zeros(fw * fh, np.float32), np.zeros(fw * fh, np.float32), np.zeros(fw * fh, np.float32))
        planes, flat, fa, fb, fc = scratch
        m = np.zeros(max(max_matches, 1), MATCH_DTYPE)
        cnt = np.zeros(3, np.int32)
        self.lib.ref_set_sampler_mode(int(sampler_mode))
        rc = self.lib.ref_frame_mt(_u8(raw), int(layout), int(qw), int(qh), C.byref(model), C.c_float(max_robot_height), C.c_float(field_scale), C.c_float(off_x), C.c_float(off_y),
                                   int(fw), int(fh), int(o), int(r), int(R), C.c_float(circ_threshold), C.c_float(min_score), int(max_matches), int(threads),
                                   _u8(planes), _u8(flat), _f32(fa), _f32(fb), _f32(fc), m.ctypes.data_as(C.c_void_p), _p(cnt, C.c_int32))
        assert rc == 0
        return m[: min(int(cnt[0]), max_matches)].copy(), cnt

    def blobscore(self, flat, circ, circ_threshold, R):
        flat = _c(flat, np.uint8)
        circ = _c(circ, np.float32)
        h, w = circ.shape
        out = np.zeros_like(circ)
        self.lib.ref_blobscore(_u8(flat), _f32(circ), w, h, _f32(out), C.c_float(circ_threshold), int(R))
        return out

    def circularize(self, g, min_r, max_r):
        g = _c(g, np.float32)
        h, w = g.shape
        out = np.zeros_like(g)
        self.lib.ref_circularize(_f32(g), w, h, int(min_r), int(max_r), _f32(out))
        return out

    # ---- debug / stream colour kernels
    @staticmethod
    def nv12_bytes(w, h):
        return w * h + w * ((h + 1) // 2)

    def quad2rgba(self, planes, layout, sampler_mode):
        planes = _c(planes, np.uint8)
        qh, qw = planes.shape[1:]
        out = np.zeros((qh, qw, 4), np.uint8)
        self.lib.ref_set_sampler_mode(int(sampler_mode))
        self.lib.ref_quad2rgba(_u8(planes[0]), _u8(planes[1]), _u8(planes[2]), _u8(planes[3]), qw, qh, layout, _u8(out))
        return out

    def quad2nv12(self, planes, layout, sampler_mode):
        planes = _c(planes, np.uint8)
        qh, qw = planes.shape[1:]
        out = np.zeros(self.nv12_bytes(qw, qh), np.uint8)
        self.lib.ref_set_sampler_mode(int(sampler_mode))
        self.lib.ref_quad2nv12(_u8(planes[0]), _u8(planes[1]), _u8(planes[2]), _u8(planes[3]), qw, qh, layout, _u8(out))
        return out

    def rgba2nv12(self, rgba):
        rgba = _c(rgba, np.uint8)
        h, w = rgba.shape[:2]
        out = np.zeros(self.nv12_bytes(w, h), np.uint8)
        self.lib.ref_rgba2nv12(_u8(rgba), w, h, _u8(out))
        return out

    def f2nv12(self, img):
        img = _c(img, np.float32)
        h, w = img.shape
        out = np.zeros(self.nv12_bytes(w, h), np.uint8)
        self.lib.ref_f2nv12(_f32(img), w, h, _u8(out))
        return out


def matches_equal(a, b, rtol=1e-4, atol=1e-4):
    """Parity rule for blob records: integer fields bit-exact; floats within 1e-4
    absolute-or-relative with NaN==NaN and inf==inf."""
    if len(a) != len(b):
        return False
    for k in ("color", "center"):
        if not np.array_equal(a[k], b[k]):
            return False
    for k in ("x", "y", "circ", "score"):
        if not np.allclose(a[k], b[k], rtol=rtol, atol=atol, equal_nan=True):
            return False
    return True
