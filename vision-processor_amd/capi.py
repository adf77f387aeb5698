"""ctypes binding of libvp_hip.so (include/vp_hip.h).  Plumbing for tests and bench.py only.

Device memory can be anything that yields a raw device pointer: `torch` CUDA(ROCm) tensors (`.data_ptr()`) or
`Context.malloc`.  No compute ever happens in Python and there is no fallback: a missing library or device raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .hostlib import VpCameraModel

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.environ.get("VP_HIP_LIB") or os.path.join(HERE, "libvp_hip.so")  # VP_HIP_LIB: A/B builds side by side (tools only)

VP_OK = 0
SAT_TILED, SAT_TWOPASS, SAT_LOCAL = 0, 1, 2
ORDER_ATOMIC, ORDER_RASTER = 0, 1

MATCH_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("color", "u1", (3,)), ("center", "u1", (3,)), ("circ", "<f4"), ("score", "<f4")])
assert MATCH_DTYPE.itemsize == 22


class VpFrameParams(C.Structure):
    _fields_ = [
        ("layout", C.c_int32), ("qw", C.c_int32), ("qh", C.c_int32), ("model", VpCameraModel),
        ("max_robot_height", C.c_float), ("field_scale", C.c_float), ("off_x", C.c_float), ("off_y", C.c_float),
        ("fw", C.c_int32), ("fh", C.c_int32), ("grad_offset", C.c_int32), ("circle_radius", C.c_int32), ("list_radius", C.c_int32),
        ("circ_threshold", C.c_float), ("min_score", C.c_float), ("max_matches", C.c_int32),
        ("sampler_mode", C.c_int32), ("sat_variant", C.c_int32), ("order", C.c_int32), ("use_coord_map", C.c_int32),
    ]


class VpFrameBuffers(C.Structure):
    _fields_ = [("flat", C.c_void_p), ("sat", C.c_void_p), ("matches", C.c_void_p), ("idx", C.c_void_p), ("counter", C.c_void_p),
                ("planes", C.c_void_p), ("graddot", C.c_void_p), ("blobcenter", C.c_void_p)]


class VpPlacement(C.Structure):
    _fields_ = [("write_gbs", C.c_float), ("best_gbs", C.c_float), ("worst_gbs", C.c_float), ("candidates", C.c_int32), ("selected", C.c_int32), ("fast_gbs", C.c_float)]

    def as_dict(self):
        d = {"write_gbs": round(self.write_gbs, 1), "best_gbs": round(self.best_gbs, 1), "worst_gbs": round(self.worst_gbs, 1), "candidates": int(self.candidates),
             "selected": bool(self.selected), "search": {0: "none", 1: "criterion met", 2: "budget exhausted"}.get(int(self.selected), "?"), "fast_gbs": round(self.fast_gbs, 1)}
        if int(self.selected) == 2 and self.write_gbs < self.fast_gbs:  # the search ended on its budget AND this block is a slow one: a run on it is a slow draw, not the kernel
            d["warning"] = "placement search ran out of candidates, bytes or memory and this block writes below the device's fast rate: expect the SAT pass 5-15 % slower on it as a table"
        return d


VIEW_PLANES, VIEW_GRADDOT, VIEW_BLOBCENTER = 1, 2, 4


class VpStageTime(C.Structure):
    _fields_ = [("stage", C.c_char_p), ("ms", C.c_float)]


class VpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vp_hip error {code}: {msg}")
        self.code = code


_lib = None


class _Fenced:
    """A C entry point behind FENCE_TORCH: torch tensors handed to it through _ptr() are ordered before the call on the DEVICE (vp_stream_wait_for on the
    tensor's own device's current torch stream: one event, no host wait), so the calls themselves still go out back to back and asynchronously, as bench.py,
    the tools and real callers issue them.  Attribute access (restype / argtypes) goes to the ctypes function."""
    __slots__ = ("_f", "_c")

    def __init__(self, f, cdll):
        object.__setattr__(self, "_f", f)
        object.__setattr__(self, "_c", cdll)

    def __getattr__(self, k):
        return getattr(self._f, k)

    def __setattr__(self, k, v):
        setattr(self._f, k, v)

    def __call__(self, *a):
        # Only entries that take (ctx, stream, ...) are fenced; which ones those are is decided by NAME (every entry of vp_hip.h except the ones listed in
        # _NO_STREAM_ENTRIES has that shape), not by what the second argument happens to look like.  Tensors seen by _ptr() stay pending until a fence
        # has actually been emitted for them: a call without a stream in between (vp_sync, vp_events_enable, ...) neither drops nor misuses them.
        if _pending and self._f.__name__ not in _NO_STREAM_ENTRIES and len(a) >= 2 and isinstance(a[1], int):
            import torch
            tensors = list(_pending)
            seen = set()
            for t in tensors:
                sp = torch.cuda.current_stream(t.device).cuda_stream
                if (t.device.index, sp) in seen:
                    continue
                seen.add((t.device.index, sp))
                if sp == 0 and t.device.index != _ctx_device.get(_handle_key(a[0]), t.device.index):
                    # vp_stream_wait_for reads a NULL stream as the legacy stream of the CONTEXT's device: wrong for a tensor of another GPU
                    raise VpError(-1, "FENCE_TORCH: tensor on cuda:%d sits on torch's default stream (handle 0) but the context is on another device; use a real torch stream" % t.device.index)
                rc = self._c.vp_stream_wait_for(a[0], a[1], C.c_void_p(sp))
                if rc != VP_OK:
                    raise VpError(rc, self._c.vp_last_error().decode())
            del _pending[:]
        return self._f(*a)


class _Lib:
    """libvp_hip.so; with FENCE_TORCH set its entry points are wrapped by _Fenced."""

    def __init__(self, cdll):
        object.__setattr__(self, "_c", cdll)

    def __getattr__(self, name):
        f = getattr(self._c, name)
        return _Fenced(f, self._c) if FENCE_TORCH and name.startswith("vp_") and name not in ("vp_last_error", "vp_stream_wait_for") else f


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_SO):
            raise FileNotFoundError(f"{HIP_SO} missing: the HIP extension is required (run `make hip` / __graft_entry__.build()); there is no fallback")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7; importing it first makes libvp_hip.so bind to
        # that copy instead of pulling /opt/rocm's next to it (two runtimes in one process lose the device).
        import torch  # noqa: F401
        c = C.CDLL(HIP_SO)
        c.vp_last_error.restype = C.c_char_p
        c.vp_device_name.restype = C.c_char_p
        c.vp_device_name.argtypes = [C.c_void_p]
        c.vp_ctx_destroy.restype = None
        c.vp_ctx_destroy.argtypes = [C.c_void_p]
        c.vp_stream_wait_for.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        c.vp_stream_release_to.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib = _Lib(c)
    return _lib


# The library's own streams are non-blocking: nothing orders them against the stream torch fills a fresh tensor on (torch.zeros / torch.full are kernels
# on torch's current stream).  A caller that hands such a tensor to a call on a library-owned stream must order the two first -- as with any two HIP
# streams (vp_stream_wait_for, or a synchronisation of its own).  The GPU tests set FENCE_TORCH (tests/conftest.py): every torch tensor handed over is
# then ordered before the call by a device-side event on the torch stream of the tensor's device.  bench.py and the tools leave it off (they synchronise
# where they must).
FENCE_TORCH = False
_pending = []  # torch tensors _ptr() has seen and no fence has covered yet (FENCE_TORCH only)
_ctx_device = {}  # context handle -> device ordinal (FENCE_TORCH's cross-device check)
# entries of vp_hip.h whose second argument is NOT a stream index (no fence is emitted for them; pending tensors wait for the next stage call)
_NO_STREAM_ENTRIES = frozenset({"vp_sync", "vp_flush", "vp_wait_word", "vp_ctx_create", "vp_ctx_adopt_stream", "vp_ctx_destroy", "vp_ctx_tune", "vp_events_enable", "vp_event_times", "vp_events_clear", "vp_frame_graph_stats",
                                "vp_malloc_device", "vp_free_device", "vp_malloc_host", "vp_free_host", "vp_malloc_device_placed", "vp_buffer_create", "vp_buffer_destroy", "vp_buffer_size",
                                "vp_image_acquire", "vp_image_release", "vp_image_info", "vp_frame_buffers_create", "vp_frame_buffers_placement", "vp_frame_buffers_destroy",
                                "vp_device_name", "vp_device_count", "vp_abi_version", "vp_abi_sizeof", "vp_stream_wait_for", "vp_last_error", "vp_halo_strip_bytes", "vp_nv12_bytes",
                                "vp_record_layout_compact", "vp_synth_frame", "vp_synth_camera", "vp_synth_truth"})


def _handle_key(h):
    return getattr(h, "value", h)


def _ptr(x):
    """Raw device pointer of a torch tensor / int / None."""
    if x is None:
        return C.c_void_p(None)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        if FENCE_TORCH and getattr(x, "is_cuda", False):
            _pending.append(x)
        return C.c_void_p(x.data_ptr())
    raise TypeError(type(x))


def frame_params(cfg, sat_variant=SAT_TILED, order=ORDER_RASTER, sampler_mode=None, use_coord_map=True) -> VpFrameParams:
    p = VpFrameParams()
    p.layout, p.qw, p.qh = cfg.layout, cfg.qw, cfg.qh
    p.model = cfg.model()
    p.max_robot_height, p.field_scale, p.off_x, p.off_y = cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y
    p.fw, p.fh = cfg.fw, cfg.fh
    p.grad_offset, p.circle_radius, p.list_radius = cfg.grad_offset, cfg.circle_radius, cfg.list_radius
    p.circ_threshold, p.min_score, p.max_matches = cfg.circ_threshold, cfg.min_score, cfg.max_matches
    p.sampler_mode = cfg.sampler_mode if sampler_mode is None else sampler_mode
    p.sat_variant, p.order = sat_variant, order
    p.use_coord_map = int(use_coord_map)
    return p


class Context:
    """VpContext wrapper.  `stream_ptr` adopts an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=0, n_streams=1, stream_ptr=None):
        self.l = lib()
        h = C.c_void_p()
        if stream_ptr is None:
            rc = self.l.vp_ctx_create(int(device), int(n_streams), C.byref(h))
        else:
            rc = self.l.vp_ctx_adopt_stream(int(device), C.c_void_p(stream_ptr), C.byref(h))
        if rc != VP_OK:
            raise VpError(rc, self.l.vp_last_error().decode())
        self.h = h
        self.n_streams = 1 if stream_ptr is not None else n_streams
        _ctx_device[h.value] = int(device)

    def close(self):
        if getattr(self, "h", None):
            self.l.vp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != VP_OK:
            raise VpError(rc, self.l.vp_last_error().decode())

    @property
    def device_name(self):
        return self.l.vp_device_name(self.h).decode()

    def sync(self, stream=-1):
        self._ck(self.l.vp_sync(self.h, int(stream)))

    def tune(self, key: str, value: int):
        """vp_ctx_tune: result-neutral knobs (tile mappings, resident workgroups, list capacity); the library reads no environment variable."""
        self._ck(self.l.vp_ctx_tune(self.h, key.encode(), int(value)))

    def graph_stats(self):
        out = (C.c_int64 * 4)()
        self._ck(self.l.vp_frame_graph_stats(self.h, out))
        return dict(captures=out[0], replays=out[1], refused=out[2], dropped=out[3])

    # ---- memory -----------------------------------------------------------------------------------------------------
    def malloc(self, nbytes) -> int:
        p = C.c_void_p()
        self._ck(self.l.vp_malloc_device(self.h, C.c_size_t(nbytes), C.byref(p)))
        return p.value

    def free(self, dptr):
        self._ck(self.l.vp_free_device(self.h, C.c_void_p(dptr)))

    def h2d(self, dptr, arr, stream=0):
        arr = np.ascontiguousarray(arr)
        self._ck(self.l.vp_copy_h2d(self.h, stream, _ptr(dptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)))
        self.sync(stream)  # pageable source: keep it alive until the copy is done

    def d2h(self, dptr, shape, dtype, stream=0):
        out = np.empty(shape, dtype)
        self._ck(self.l.vp_copy_d2h_sync(self.h, stream, out.ctypes.data_as(C.c_void_p), _ptr(dptr), C.c_size_t(out.nbytes)))
        return out

    def memset(self, dptr, value, nbytes, stream=0):
        self._ck(self.l.vp_memset_device(self.h, stream, _ptr(dptr), int(value), C.c_size_t(nbytes)))

    # ---- events -------------------------------------------------------------------------------------------------------
    def events_enable(self, on=True):
        self._ck(self.l.vp_events_enable(self.h, int(bool(on))))

    def event_times(self, cap=4096):
        buf = (VpStageTime * cap)()
        n = self.l.vp_event_times(self.h, buf, cap)
        if n < 0:
            self._ck(n)
        return [(buf[i].stage.decode(), float(buf[i].ms)) for i in range(n)]

    def events_clear(self):
        self._ck(self.l.vp_events_clear(self.h))

    # ---- stages -------------------------------------------------------------------------------------------------------
    def raw2quad(self, raw, layout, qw, qh, batch, planes, stream=0):
        self._ck(self.l.vp_raw2quad(self.h, stream, _ptr(raw), int(layout), int(qw), int(qh), int(batch), _ptr(planes)))

    def _resample(self, fn, src, layout, qw, qh, model, height, fs, ox, oy, mode, fw, fh, batch, flat, stream):
        self._ck(fn(self.h, stream, _ptr(src), int(layout), int(qw), int(qh), C.byref(model), C.c_float(height), C.c_float(fs), C.c_float(ox), C.c_float(oy),
                    int(mode), int(fw), int(fh), int(batch), _ptr(flat)))

    def resample(self, planes, layout, qw, qh, model, height, fs, ox, oy, mode, fw, fh, batch, flat, stream=0):
        self._resample(self.l.vp_resample, planes, layout, qw, qh, model, height, fs, ox, oy, mode, fw, fh, batch, flat, stream)

    def resample_raw(self, raw, layout, qw, qh, model, height, fs, ox, oy, mode, fw, fh, batch, flat, stream=0):
        self._resample(self.l.vp_resample_raw, raw, layout, qw, qh, model, height, fs, ox, oy, mode, fw, fh, batch, flat, stream)

    # ---- debug / stream colour kernels
    def nv12_bytes(self, w, h):
        self.l.vp_nv12_bytes.restype = C.c_size_t
        return int(self.l.vp_nv12_bytes(int(w), int(h)))

    def quad2rgba(self, planes, layout, qw, qh, mode, batch, rgba, stream=0):
        self._ck(self.l.vp_quad2rgba(self.h, stream, _ptr(planes), int(layout), int(qw), int(qh), int(mode), int(batch), _ptr(rgba)))

    def raw2rgba(self, raw, layout, qw, qh, mode, batch, rgba, stream=0):
        self._ck(self.l.vp_raw2rgba(self.h, stream, _ptr(raw), int(layout), int(qw), int(qh), int(mode), int(batch), _ptr(rgba)))

    def quad2nv12(self, planes, layout, qw, qh, mode, batch, nv12, stream=0):
        self._ck(self.l.vp_quad2nv12(self.h, stream, _ptr(planes), int(layout), int(qw), int(qh), int(mode), int(batch), _ptr(nv12)))

    def raw2nv12(self, raw, layout, qw, qh, mode, batch, nv12, stream=0):
        self._ck(self.l.vp_raw2nv12(self.h, stream, _ptr(raw), int(layout), int(qw), int(qh), int(mode), int(batch), _ptr(nv12)))

    def quad2rgba4(self, p0, p1, p2, p3, layout, qw, qh, mode, rgba, stream=0):
        self._ck(self.l.vp_quad2rgba4(self.h, stream, _ptr(p0), _ptr(p1), _ptr(p2), _ptr(p3), int(layout), int(qw), int(qh), int(mode), _ptr(rgba)))

    def quad2nv124(self, p0, p1, p2, p3, layout, qw, qh, mode, nv12, stream=0):
        self._ck(self.l.vp_quad2nv124(self.h, stream, _ptr(p0), _ptr(p1), _ptr(p2), _ptr(p3), int(layout), int(qw), int(qh), int(mode), _ptr(nv12)))

    def rgba2nv12(self, rgba, w, h, batch, nv12, stream=0):
        self._ck(self.l.vp_rgba2nv12(self.h, stream, _ptr(rgba), int(w), int(h), int(batch), _ptr(nv12)))

    def f2nv12(self, img, w, h, batch, nv12, stream=0):
        self._ck(self.l.vp_f2nv12(self.h, stream, _ptr(img), int(w), int(h), int(batch), _ptr(nv12)))

    def resample_map(self, model, height, fs, ox, oy, fw, fh, map_out, stream=0):
        self._ck(self.l.vp_resample_map(self.h, stream, C.byref(model), C.c_float(height), C.c_float(fs), C.c_float(ox), C.c_float(oy), int(fw), int(fh), _ptr(map_out)))

    def resample_raw_mapped(self, raw, layout, qw, qh, map_in, mode, fw, fh, batch, flat, stream=0):
        self._ck(self.l.vp_resample_raw_mapped(self.h, stream, _ptr(raw), int(layout), int(qw), int(qh), _ptr(map_in), int(mode), int(fw), int(fh), int(batch), _ptr(flat)))

    def graddot(self, flat, w, h, offset, batch, out, stream=0):
        self._ck(self.l.vp_graddot(self.h, stream, _ptr(flat), int(w), int(h), int(offset), int(batch), _ptr(out)))

    def sat(self, graddot, w, h, batch, variant, sat, stream=0):
        self._ck(self.l.vp_sat(self.h, stream, _ptr(graddot), int(w), int(h), int(batch), int(variant), _ptr(sat)))

    def graddot_sat(self, flat, w, h, offset, batch, variant, sat, graddot_out=None, stream=0):
        self._ck(self.l.vp_graddot_sat(self.h, stream, _ptr(flat), int(w), int(h), int(offset), int(batch), int(variant), _ptr(sat), _ptr(graddot_out)))

    def halo_strip_bytes(self, w, h, offset):
        self.l.vp_halo_strip_bytes.restype = C.c_size_t
        return int(self.l.vp_halo_strip_bytes(int(w), int(h), int(offset)))

    def halo_strips(self, flat, w, h, offset, batch, strips, stream=0):
        self._ck(self.l.vp_halo_strips(self.h, stream, _ptr(flat), int(w), int(h), int(offset), int(batch), _ptr(strips)))

    def graddot_sat_strips(self, flat, strips, w, h, offset, batch, variant, sat, graddot_out=None, stream=0):
        self._ck(self.l.vp_graddot_sat_strips(self.h, stream, _ptr(flat), _ptr(strips), int(w), int(h), int(offset), int(batch), int(variant), _ptr(sat), _ptr(graddot_out)))

    def blobcenter(self, sat, w, h, radius, batch, out, stream=0):
        self._ck(self.l.vp_blobcenter(self.h, stream, _ptr(sat), int(w), int(h), int(radius), int(batch), _ptr(out)))

    def bloblist(self, flat, blobcenter, w, h, batch, thr, min_score, radius, max_matches, order, matches, idx, counter, stream=0):
        self._ck(self.l.vp_bloblist(self.h, stream, _ptr(flat), _ptr(blobcenter), int(w), int(h), int(batch), C.c_float(thr), C.c_float(min_score), int(radius),
                                    int(max_matches), int(order), _ptr(matches), _ptr(idx), _ptr(counter)))

    def blobcenter_list(self, flat, sat, w, h, batch, circle_radius, thr, min_score, list_radius, max_matches, order, matches, idx, counter, blobcenter_out=None, stream=0):
        self._ck(self.l.vp_blobcenter_list(self.h, stream, _ptr(flat), _ptr(sat), int(w), int(h), int(batch), int(circle_radius), C.c_float(thr), C.c_float(min_score),
                                           int(list_radius), int(max_matches), int(order), _ptr(matches), _ptr(idx), _ptr(counter), _ptr(blobcenter_out)))

    def flat_bloblist(self, flat, w, h, batch, grad_offset, circle_radius, thr, min_score, list_radius, max_matches, order, matches, idx, counter, stream=0):
        self._ck(self.l.vp_flat_bloblist(self.h, stream, _ptr(flat), int(w), int(h), int(batch), int(grad_offset), int(circle_radius), C.c_float(thr), C.c_float(min_score),
                                         int(list_radius), int(max_matches), int(order), _ptr(matches), _ptr(idx), _ptr(counter)))

    def blobscore(self, flat, blobcenter, w, h, batch, thr, radius, score, stream=0):
        self._ck(self.l.vp_blobscore(self.h, stream, _ptr(flat), _ptr(blobcenter), int(w), int(h), int(batch), C.c_float(thr), int(radius), _ptr(score)))

    def circularize(self, graddot, w, h, batch, min_r, max_r, out, stream=0):
        self._ck(self.l.vp_circularize(self.h, stream, _ptr(graddot), int(w), int(h), int(batch), int(min_r), int(max_r), _ptr(out)))

    def record_layout_compact(self, batch, cap):
        self.l.vp_record_layout_compact.restype = C.c_size_t
        slot, om, oi = C.c_size_t(), C.c_size_t(), C.c_size_t()
        total = int(self.l.vp_record_layout_compact(int(batch), int(cap), C.byref(slot), C.byref(om), C.byref(oi)))
        return total, int(slot.value), int(om.value), int(oi.value)

    def records_compact(self, counter, matches, idx, batch, max_matches, cap, out, stream=0):
        """vp_records_compact: counter[3], n and the n records of every frame into one slot per frame (what a host loop reads back / a gather carries)."""
        self._ck(self.l.vp_records_compact(self.h, stream, _ptr(counter), _ptr(matches), _ptr(idx), int(batch), int(max_matches), int(cap), _ptr(out)))

    def records_compact_signal(self, counter, matches, idx, batch, max_matches, cap, out_host_ptr, done_word_ptr, done_value, stream=0):
        """vp_records_compact_signal: the slots go straight into pinned host memory (malloc_host) and the last workgroup stores done_value into the completion word."""
        self._ck(self.l.vp_records_compact_signal(self.h, stream, _ptr(counter), _ptr(matches), _ptr(idx), int(batch), int(max_matches), int(cap), C.c_void_p(out_host_ptr),
                                                  C.c_void_p(done_word_ptr), C.c_uint32(done_value)))

    def wait_word(self, word_ptr, value, timeout_s=10.0):
        self._ck(self.l.vp_wait_word(C.c_void_p(word_ptr), C.c_uint32(value), C.c_double(timeout_s)))

    def flush(self, stream=-1):
        self._ck(self.l.vp_flush(self.h, int(stream)))

    def malloc_host(self, nbytes) -> int:
        p = C.c_void_p()
        self._ck(self.l.vp_malloc_host(self.h, C.c_size_t(nbytes), C.byref(p)))
        return p.value

    def free_host(self, hptr):
        self._ck(self.l.vp_free_host(self.h, C.c_void_p(hptr)))

    def frame_slab(self, params: VpFrameParams, batch, views=0) -> "FrameSlab":
        return FrameSlab(self, params, batch, views)

    def frame_buffers(self, params: VpFrameParams, raw, batch, buffers: VpFrameBuffers, stream=0):
        """vp_frame on a VpFrameBuffers (e.g. FrameSlab.b)."""
        self._ck(self.l.vp_frame(self.h, stream, C.byref(params), _ptr(raw), int(batch), C.byref(buffers)))

    def frame(self, params: VpFrameParams, raw, batch, flat, sat, matches, counter, idx=None, planes=None, graddot=None, blobcenter=None, stream=0):
        b = VpFrameBuffers(_ptr(flat), _ptr(sat), _ptr(matches), _ptr(idx), _ptr(counter), _ptr(planes), _ptr(graddot), _ptr(blobcenter))
        self._ck(self.l.vp_frame(self.h, stream, C.byref(params), _ptr(raw), int(batch), C.byref(b)))


def as_torch(ptr, shape, typestr):
    """Zero-copy torch view of library-owned device memory (typestr as in numpy: "|u1", "<i4", "<f4"); the owner must outlive the view."""
    import torch

    class _View:
        pass
    v = _View()
    v.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(v, device="cuda")


class FrameSlab:
    """vp_frame_buffers_create: every buffer vp_frame writes for `batch` frames, allocated once by the library (the table and flat from
    placement-checked blocks).  `.b` is the VpFrameBuffers to hand to Context.frame_buffers(); the accessors copy single frames to the host."""

    def __init__(self, ctx, params, batch, views=0):
        self.ctx, self.batch = ctx, int(batch)
        self.fw, self.fh, self.mm = int(params.fw), int(params.fh), int(params.max_matches)
        self.b = VpFrameBuffers()
        self.h = C.c_void_p()
        ctx._ck(ctx.l.vp_frame_buffers_create(ctx.h, C.byref(params), int(batch), int(views), C.byref(self.b), C.byref(self.h)))
        sat, flat = VpPlacement(), VpPlacement()
        ctx._ck(ctx.l.vp_frame_buffers_placement(self.h, C.byref(sat), C.byref(flat)))
        self.placement = {"sat": sat.as_dict(), "flat": flat.as_dict()}

    def torch_views(self):
        """(flat, sat or None, matches, idx, counter) as torch tensors over the slab's memory."""
        B, mm = self.batch, self.mm
        return (as_torch(self.b.flat, (B, self.fh, self.fw, 4), "|u1"), as_torch(self.b.sat, (B, self.fh, self.fw), "<i4") if self.b.sat else None,
                as_torch(self.b.matches, (B, 22 * mm), "|u1"), as_torch(self.b.idx, (B, mm), "<i4"), as_torch(self.b.counter, (B, 3), "<i4"))

    def flat(self, k):
        n = self.fw * self.fh * 4
        return self.ctx.d2h(self.b.flat + k * n, (self.fh, self.fw, 4), np.uint8)

    def sat(self, k):
        n = self.fw * self.fh * 4
        return self.ctx.d2h(self.b.sat + k * n, (self.fh, self.fw), np.int32)

    def counters(self):
        return self.ctx.d2h(self.b.counter, (self.batch, 3), np.int32)

    def idx(self, k):
        return self.ctx.d2h(self.b.idx + k * 4 * self.mm, (self.mm,), np.uint32)

    def matches(self, k):
        return self.ctx.d2h(self.b.matches + k * 22 * self.mm, (22 * self.mm,), np.uint8)

    def destroy(self):
        if self.h:
            self.ctx._ck(self.ctx.l.vp_frame_buffers_destroy(self.ctx.h, self.h))
            self.h = None


def sorted_blobs(matches_bytes, idx, counter, max_matches):
    """Host view of one frame's list as (records sorted by pixel index, sorted idx): the order-independent parity key."""
    n = min(int(counter[0]), max_matches)
    m = np.frombuffer(np.ascontiguousarray(matches_bytes).tobytes(), MATCH_DTYPE)[:n]
    idx = np.asarray(idx)[:n]
    order = np.argsort(idx, kind="stable")
    return m[order], idx[order]
