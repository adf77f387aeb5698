"""The stand-in everything hangs on: oracle/ref_shim.c's OpenCL C builtins, held to the OpenCL 3.0 specification's own text.

oracle/_ref/libvpref.so is the reference's kernel/*.cl compiled unmodified and linked against oracle/ref_shim.c, this repo's stand-in for the OpenCL
builtin library.  The port, the goldens and the HIP path are all held to it -- so a misreading of the specification inside the
shim would make all four agree on a wrong answer.  This file calls every builtin the shim defines DIRECTLY (through tests/shim_probe.c: the mangled
names the compiled kernels use) on cases worked out by hand from the specification, not from the kernels:

  * OpenCL 3.0 API specification, section 8.2 "image addressing and filtering": CLK_FILTER_NEAREST takes the texel (i, j) = (floor(u), floor(v));
    CLK_ADDRESS_CLAMP_TO_EDGE clamps i to [0, w - 1]; CLK_FILTER_LINEAR takes i0 = floor(u - 0.5), i1 = i0 + 1 (each put through the address mode),
    a = frac(u - 0.5), T = (1-a)(1-b) T(i0,j0) + a(1-b) T(i1,j0) + (1-a)b T(i0,j1) + ab T(i1,j1);
  * OpenCL C 3.0 section 6.15.15 (image read / write functions): a CL_R image reads as (r, 0, 0, 1); write_imageui to CL_UNSIGNED_INT8 stores
    convert_uchar_sat(channel);
  * section 6.4.3 (conversions): _sat clamps to the destination range, NaN converts to 0, float -> integer rounds toward zero by default;
  * section 6.15.4 (common functions): min(x, y) = y < x ? y : x; section 6.15.12 / 6.15.2: atomic_inc returns the OLD value, get_global_id /
    get_global_size of a dimension the NDRange does not have are 0 / 1.

read_imageui with CLK_FILTER_LINEAR on an integer image is left UNDEFINED by the specification.  The shim's "BILINEAR_RTZ" mode is this
project's frozen definition: the section 8.2 formula in binary32, one rounding per operation in the order stated in ref_shim.c's header, truncated.
test_linear_filter_* pin it two ways: a float64 evaluation of the formula (equal except within 2e-4 of an integer, where sequential binary32
rounding may land on the other side -- counted and reported), and a numpy binary32 evaluation in the stated order (equal everywhere).

CPU only; skipped where oracle/_ref was never built (it needs /root/reference at build time).  This does not turn "parity unpinned" green: it removes the one
unreviewed link in the chain."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libvpref.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang"

NEAREST_EDGE = 0x10 | 2   # CLK_FILTER_NEAREST | CLK_NORMALIZED_COORDS_FALSE | CLK_ADDRESS_CLAMP_TO_EDGE (clang opencl-c-base.h:443-459)
NEAREST_NONE = 0x10 | 0   # ... | CLK_ADDRESS_NONE (f2nv12.cl:20)
LINEAR_EDGE = 0x20 | 2    # CLK_FILTER_LINEAR | ... | CLK_ADDRESS_CLAMP_TO_EDGE (resampling.cl:28)
FMT_U8, FMT_RGBA8, FMT_F32 = 0, 1, 2


class ProbeImage(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("fmt", C.c_int), ("data", C.c_void_p)]


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    if not os.path.exists(REF_SO) and os.path.isdir("/root/reference/kernel"):
        import oracle as O
        O.build(ref=True)  # the build container: compile the reference's kernels + the shim now (oracle/Makefile)
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libvpref.so not built (needs /root/reference)")
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang absent: the probe must be built with the compiler that built the shim")
    out = str(tmp_path_factory.mktemp("shim") / "libshimprobe.so")
    subprocess.run([CLANG, "-target", "x86_64-unknown-linux-gnu", "-O1", "-fPIC", "-shared", os.path.join(ROOT, "tests", "shim_probe.c"), "-o", out,
                    "-L" + os.path.dirname(REF_SO), "-lvpref", "-Wl,-rpath," + os.path.dirname(REF_SO)], check=True)
    lib = C.CDLL(out)
    ref = C.CDLL(REF_SO)
    lib.p_get_global_id.restype = lib.p_get_global_size.restype = C.c_ulong
    lib.p_min.restype = C.c_float
    lib.p_min.argtypes = [C.c_float, C.c_float]
    lib.p_convert_uchar_sat_f.argtypes = [C.c_float]
    lib.p_convert_uchar_sat_u.argtypes = [C.c_uint]
    lib.p_read_imageui_f.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
    lib.ref = ref
    yield lib
    ref.ref_set_sampler_mode(1)


def img(a, fmt):
    a = np.ascontiguousarray(a)
    h, w = a.shape[:2]
    im = ProbeImage(w, h, fmt, a.ctypes.data)
    im._keep = a
    return im


def rd_ui_f(lib, im, s, u, v):
    out = (C.c_uint * 4)()
    lib.p_read_imageui_f(C.byref(im), s, u, v, out)
    return tuple(out)


def rd_ui_i(lib, im, s, x, y):
    out = (C.c_uint * 4)()
    lib.p_read_imageui_i(C.byref(im), s, x, y, out)
    return tuple(out)


def rd_f_i(lib, im, s, x, y):
    out = (C.c_float * 4)()
    lib.p_read_imagef_i(C.byref(im), s, x, y, out)
    return tuple(out)


U8_4x3 = np.array([[1, 2, 3, 4], [11, 12, 13, 14], [21, 22, 23, 24]], np.uint8)  # T(i, j) = 10 j + i + 1


def test_nearest_filter_takes_floor_and_clamps_to_edge(probe):
    im = img(U8_4x3, FMT_U8)
    probe.ref.ref_set_sampler_mode(1)  # the integer-linear switch must not touch a NEAREST sampler
    for (u, v), want in {(-0.5, 0.0): 1,          # floor(-0.5) = -1 -> clamped to 0
                         (-1e-7, 0.0): 1,         # floor(-1e-7) = -1 (NOT a truncation to 0, though the texel is the same here ...)
                         (-1e-7, 1.0): 11,
                         (0.0, 0.0): 1,
                         (0.999999, 1.5): 11,     # i = 0, j = 1
                         (1.0, 1.0): 12,
                         (3.5, 2.999): 24,        # u = w - 0.5 -> i = 3
                         (4.0, 3.0): 24,          # i = w -> w - 1, j = h -> h - 1
                         (100.0, -100.0): 4}.items():
        assert rd_ui_f(probe, im, NEAREST_EDGE, u, v) == (want, 0, 0, 1), (u, v)
    # ... and where floor and truncation pick different texels: -0.5 on an image whose texel -1 does not exist but texel 0 vs 1 at 0.5 +- eps does
    assert rd_ui_f(probe, im, NEAREST_EDGE, np.nextafter(np.float32(1.0), np.float32(0.0)), 0.0)[0] == 1
    assert rd_ui_f(probe, im, NEAREST_EDGE, 1.0, 0.0)[0] == 2


def test_integer_coordinates_clamp_to_edge(probe):
    im = img(U8_4x3, FMT_U8)
    for (x, y), want in {(-1, -1): 1, (-100, 1): 11, (4, 2): 24, (2, 3): 23, (2, 1): 13, (0, 0): 1, (3, 2): 24, (1000000, -1000000): 4}.items():
        assert rd_ui_i(probe, im, NEAREST_EDGE, x, y) == (want, 0, 0, 1), (x, y)
    assert rd_ui_i(probe, im, NEAREST_NONE, 2, 1) == (13, 0, 0, 1)  # CLK_ADDRESS_NONE, in range
    assert probe.p_image_width(C.byref(im)) == 4 and probe.p_image_height(C.byref(im)) == 3


def test_channel_mapping_of_reads(probe):
    rgba = np.arange(2 * 3 * 4, dtype=np.uint8).reshape(2, 3, 4) + 100
    im = img(rgba, FMT_RGBA8)
    assert rd_ui_i(probe, im, NEAREST_EDGE, 1, 1) == tuple(int(c) for c in rgba[1, 1])   # CL_RGBA: all four channels
    assert rd_ui_i(probe, im, NEAREST_EDGE, -5, 9) == tuple(int(c) for c in rgba[1, 0])
    f = np.array([[0.5, -2.0, 3e9], [np.inf, 1e-30, 7.0]], np.float32)
    imf = img(f, FMT_F32)
    assert rd_f_i(probe, imf, NEAREST_EDGE, 1, 0) == (-2.0, 0.0, 0.0, 1.0)               # CL_R, CL_FLOAT: (r, 0, 0, 1.0)
    assert rd_f_i(probe, imf, NEAREST_EDGE, 2, 0)[0] == float(np.float32(3e9))
    assert rd_f_i(probe, imf, NEAREST_EDGE, -1, 5)[0] == math.inf                         # (-1, 5) -> (0, 1)
    assert rd_f_i(probe, imf, NEAREST_EDGE, 3, -1)[0] == float(np.float32(3e9))           # (3, -1) -> (2, 0)
    assert rd_f_i(probe, imf, NEAREST_NONE, 2, 1)[0] == 7.0


def test_write_imageui_saturates_to_eight_bits(probe):
    a = np.zeros((2, 3), np.uint8)
    im = img(a, FMT_U8)
    for x, v, want in [(0, 7, 7), (1, 256, 255), (2, 2 ** 31, 255)]:
        probe.p_write_imageui(C.byref(im), x, 1, (C.c_uint * 4)(v, 9, 9, 9))
        assert a[1, x] == want
    probe.p_write_imageui(C.byref(im), 0, 0, (C.c_uint * 4)(255, 0, 0, 0))
    assert a[0, 0] == 255 and a[0, 1] == 0
    rgba = np.zeros((1, 2, 4), np.uint8)
    im4 = img(rgba, FMT_RGBA8)
    probe.p_write_imageui(C.byref(im4), 1, 0, (C.c_uint * 4)(300, 0, 255, 256))
    assert rgba[0, 1].tolist() == [255, 0, 255, 255] and rgba[0, 0].tolist() == [0, 0, 0, 0]
    f = np.zeros((2, 2), np.float32)
    imf = img(f, FMT_F32)
    probe.p_write_imagef(C.byref(imf), 1, 0, (C.c_float * 4)(-1.5, 8.0, 8.0, 8.0))
    assert f.tolist() == [[0.0, -1.5], [0.0, 0.0]]  # CL_R: channel x is stored


def test_conversions(probe):
    nan, inf = float("nan"), float("inf")
    for v, want in [(nan, 0), (-0.0, 0), (0.0, 0), (0.999, 0), (1.0, 1), (254.999, 254), (255.0, 255), (255.5, 255), (1e10, 255), (inf, 255), (-inf, 0), (-3.5, 0), (127.5, 127)]:
        assert probe.p_convert_uchar_sat_f(v) == want, v   # saturating, round toward zero, NaN -> 0 (section 6.4.3)
    for v, want in [(-1, 0), (300, 255), (77, 77), (-2 ** 31, 0), (255, 255), (256, 255)]:
        assert probe.p_convert_uchar_sat_i(v) == want, v
    for v, want in [(0, 0), (255, 255), (256, 255), (2 ** 32 - 1, 255)]:
        assert probe.p_convert_uchar_sat_u(v) == want, v
    out = (C.c_float * 4)()
    probe.p_convert_float4((C.c_uint * 4)(0, 1, 16777217, 2 ** 32 - 1), out)
    assert list(out) == [0.0, 1.0, 16777216.0, 4294967296.0]   # uint -> float rounds to nearest even
    probe.p_native_sqrt((C.c_float * 4)(0.0, 4.0, 2.0, 1e-20), out)
    assert list(out) == [float(x) for x in np.sqrt(np.array([0.0, 4.0, 2.0, 1e-20], np.float32))]


def test_min_is_the_specifications_formula(probe):
    nan = float("nan")
    assert probe.p_min(1.0, 2.0) == 1.0 and probe.p_min(2.0, 1.0) == 1.0 and probe.p_min(-5.0, -5.0) == -5.0
    # min(x, y) = y < x ? y : x, applied literally (the specification leaves NaN operands undefined; the kernels' operands are finite)
    assert math.isnan(probe.p_min(nan, 1.0))        # 1 < NaN is false -> x = NaN
    assert probe.p_min(1.0, nan) == 1.0             # NaN < 1 is false -> x = 1
    assert math.copysign(1.0, probe.p_min(-0.0, 0.0)) == -1.0 and math.copysign(1.0, probe.p_min(0.0, -0.0)) == 1.0   # -0 < 0 is false either way: x


def test_atomic_inc_and_work_item_functions(probe):
    c = C.c_int(41)
    assert probe.p_atomic_inc(C.byref(c)) == 41 and c.value == 42    # returns the OLD value
    assert probe.p_atomic_inc(C.byref(c)) == 42 and c.value == 43
    assert probe.p_get_global_id(2) == 0 and probe.p_get_global_size(2) == 1 and probe.p_get_global_id(7) == 0 and probe.p_get_global_size(7) == 1
    # the NDRange loops: a 1-D launch of h work-items (satHorizontal.cl, Resources.cpp:161) leaves the last id and the size behind in this thread
    a = np.arange(15, dtype=np.float32).reshape(3, 5)
    out = np.zeros_like(a)
    probe.ref.ref_sat_horizontal(a.ctypes.data_as(C.c_void_p), 5, 3, out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, np.cumsum(a, axis=1))
    assert probe.p_get_global_size(0) == 3 and probe.p_get_global_id(0) == 2 and probe.p_get_global_size(1) == 1 and probe.p_get_global_id(1) == 0


def test_linear_filter_by_hand(probe):
    probe.ref.ref_set_sampler_mode(1)
    im = img(np.array([[0, 100], [200, 50]], np.uint8), FMT_U8)
    for (u, v), want in {(0.5, 0.5): 0,      # a texel centre: a = b = 0
                         (1.5, 1.5): 50,
                         (1.0, 0.5): 50,     # halfway between T(0,0) and T(1,0)
                         (1.25, 0.5): 75,
                         (0.5, 1.0): 100,    # halfway between T(0,0) = 0 and T(0,1) = 200
                         (1.0, 1.0): 87,     # (0 + 100 + 200 + 50) / 4 = 87.5, truncated
                         (-3.0, 0.5): 0,     # i0 = -4, i1 = -3: both clamp to column 0
                         (5.0, 0.5): 100,    # both clamp to column 1
                         (0.25, 5.0): 200,   # u - 0.5 = -0.25: i0 = -1 -> 0, i1 = 0: column 0 whatever a is; rows clamp to 1
                         (2.0, -1.0): 100}.items():
        assert rd_ui_f(probe, im, LINEAR_EDGE, u, v) == (want, 0, 0, 1), (u, v)
    probe.ref.ref_set_sampler_mode(0)        # the recorded switch: CLK_FILTER_LINEAR on an integer image degraded to nearest
    assert rd_ui_f(probe, im, LINEAR_EDGE, 1.0, 1.0) == (50, 0, 0, 1) and rd_ui_f(probe, im, LINEAR_EDGE, 0.99, 1.0) == (200, 0, 0, 1)
    probe.ref.ref_set_sampler_mode(1)


def _coords(rng, w, h, n):
    uv = np.stack([rng.uniform(-2.0, w + 2.0, n), rng.uniform(-2.0, h + 2.0, n)], axis=1).astype(np.float32)
    k = n // 10
    uv[:k] = np.floor(uv[:k]) + np.float32(0.5)                      # texel centres, borders included
    uv[k:2 * k] = np.floor(uv[k:2 * k])                              # texel corners: a = b = 0.5
    uv[2 * k:3 * k, 0] = rng.choice([-0.5, 0.0, 0.5, w - 0.5, w, w + 0.5], k).astype(np.float32)   # the columns where clamp-to-edge starts to act
    return np.ascontiguousarray(uv)


def test_linear_filter_against_independent_evaluations(probe):
    probe.ref.ref_set_sampler_mode(1)
    rng = np.random.default_rng(20260106)
    w, h, n = 37, 23, 10000
    tex = rng.integers(0, 256, (h, w), dtype=np.uint8)
    tex[::5, ::3] = 255
    im = img(tex, FMT_U8)
    uv = _coords(rng, w, h, n)
    got = np.zeros(n, np.uint32)
    probe.p_read_imageui_f_many(C.byref(im), LINEAR_EDGE, uv.ctypes.data_as(C.c_void_p), n, got.ctypes.data_as(C.c_void_p))

    def taps(fu, fv):
        i0, i1 = np.clip(fu, 0, w - 1).astype(int), np.clip(fu + 1, 0, w - 1).astype(int)
        j0, j1 = np.clip(fv, 0, h - 1).astype(int), np.clip(fv + 1, 0, h - 1).astype(int)
        return tex[j0, i0], tex[j0, i1], tex[j1, i0], tex[j1, i1]

    # (i) section 8.2 in float64, nothing rounded before the end
    u, v = uv[:, 0].astype(np.float64) - 0.5, uv[:, 1].astype(np.float64) - 0.5
    fu, fv = np.floor(u), np.floor(v)
    a, b = u - fu, v - fv
    t00, t10, t01, t11 = (t.astype(np.float64) for t in taps(fu, fv))
    T = (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11
    want64 = np.floor(T).astype(np.uint32)
    differ = got != want64
    near_integer = np.abs(T - np.rint(T)) < 2e-4
    assert np.all(near_integer[differ]), "a result differs from the float64 evaluation away from an integer boundary"
    assert np.all(np.abs(got.astype(np.float64) - T) < 1.0 + 2e-4)
    assert differ.mean() < 0.02, differ.mean()
    print(f"\nlin_u8 vs float64 section 8.2: {int(differ.sum())} of {n} coordinates land on the other side of an integer (all within 2e-4 of it): the frozen BILINEAR_RTZ definition is the binary32 one")

    # (ii) the frozen definition itself: binary32, one rounding per operation, in the order of ref_shim.c's header
    f = np.float32
    um, vm = uv[:, 0] - f(0.5), uv[:, 1] - f(0.5)
    fu, fv = np.floor(um), np.floor(vm)
    a, b = um - fu, vm - fv
    t00, t10, t01, t11 = (t.astype(np.float32) for t in taps(fu, fv))
    oma, omb = f(1.0) - a, f(1.0) - b
    w00, w10, w01, w11 = oma * omb, a * omb, oma * b, a * b
    t = w00 * t00
    t = t + w10 * t10
    t = t + w01 * t01
    t = t + w11 * t11
    assert t.dtype == np.float32
    assert np.array_equal(got, t.astype(np.uint32))
