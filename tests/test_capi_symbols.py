"""CPU-side checks of the drop-in boundary: the shared libraries load and export every entry point that
include/vp_hip.h declares; without a GPU the context constructor fails loudly (no fallback path exists)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import cases

ROOT = cases.ROOT
PKG = os.path.join(ROOT, "vision-processor_amd")


def declared_functions(header="vp_hip.h", prefix="vp_"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    subprocess.run(["make", "-s", "-C", ROOT, "hip", "host"], check=True)


def test_header_declares_the_hot_path():
    fns = declared_functions()
    for name in ("vp_raw2quad", "vp_resample", "vp_resample_raw", "vp_graddot", "vp_sat", "vp_graddot_sat", "vp_blobcenter", "vp_bloblist",
                 "vp_blobcenter_list", "vp_blobscore", "vp_circularize", "vp_frame", "vp_ctx_create", "vp_ctx_adopt_stream", "vp_sync", "vp_event_times"):
        assert name in fns


def test_libraries_export_every_declared_symbol(built):
    hip = C.CDLL(os.path.join(PKG, "libvp_hip.so"))
    host = C.CDLL(os.path.join(PKG, "libvp_host.so"))
    missing = [f for f in declared_functions() if not (hasattr(hip, f) or hasattr(host, f))]
    assert not missing, missing
    # the synthetic-input helpers are host-only; everything else must come from the HIP library
    for f in declared_functions():
        if not f.startswith("vp_synth_"):
            assert hasattr(hip, f), f


def test_multi_gpu_library_exports_the_gather(built):
    """include/vp_multi.h (one process, N devices, RCCL all-gather of the per-step records) lives in libvp_multi.so; libvp_hip.so
    itself stays free of the RCCL dependency.  Without a GPU the constructor fails loudly."""
    import torch  # noqa: F401  (one HIP runtime per process: the copy bundled with PyTorch, as capi.lib() does)
    C.CDLL(os.path.join(PKG, "libvp_hip.so"), mode=C.RTLD_GLOBAL)
    multi = C.CDLL(os.path.join(PKG, "libvp_multi.so"))
    fns = declared_functions("vp_multi.h", "vp_")
    assert {"vp_gather_create", "vp_gather_start", "vp_gather_wait", "vp_gather_send", "vp_gather_recv", "vp_record_layout", "vp_gather_destroy"} <= set(fns)
    missing = [f for f in fns if not hasattr(multi, f)]
    assert not missing, missing
    assert "rccl" in subprocess.run(["ldd", os.path.join(PKG, "libvp_multi.so")], capture_output=True, text=True).stdout
    assert "rccl" not in subprocess.run(["ldd", os.path.join(PKG, "libvp_hip.so")], capture_output=True, text=True).stdout
    multi.vp_record_layout.restype = C.c_size_t
    oc, om, oi = C.c_size_t(), C.c_size_t(), C.c_size_t()
    total = multi.vp_record_layout(3, 2000, C.byref(oc), C.byref(om), C.byref(oi))
    assert oc.value == 0 and om.value == 256 and oi.value == 256 + 132096 and total == oi.value + 24064 and total % 256 == 0
    if not torch.cuda.is_available():
        g = C.c_void_p()
        dev = (C.c_int * 1)(0)
        assert multi.vp_gather_create(1, dev, C.c_size_t(total), 2, C.byref(g)) == -5  # VP_ERR_NODEVICE


def test_host_library_exports_the_detection_stage(built):
    """include/vp_host.h (blob list -> SSL_DetectionFrame -> wire bytes) lives in libvp_host.so, which needs no GPU."""
    host = C.CDLL(os.path.join(PKG, "libvp_host.so"))
    fns = declared_functions("vp_host.h", "vph_")
    assert len(fns) >= 20 and "vph_detect" in fns and "vph_encode" in fns
    missing = [f for f in fns if not hasattr(host, f)]
    assert not missing, missing
    out = subprocess.run(["ldd", os.path.join(PKG, "libvp_host.so")], capture_output=True, text=True).stdout
    assert "amdhip" not in out and "vp_hip" not in out


def test_product_does_not_link_the_oracle(built):
    for so in ("libvp_hip.so", "libvp_host.so", "libvp_multi.so"):
        out = subprocess.run(["ldd", os.path.join(PKG, so)], capture_output=True, text=True).stdout
        assert "oracle" not in out and "vpref" not in out
    for dirpath, _, files in os.walk(PKG):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "vp_oracle.h" not in text and "libvporacle" not in text and "import detect_oracle" not in text, fn


def test_no_gpu_means_loud_failure(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    capi = cases.vp.load_capi()
    with pytest.raises(capi.VpError) as e:
        capi.Context(0, 1)
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)


def test_struct_layouts_match_the_reference():
    capi = cases.vp.load_capi()
    assert C.sizeof(cases.vp.hostlib.VpCameraModel) == 72  # Perspective.h:22-29
    assert capi.MATCH_DTYPE.itemsize == 22                 # main.cpp:33-41
    assert [capi.MATCH_DTYPE.fields[k][1] for k in ("x", "y", "color", "center", "circ", "score")] == [0, 4, 8, 11, 14, 18]
