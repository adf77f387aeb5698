"""CPU-side checks of the drop-in boundary: the shared libraries load and export every entry point that
include/vp_hip.h declares; without a GPU the context constructor fails loudly (no fallback path exists)."""
import ctypes as C
import os
import re
import subprocess
import sys

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
    hip = cases.vp.load_capi().lib()  # imports torch first: one HIP runtime per process (the copy PyTorch bundles)
    host = C.CDLL(os.path.join(PKG, "libvp_host.so"))
    missing = [f for f in declared_functions() if not (hasattr(hip, f) or hasattr(host, f))]
    assert not missing, missing
    # the synthetic-input helpers are host-only; everything else must come from the HIP library
    for f in declared_functions():
        if not f.startswith("vp_synth_"):
            assert hasattr(hip, f), f


def test_abi_version_and_struct_sizes(built):
    """vp_abi_version / vp_abi_sizeof: a caller built against an older header can tell before it hands over arrays of a struct that has grown."""
    capi = cases.vp.load_capi()
    hip = capi.lib()
    src = open(os.path.join(ROOT, "include", "vp_hip.h")).read()
    assert hip.vp_abi_version() == int(re.search(r"#define VP_ABI_VERSION (\d+)", src).group(1))
    hip.vp_abi_sizeof.restype = C.c_size_t
    assert hip.vp_abi_sizeof(b"VpPlacement") == C.sizeof(capi.VpPlacement) == 24
    assert hip.vp_abi_sizeof(b"VpFrameParams") == C.sizeof(capi.VpFrameParams)
    assert hip.vp_abi_sizeof(b"VpFrameBuffers") == C.sizeof(capi.VpFrameBuffers)
    assert hip.vp_abi_sizeof(b"VpMatch") == 22 and hip.vp_abi_sizeof(b"VpCameraModel") == 72 and hip.vp_abi_sizeof(b"nope") == 0


def test_wait_word_is_a_pure_host_wait(built):
    """vp_wait_word (the host half of vp_records_compact_signal) touches no device: it returns when the word holds the value, times out with VP_ERR_TIMEOUT otherwise."""
    import threading
    import time
    import numpy as np
    hip = cases.vp.load_capi().lib()
    hip.vp_wait_word.argtypes = [C.c_void_p, C.c_uint32, C.c_double]
    w = np.zeros(16, np.uint32)
    w[0] = 7
    assert hip.vp_wait_word(w.ctypes.data, 7, 0.5) == 0
    t0 = time.perf_counter()
    assert hip.vp_wait_word(w.ctypes.data, 8, 0.05) == -4          # VP_ERR_TIMEOUT
    assert 0.04 < time.perf_counter() - t0 < 2.0
    threading.Timer(0.05, lambda: w.__setitem__(0, 8)).start()      # another thread stores the value while the call spins (ctypes releases the GIL)
    assert hip.vp_wait_word(w.ctypes.data, 8, 5.0) == 0
    assert hip.vp_wait_word(None, 1, 0.01) == -1                    # VP_ERR_INVALID


def test_multi_gpu_library_exports_the_gather(built):
    """include/vp_multi.h (one process, N devices, RCCL all-gather of the per-step records) lives in libvp_multi.so; libvp_hip.so
    itself stays free of the RCCL dependency.  Without a GPU the constructor fails loudly.  (Loaded in a child process: RCCL brings the
    system HIP runtime, which must not meet the copy PyTorch bundles in the test process.)"""
    fns = declared_functions("vp_multi.h", "vp_")
    assert {"vp_gather_create", "vp_gather_start", "vp_gather_wait", "vp_gather_send", "vp_gather_recv", "vp_record_layout", "vp_gather_destroy"} <= set(fns)
    assert "rccl" in subprocess.run(["ldd", os.path.join(PKG, "libvp_multi.so")], capture_output=True, text=True).stdout
    assert "rccl" not in subprocess.run(["ldd", os.path.join(PKG, "libvp_hip.so")], capture_output=True, text=True).stdout
    code = f"""
import ctypes as C, os, json
C.CDLL(os.path.join({PKG!r}, "libvp_hip.so"), mode=C.RTLD_GLOBAL)
multi = C.CDLL(os.path.join({PKG!r}, "libvp_multi.so"))
missing = [f for f in {fns!r} if not hasattr(multi, f)]
multi.vp_record_layout.restype = C.c_size_t
oc, om, oi = C.c_size_t(), C.c_size_t(), C.c_size_t()
total = multi.vp_record_layout(3, 2000, C.byref(oc), C.byref(om), C.byref(oi))
g = C.c_void_p(); dev = (C.c_int * 1)(0)
rc = multi.vp_gather_create(1, dev, C.c_size_t(total), 2, C.byref(g))
print(json.dumps(dict(missing=missing, layout=[oc.value, om.value, oi.value, total], rc=rc)))
"""
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    import json
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert not r["missing"], r["missing"]
    assert r["layout"] == [0, 256, 256 + 132096, 256 + 132096 + 24064]
    import torch
    if not torch.cuda.is_available():
        assert r["rc"] == -5  # VP_ERR_NODEVICE: no CPU fallback


def device_disassembly(so_path, tmp_path):
    """gfx950 disassembly of every code object bundled in a HIP shared library (one offload bundle per translation unit)."""
    llvm = "/opt/rocm/lib/llvm/bin"
    fat = os.path.join(tmp_path, "fat.bin")
    subprocess.run(["objcopy", "--dump-section", f".hip_fatbin={fat}", so_path, os.path.join(tmp_path, "copy.so")], check=True)  # never without an output file: objcopy would rewrite the library in place
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for i, a in enumerate(starts):
        piece = os.path.join(tmp_path, f"bundle{i}.bin")
        open(piece, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(tmp_path, f"dev{i}.co")
        subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={piece}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        out.append(subprocess.run([f"{llvm}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout)
    return out


def device_kernel_metadata(so_path, tmp_path):
    """{demangled-ish kernel symbol: (vgprs, LDS bytes, scratch bytes)} from the AMDGPU metadata notes of every gfx950 code object in the library."""
    import yaml
    llvm = "/opt/rocm/lib/llvm/bin"
    fat = os.path.join(tmp_path, "fat_md.bin")
    subprocess.run(["objcopy", "--dump-section", f".hip_fatbin={fat}", so_path, os.path.join(tmp_path, "copy_md.so")], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    kernels = {}
    for i, a in enumerate(starts):
        piece = os.path.join(tmp_path, f"md_bundle{i}.bin")
        open(piece, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(tmp_path, f"md_dev{i}.co")
        subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={piece}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        doc = notes[notes.index("---"):]
        doc = doc[:doc.index("...")] if "..." in doc else doc
        for k in yaml.safe_load(doc).get("amdhsa.kernels", []):
            name = subprocess.run(["c++filt", k[".name"]], capture_output=True, text=True).stdout.strip() or k[".name"]
            kernels[name] = (k[".vgpr_count"] + k.get(".agpr_count", 0), k[".group_segment_fixed_size"], k[".private_segment_fixed_size"])
    return kernels


def test_hot_kernels_keep_the_occupancy_their_design_depends_on(built, tmp_path):
    """The register and LDS budgets DESIGN.md section 3 quotes, read back from the built library: the frame-pair resampling kernel at <= 64 VGPRs and 20 KB of LDS (eight
    wavefronts per SIMD; round 5 found 7 -> 8 worth 3 %), the single-pass SAT at <= 96 VGPRs (five workgroups per CU; its LDS is dynamic), the register walkers at <= 64 / <= 80,
    the record kernel within one workgroup's share, and nothing on the hot path spilling to scratch.  A compiler or source change that silently costs a wavefront fails here, on the
    CPU, before anybody measures it on the GPU."""
    md = device_kernel_metadata(os.path.join(PKG, "libvp_hip.so"), str(tmp_path))

    def pick(prefix):
        hits = {k: v for k, v in md.items() if k.startswith(prefix)}
        assert hits, (prefix, sorted(md)[:10])
        return hits

    for name, (vgpr, lds, scratch) in pick("void k_resample_staged<0, 3,").items():
        assert vgpr <= 64 and lds <= 20480 and scratch == 0, (name, vgpr, lds, scratch)
    for name, (vgpr, lds, scratch) in pick("void (anonymous namespace)::k_sat_fast2<2, 5,").items():
        hot = "<2, 5, false, false," in name  # the benchmark's instantiations; the gradDot-view / ragged-width ones may keep a few bytes of scratch
        assert vgpr <= 96 and lds == 0 and (scratch == 0 if hot else scratch <= 64), (name, vgpr, lds, scratch)
    for name, (vgpr, lds, scratch) in pick("void (anonymous namespace)::k_sat_fast2<2, 6,").items():
        assert vgpr <= 80 and scratch <= 64, (name, vgpr, scratch)  # the optional six-per-CU form buys its wavefront with a few spilled dwords
    for name, (vgpr, lds, scratch) in pick("void k_circle_walk<5,").items():
        assert vgpr <= 64 and lds <= 16384 and scratch == 0, (name, vgpr, lds, scratch)
    for name, (vgpr, lds, scratch) in pick("void k_flat_walk<5, 2>").items():
        assert vgpr <= 80 and scratch == 0, (name, vgpr)
    for name, (vgpr, lds, scratch) in pick("k_emit_queue").items():
        assert vgpr <= 64 and lds <= 65536 and scratch == 0, (name, vgpr, lds, scratch)
    spills = {k: v[2] for k, v in md.items() if v[2] and any(t in k for t in ("k_resample_staged<0, 3", "k_sat_fast2<2, 5, false, false", "k_circle_walk<5, false", "k_flat_walk<5, 2", "k_emit_queue", "k_records_compact"))}
    assert not spills, spills


def test_no_reversed_dpp_subtraction_in_the_device_code(built, tmp_path):
    """On gfx950 (ROCm 7.2) v_subrev_u32_dpp / v_subrev_f32_dpp shift the wrong operand (a - shl(e) executes as shl(a) - e; shown in
    isolation by tools/dbg/dpp_test.hip, found as wrong blobCenter values in round 2).  hipcc emits the instruction when a DPP-shifted
    value is the subtrahend of a subtraction it can fold the shift into; the kernels are written so that this never happens."""
    texts = device_disassembly(os.path.join(PKG, "libvp_hip.so"), str(tmp_path))
    assert len(texts) >= 4 and sum(t.count("_dpp") for t in texts) > 50  # the check really sees the kernels' DPP instructions
    bad = [ln.strip() for t in texts for ln in t.splitlines() if re.search(r"v_subrev\w*_dpp|v_subbrev\w*_dpp", ln)]
    assert not bad, bad[:5]


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


def test_bench_rank_launcher_stops_all_ranks_when_one_dies():
    """`python bench.py --gpus N` without a launcher starts its N ranks itself and polls them all: here (no GPU) every rank fails its first check,
    and the parent must come back at once with a non-zero exit code and that rank's stderr -- not sit in a rendezvous until a collective times out."""
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check of the launcher's failure path")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and time.time() - t0 < 60
    assert "exited with code" in out.stderr and "the other ranks were stopped" in out.stderr and "needs a GPU" in out.stderr
    assert not any(line.startswith("{") for line in out.stdout.splitlines())  # no bench line on a failed run
