"""The CPU restatement (oracle/vp_oracle.c) against the committed golden vectors, stage by stage.

The vectors were produced by the reference's own kernels compiled for x86 (tests/golden/make_golden.py); this test
runs everywhere (no /root/reference, no GPU needed)."""
import numpy as np
import pytest

import cases
import oracle as O


def _matches(buf):
    return np.frombuffer(buf.tobytes(), O.MATCH_DTYPE)


@pytest.mark.parametrize("cfg,mode,seed", cases.CHAIN_CASES, ids=[cases.chain_case_name(*c) for c in cases.CHAIN_CASES])
def test_chain_stages_bit_exact(port, cfg, mode, seed):
    g = cases.load_golden(cases.chain_case_name(cfg, mode, seed))
    assert np.array_equal(cfg.synth(seed), g["raw"]), "synthetic frame generator drifted from the fixture"
    model = O.CameraModel.from_array(g["model"])
    mrh, fs, ox, oy, thr, min_score = [float(v) for v in g["scalars"]]
    o, r, R, max_matches, smode = [int(v) for v in g["ints"]]
    assert (o, r, R, smode) == (cfg.grad_offset, cfg.circle_radius, cfg.list_radius, mode)

    planes = port.raw2quad(g["raw"], int(g["layout"]), int(g["qw"]), int(g["qh"]))
    n_planes = 3 if cfg.layout == cases.cfgs.BGR else 4
    assert np.array_equal(planes[:n_planes], g["planes"][:n_planes])
    flat = port.resample(g["planes"], int(g["layout"]), model, mrh, fs, ox, oy, smode, int(g["fw"]), int(g["fh"]))
    assert np.array_equal(flat, g["flat"])
    gd = port.graddot(g["flat"], o)
    assert np.array_equal(gd, g["graddot"].astype(np.float32))
    sat = port.sat(gd)
    assert np.array_equal(sat, g["sat"].astype(np.float32))
    sat_i, max_abs = port.sat_i32(gd)
    assert max_abs == int(g["max_abs_sat"]) < 2 ** 24
    assert np.array_equal(sat_i, g["sat"])  # int32 table == binary32 table under the guard (H2)
    bc = port.blobcenter(sat, r)
    assert np.array_equal(bc, g["blobcenter"])
    assert np.array_equal(port.blobcenter_i32(sat_i, r), g["blobcenter"])
    m, idx, cnt = port.bloblist(g["flat"], g["blobcenter"], thr, min_score, R, max_matches)
    assert np.array_equal(cnt, g["counter"])
    assert np.array_equal(idx, g["idx"])
    assert m.tobytes() == g["matches"].tobytes()
    # debug / stream colour kernels
    assert np.array_equal(port.quad2rgba(g["planes"], int(g["layout"]), smode), g["quad_rgba"])
    if "quad_nv12" in g:
        assert np.array_equal(port.quad2nv12(g["planes"], int(g["layout"]), smode), g["quad_nv12"])
    if "flat_nv12" in g:
        assert np.array_equal(port.rgba2nv12(g["flat"]), g["flat_nv12"])
        assert np.array_equal(port.f2nv12(g["blobcenter"]), g["blobcenter_nv12"])
    if "blobscore" in g:
        assert np.array_equal(port.blobscore(g["flat"], g["blobcenter"], thr, R), g["blobscore"])
        assert np.array_equal(port.circularize(gd, R, r), g["circularize"], equal_nan=True)


@pytest.mark.parametrize("name", sorted(cases.flat_cases()))
def test_flat_edge_cases_bit_exact(port, name):
    g = cases.load_golden(name)
    c = cases.flat_cases()[name]
    assert np.array_equal(c["flat"], g["flat"])
    o, r, R, max_matches = [int(v) for v in g["ints"]]
    thr, min_score = [float(v) for v in g["scalars"]]
    gd = port.graddot(g["flat"], o)
    assert np.array_equal(gd, g["graddot"].astype(np.float32))
    sat = port.sat(gd)
    sat_i, max_abs = port.sat_i32(gd)
    if "sat" in g:
        assert np.array_equal(sat, g["sat"].astype(np.float32)) and np.array_equal(sat_i, g["sat"])
    else:  # guard violated: the fp32 path must still reproduce the reference's rounding; the int path is the exact table
        assert max_abs >= 2 ** 24
        assert np.array_equal(sat, g["sat_f32"])
        assert not np.array_equal(sat_i.astype(np.float32), g["sat_f32"])
    bc = port.blobcenter(sat, r)
    assert np.array_equal(bc, g["blobcenter"])
    m, idx, cnt = port.bloblist(g["flat"], g["blobcenter"], thr, min_score, R, max_matches)
    assert np.array_equal(cnt, g["counter"]) and np.array_equal(idx, g["idx"])
    assert m.tobytes() == g["matches"].tobytes()
    gm = _matches(g["matches"])
    assert len(gm) == min(int(cnt[0]), max_matches)


def test_edge_cases_do_what_they_claim():
    g = cases.load_golden("flat_uniform_discs")
    m = _matches(g["matches"])
    assert np.isinf(m["score"]).any() and np.isnan(m["x"]).any()       # H5
    g = cases.load_golden("flat_constant_overflow")
    assert g["counter"][0] == 24 * 32 > g["ints"][3]                    # every pixel appended, list overflows
    assert np.isnan(_matches(g["matches"])["score"]).all()
    g = cases.load_golden("flat_overflow_minscore")
    assert g["counter"][0] > g["ints"][3] and g["counter"][1] > 0
    g = cases.load_golden("flat_offset0")
    assert not g["graddot"].any() and g["counter"][0] == 0
    assert "sat_f32" in cases.load_golden("flat_guard_violation")


def _sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


@pytest.mark.parametrize("mode", [0, 1], ids=["nearest", "bilinear_rtz"])
def test_config0_640x480_bgr_frame(port, mode):
    """BASELINE config 0 at its own size: one 640x480 BGR frame (OpenCV-driver layout).  The fixture holds the input, the blob list the
    reference's own compiled kernels produce and SHA-256 digests of their intermediate images; the port must reproduce all of them."""
    g = cases.load_golden("config0_640x480_bgr")
    model = O.CameraModel.from_array(g["model"])
    mrh, fs, ox, oy, thr, min_score = [float(v) for v in g["scalars"]]
    o, r, R, mm = [int(v) for v in g["ints"]]
    qw, qh, fw, fh, layout = int(g["qw"]), int(g["qh"]), int(g["fw"]), int(g["fh"]), int(g["layout"])
    assert (qw, qh, fw, fh, layout) == (640, 480, 640, 480, 2)
    planes = port.raw2quad(g["raw"], layout, qw, qh)
    assert np.array_equal(_sha(planes[:3]), g["sha_planes"])
    flat = port.resample(planes, layout, model, mrh, fs, ox, oy, mode, fw, fh)
    assert np.array_equal(_sha(flat), g[f"m{mode}_sha_flat"])
    gd = port.graddot(flat, o)
    assert np.array_equal(_sha(gd.astype(np.int32)), g[f"m{mode}_sha_graddot"])
    sat = port.sat_vertical(port.sat_horizontal(gd))
    assert np.array_equal(_sha(sat.astype(np.int32)), g[f"m{mode}_sha_sat"])
    bc = port.blobcenter(sat, r)
    assert np.array_equal(_sha(bc), g[f"m{mode}_sha_blobcenter"])
    m, idx, cnt = port.bloblist(flat, bc, thr, min_score, R, mm)
    assert np.array_equal(cnt, g[f"m{mode}_counter"]) and np.array_equal(idx, g[f"m{mode}_idx"])
    assert m.tobytes() == g[f"m{mode}_matches"].tobytes() and len(m) > 20
