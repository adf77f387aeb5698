"""GPU parity tests: the HIP path, called through the C ABI (libvp_hip.so), against
  (a) the committed golden vectors (generated from the reference's own kernels),
  (b) the CPU oracle on the same seeded inputs at sizes it finishes in seconds,
  (c) size-independent properties at BASELINE.json's full 2048x1536 size.
Bar: integer / byte / index outputs bit-exact; float circularity / centroid / score within 1e-4 (abs-or-rel, NaN==NaN,
inf==inf) -- in practice they are bit-exact too and the tests say so where they check it."""
import numpy as np
import pytest

import cases
import oracle as O

pytestmark = pytest.mark.gpu
cfgs = cases.cfgs


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    capi = cases.vp.load_capi()
    ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
    yield capi, ctx, torch
    ctx.close()


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


import os

VARIANTS = [("twopass", 1), ("tiled", 0)]
if os.environ.get("VP_TEST_SAT"):  # e.g. VP_TEST_SAT=twopass while bisecting a variant
    VARIANTS = [v for v in VARIANTS if v[0] in os.environ["VP_TEST_SAT"].split(",")]
DEFAULT_VARIANT = VARIANTS[-1][1]


def run_tail(capi, ctx, torch, flat, bc, w, h, thr, min_score, R, max_matches, order):
    d_m = torch.zeros(max_matches * 22, dtype=torch.uint8, device="cuda")
    d_idx = torch.zeros(max_matches, dtype=torch.int32, device="cuda")
    d_cnt = torch.full((3,), -7, dtype=torch.int32, device="cuda")
    ctx.bloblist(flat, bc, w, h, 1, thr, min_score, R, max_matches, order, d_m, d_idx, d_cnt)
    ctx.sync()
    return host(d_m), host(d_idx).view(np.uint32), host(d_cnt)


def assert_list_equal(capi, got, want_matches_bytes, want_idx, want_cnt, max_matches, overflow_ok=False):
    m_b, idx, cnt = got
    assert np.array_equal(cnt, want_cnt), (cnt, want_cnt)
    gm, gi = capi.sorted_blobs(m_b, idx, cnt, max_matches)
    wm = np.frombuffer(np.ascontiguousarray(want_matches_bytes).tobytes(), capi.MATCH_DTYPE)
    if int(cnt[0]) > max_matches:
        # list overflow: which records survive is run-dependent in the reference too (blobList.cl:87-89); every survivor must be genuine
        assert overflow_ok and len(gm) == max_matches and len(np.unique(gi)) == max_matches
        return
    assert np.array_equal(gi, want_idx)
    assert np.array_equal(gm["color"], wm["color"]) and np.array_equal(gm["center"], wm["center"])
    for k in ("x", "y", "circ", "score"):
        assert np.allclose(gm[k], wm[k], rtol=1e-4, atol=1e-4, equal_nan=True), k
    assert gm.tobytes() == wm.tobytes(), "floats are expected to be bit-exact as well"


@pytest.mark.parametrize("cfg,mode,seed", cases.CHAIN_CASES, ids=[cases.chain_case_name(*c) for c in cases.CHAIN_CASES])
def test_golden_chain_stage_by_stage(gpu, cfg, mode, seed):
    capi, ctx, torch = gpu
    g = cases.load_golden(cases.chain_case_name(cfg, mode, seed))
    model = cases.vp.hostlib.VpCameraModel.from_buffer_copy(g["model"].tobytes())
    mrh, fs, ox, oy, thr, min_score = [float(v) for v in g["scalars"]]
    o, r, R, max_matches, smode = [int(v) for v in g["ints"]]
    qw, qh, fw, fh, layout = int(g["qw"]), int(g["qh"]), int(g["fw"]), int(g["fh"]), int(g["layout"])
    n_planes = 3 if layout == cfgs.BGR else 4

    d_raw = dev(torch, g["raw"])
    d_planes = torch.zeros((4, qh, qw), dtype=torch.uint8, device="cuda")
    ctx.raw2quad(d_raw, layout, qw, qh, 1, d_planes)
    assert np.array_equal(host(d_planes)[:n_planes], g["planes"][:n_planes])

    d_flat = torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda")
    ctx.resample(dev(torch, g["planes"]), layout, qw, qh, model, mrh, fs, ox, oy, smode, fw, fh, 1, d_flat)
    assert np.array_equal(host(d_flat), g["flat"])
    d_flat2 = torch.zeros_like(d_flat)
    ctx.resample_raw(d_raw, layout, qw, qh, model, mrh, fs, ox, oy, smode, fw, fh, 1, d_flat2)
    assert np.array_equal(host(d_flat2), g["flat"]), "fused raw2quad+resampling"
    d_map = torch.zeros((fh, fw, 2), dtype=torch.float32, device="cuda")
    ctx.resample_map(model, mrh, fs, ox, oy, fw, fh, d_map)
    d_flat3 = torch.zeros_like(d_flat)
    ctx.resample_raw_mapped(d_raw, layout, qw, qh, d_map, smode, fw, fh, 1, d_flat3)
    assert np.array_equal(host(d_flat3), g["flat"]), "coordinate-map resampling"

    d_flat = dev(torch, g["flat"])
    d_gd = torch.zeros((fh, fw), dtype=torch.float32, device="cuda")
    ctx.graddot(d_flat, fw, fh, o, 1, d_gd)
    assert np.array_equal(host(d_gd), g["graddot"].astype(np.float32))

    for name, variant in VARIANTS:
        d_sat = torch.zeros((fh, fw), dtype=torch.int32, device="cuda")
        ctx.sat(d_gd, fw, fh, 1, variant, d_sat)
        assert np.array_equal(host(d_sat), g["sat"]), f"sat {name}"
        d_sat2 = torch.zeros_like(d_sat)
        d_gd2 = torch.zeros_like(d_gd)
        ctx.graddot_sat(d_flat, fw, fh, o, 1, variant, d_sat2, d_gd2)
        assert np.array_equal(host(d_sat2), g["sat"]), f"fused graddot+sat {name}"
        assert np.array_equal(host(d_gd2), g["graddot"].astype(np.float32))

    d_sat = dev(torch, g["sat"])
    d_bc = torch.zeros((fh, fw), dtype=torch.float32, device="cuda")
    ctx.blobcenter(d_sat, fw, fh, r, 1, d_bc)
    assert np.array_equal(host(d_bc), g["blobcenter"])

    d_bc = dev(torch, g["blobcenter"])
    for order in (capi.ORDER_ATOMIC, capi.ORDER_RASTER):
        got = run_tail(capi, ctx, torch, d_flat, d_bc, fw, fh, thr, min_score, R, max_matches, order)
        assert_list_equal(capi, got, g["matches"], g["idx"], g["counter"], max_matches)
        if order == capi.ORDER_RASTER:  # RASTER is already sorted: byte-identical to the golden list
            n = int(got[2][0])
            assert got[0][: 22 * n].tobytes() == g["matches"].tobytes() and np.array_equal(got[1][:n], g["idx"])

    if "blobscore" in g:
        d_s = torch.zeros((fh, fw), dtype=torch.float32, device="cuda")
        ctx.blobscore(d_flat, d_bc, fw, fh, 1, thr, R, d_s)
        assert np.array_equal(host(d_s), g["blobscore"])
        ctx.circularize(d_gd, fw, fh, 1, R, r, d_s)
        assert np.array_equal(host(d_s), g["circularize"], equal_nan=True)


@pytest.mark.parametrize("cfg,mode,seed", cases.CHAIN_CASES, ids=[cases.chain_case_name(*c) for c in cases.CHAIN_CASES])
@pytest.mark.parametrize("variant", VARIANTS, ids=lambda v: v[0])
def test_golden_whole_frame(gpu, cfg, mode, seed, variant):
    capi, ctx, torch = gpu
    g = cases.load_golden(cases.chain_case_name(cfg, mode, seed))
    p = capi.frame_params(cfg, sat_variant=variant[1], order=capi.ORDER_RASTER, sampler_mode=mode, use_coord_map=(seed + mode) % 2 == 0)
    mm = cfg.max_matches
    d = dict(flat=torch.zeros((cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.zeros((cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
             matches=torch.zeros(mm * 22, dtype=torch.uint8, device="cuda"), idx=torch.zeros(mm, dtype=torch.int32, device="cuda"),
             counter=torch.zeros(3, dtype=torch.int32, device="cuda"))
    # hot path: nothing optional materialised
    ctx.frame(p, dev(torch, g["raw"]), 1, **d)
    ctx.sync()
    assert np.array_equal(host(d["flat"]), g["flat"]) and np.array_equal(host(d["sat"]), g["sat"])
    assert_list_equal(capi, (host(d["matches"]), host(d["idx"]).view(np.uint32), host(d["counter"])), g["matches"], g["idx"], g["counter"], mm)
    # with every debug view requested (the Resources::rgba2blobCenter contract returns flat, gradDot, blobCenter)
    views = dict(planes=torch.zeros((4, cfg.qh, cfg.qw), dtype=torch.uint8, device="cuda"), graddot=torch.zeros((cfg.fh, cfg.fw), dtype=torch.float32, device="cuda"),
                 blobcenter=torch.zeros((cfg.fh, cfg.fw), dtype=torch.float32, device="cuda"))
    ctx.frame(p, dev(torch, g["raw"]), 1, **d, **views)
    ctx.sync()
    n_planes = 3 if cfg.layout == cfgs.BGR else 4
    assert np.array_equal(host(views["planes"])[:n_planes], g["planes"][:n_planes])
    assert np.array_equal(host(views["graddot"]), g["graddot"].astype(np.float32))
    assert np.array_equal(host(views["blobcenter"]), g["blobcenter"])
    assert_list_equal(capi, (host(d["matches"]), host(d["idx"]).view(np.uint32), host(d["counter"])), g["matches"], g["idx"], g["counter"], mm)


@pytest.mark.parametrize("name", sorted(cases.flat_cases()))
@pytest.mark.parametrize("variant", VARIANTS, ids=lambda v: v[0])
def test_golden_flat_edge_cases(gpu, name, variant):
    capi, ctx, torch = gpu
    g = cases.load_golden(name)
    o, r, R, max_matches = [int(v) for v in g["ints"]]
    thr, min_score = [float(v) for v in g["scalars"]]
    h, w = g["flat"].shape[:2]
    d_flat = dev(torch, g["flat"])
    d_sat = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    d_gd = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    ctx.graddot_sat(d_flat, w, h, o, 1, variant[1], d_sat, d_gd)
    assert np.array_equal(host(d_gd), g["graddot"].astype(np.float32))
    exact_sat, _ = O.Port().sat_i32(g["graddot"].astype(np.float32))
    assert np.array_equal(host(d_sat), exact_sat), "int32 wrap-around table"
    d_bc = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    ctx.blobcenter(d_sat, w, h, r, 1, d_bc)
    if "sat" in g:
        assert np.array_equal(host(d_sat), g["sat"])
        assert np.array_equal(host(d_bc), g["blobcenter"])
    else:
        # 2^24 guard violated (H2): the reference's own binary32 table has lost integers; the exact table may differ from it
        # by the reference's rounding error only.  Documented tolerance: |delta blobCenter| <= 16 * ulp(max|SAT|) / r^2.
        tol = 16 * np.spacing(np.float32(g["max_abs_sat"])) / (r * r)
        assert np.abs(host(d_bc) - g["blobcenter"]).max() <= tol
        assert np.array_equal(host(d_bc), O.Port().blobcenter_i32(exact_sat, r))
    # the list stage consumes blobCenter as given: unconditional parity
    d_bc = dev(torch, g["blobcenter"])
    for order in (capi.ORDER_ATOMIC, capi.ORDER_RASTER):
        got = run_tail(capi, ctx, torch, d_flat, d_bc, w, h, thr, min_score, R, max_matches, order)
        assert_list_equal(capi, got, g["matches"], g["idx"], g["counter"], max_matches, overflow_ok=True)


@pytest.mark.parametrize("cfg", [cfgs.T_MID, cfgs.CFG1], ids=lambda c: c.name)
def test_live_oracle_mid_sizes(gpu, port, cfg):
    """Seeded inputs not in the fixtures, checked against the CPU oracle run in the same process."""
    capi, ctx, torch = gpu
    model_o = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    B = 3
    raws = [cfg.synth(100 + i) for i in range(B)]
    p = capi.frame_params(cfg, sat_variant=DEFAULT_VARIANT, order=capi.ORDER_RASTER)
    mm = cfg.max_matches
    d_raw = dev(torch, np.stack(raws))
    d = dict(flat=torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
             matches=torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"), idx=torch.zeros((B, mm), dtype=torch.int32, device="cuda"),
             counter=torch.zeros((B, 3), dtype=torch.int32, device="cuda"))
    ctx.frame(p, d_raw, B, **d)
    ctx.sync()
    for i in range(B):
        m, idx, cnt = port.frame(raws[i], cfg.layout, cfg.qw, cfg.qh, model_o, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                                 cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm)
        assert len(m) > 5
        assert_list_equal(capi, (host(d["matches"][i]), host(d["idx"][i]).view(np.uint32), host(d["counter"][i])), np.frombuffer(m.tobytes(), np.uint8), idx, cnt, mm)


RESAMPLE_BATCH_CASES = [
    (cfgs.T_RGGB, 9),                                                      # 8 + 1: a frame group with a single frame
    (cfgs.T_GRBG, 3),
    (cfgs.T_RAGGED.with_(name="t_ragged_even", qw=68, qh=46), 10),          # even rows -> staged; flat overhangs all four sensor edges
    (cfgs.T_RAGGED, 5),                                                    # odd quad width -> unstaged kernel
    (cfgs.T_MID.with_(name="t_mid_wide", visible_len=9000.0), 4),          # strong minification: large bounding boxes per workgroup
    (cfgs.T_BGR, 3),
]


@pytest.fixture(params=["auto", "xcd_bands", "plain"])
def rs_tile_mapping(request):
    """The staged resampling kernels pick XCD-contiguous tile bands for large frames only; force both mappings on the small test
    frames as well (the tile count per frame group is then not a multiple of 8: the remap must stay a bijection)."""
    return {"auto": -1, "xcd_bands": 1, "plain": 0}[request.param]


@pytest.mark.parametrize("cfg,B", RESAMPLE_BATCH_CASES, ids=[c.name for c, _ in RESAMPLE_BATCH_CASES])
@pytest.mark.parametrize("mode", [0, 1], ids=["nearest", "bilinear_rtz"])
def test_resample_mapped_batches(gpu, port, cfg, B, mode, rs_tile_mapping):
    """Coordinate-map resampling over whole batches (the staged LDS kernel for Bayer + BILINEAR_RTZ) against the CPU oracle, per frame."""
    capi, ctx, torch = gpu
    ctx.tune("resample_xcd_bands", rs_tile_mapping)
    model = cfg.model()
    model_o = O.CameraModel.from_array(np.frombuffer(bytes(model), np.uint8))
    raws = [cfg.synth(300 + 7 * i) for i in range(B)]
    # make the sensor borders count: saturate a frame of pixels so clamp-to-edge taps are distinguishable
    for r in raws:
        r[:2], r[-2:], r[:, :2], r[:, -2:] = 250, 5, 3, 247
    d_raw = dev(torch, np.stack(raws))
    d_map = torch.zeros((cfg.fh, cfg.fw, 2), dtype=torch.float32, device="cuda")
    ctx.resample_map(model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh, d_map)
    d_flat = torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
    try:
        ctx.resample_raw_mapped(d_raw, cfg.layout, cfg.qw, cfg.qh, d_map, mode, cfg.fw, cfg.fh, B, d_flat)
        ctx.sync()
    finally:
        ctx.tune("resample_xcd_bands", -1)
    got = host(d_flat)
    for i in range(B):
        planes = port.raw2quad(raws[i], cfg.layout, cfg.qw, cfg.qh)
        want = port.resample(planes, cfg.layout, model_o, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, mode, cfg.fw, cfg.fh)
        assert np.array_equal(got[i], want), f"frame {i}"


@pytest.mark.parametrize("cfg,mode,seed", cases.CHAIN_CASES, ids=[cases.chain_case_name(*c) for c in cases.CHAIN_CASES])
def test_golden_colour_kernels(gpu, cfg, mode, seed):
    """Debug / stream colour kernels through the C ABI against the golden vectors."""
    capi, ctx, torch = gpu
    g = cases.load_golden(cases.chain_case_name(cfg, mode, seed))
    qw, qh, fw, fh, layout = int(g["qw"]), int(g["qh"]), int(g["fw"]), int(g["fh"]), int(g["layout"])
    d_planes, d_raw = dev(torch, g["planes"]), dev(torch, g["raw"])
    d_rgba = torch.zeros((qh, qw, 4), dtype=torch.uint8, device="cuda")
    ctx.quad2rgba(d_planes, layout, qw, qh, mode, 1, d_rgba)
    assert np.array_equal(host(d_rgba), g["quad_rgba"])
    d_rgba.zero_()
    ctx.raw2rgba(d_raw, layout, qw, qh, mode, 1, d_rgba)
    assert np.array_equal(host(d_rgba), g["quad_rgba"]), "fused raw2quad+quad2rgba"
    d_rgba.zero_()
    pl = [d_planes[i] for i in range(4)]
    ctx.quad2rgba4(pl[0], pl[1], pl[2], pl[3] if layout != cfgs.BGR else None, layout, qw, qh, mode, d_rgba)
    assert np.array_equal(host(d_rgba), g["quad_rgba"]), "separately allocated planes"
    if "quad_nv12" in g:
        assert ctx.nv12_bytes(qw, qh) == g["quad_nv12"].size
        for fn, src in ((ctx.quad2nv12, d_planes), (ctx.raw2nv12, d_raw)):
            d_nv = torch.full((g["quad_nv12"].size,), 0xEE, dtype=torch.uint8, device="cuda")
            fn(src, layout, qw, qh, mode, 1, d_nv)
            assert np.array_equal(host(d_nv), g["quad_nv12"]), fn.__name__
    else:
        with pytest.raises(capi.VpError):
            ctx.quad2nv12(d_planes, layout, qw, qh, mode, 1, torch.zeros(ctx.nv12_bytes(qw, qh), dtype=torch.uint8, device="cuda"))  # odd width is refused
    if "flat_nv12" in g:
        d_nv = torch.full((g["flat_nv12"].size,), 0xEE, dtype=torch.uint8, device="cuda")
        ctx.rgba2nv12(dev(torch, g["flat"]), fw, fh, 1, d_nv)
        assert np.array_equal(host(d_nv), g["flat_nv12"])
        d_nv.fill_(0xEE)
        ctx.f2nv12(dev(torch, g["blobcenter"]), fw, fh, 1, d_nv)
        assert np.array_equal(host(d_nv), g["blobcenter_nv12"])


def test_colour_kernels_batches_and_odd_heights(gpu, port):
    """Batched launches, odd heights (last chroma row owned by a single luma row), special floats; live against the CPU oracle."""
    capi, ctx, torch = gpu
    rng = np.random.default_rng(5)
    B, qw, qh = 3, 70, 37
    planes = rng.integers(0, 256, (B, 4, qh, qw), dtype=np.uint8)
    d_planes = dev(torch, planes)
    nb = ctx.nv12_bytes(qw, qh)
    for layout in (cfgs.RGGB, cfgs.GRBG, cfgs.BGR):
        for mode in (0, 1):
            d_rgba = torch.zeros((B, qh, qw, 4), dtype=torch.uint8, device="cuda")
            d_nv = torch.zeros((B, nb), dtype=torch.uint8, device="cuda")
            ctx.quad2rgba(d_planes, layout, qw, qh, mode, B, d_rgba)
            ctx.quad2nv12(d_planes, layout, qw, qh, mode, B, d_nv)
            for i in range(B):
                assert np.array_equal(host(d_rgba[i]), port.quad2rgba(planes[i], layout, mode)), (layout, mode, i)
                assert np.array_equal(host(d_nv[i]), port.quad2nv12(planes[i], layout, mode)), (layout, mode, i)
    rgba = rng.integers(0, 256, (B, qh, qw, 4), dtype=np.uint8)
    f = (rng.standard_normal((B, qh, qw)) * 200).astype(np.float32)
    f[0, 0, :6] = [np.nan, np.inf, -np.inf, 127.999, -127.0, 128.0]
    d_nv = torch.zeros((B, nb), dtype=torch.uint8, device="cuda")
    ctx.rgba2nv12(dev(torch, rgba), qw, qh, B, d_nv)
    for i in range(B):
        assert np.array_equal(host(d_nv[i]), port.rgba2nv12(rgba[i]))
    ctx.f2nv12(dev(torch, f), qw, qh, B, d_nv)
    for i in range(B):
        assert np.array_equal(host(d_nv[i]), port.f2nv12(f[i]))


@pytest.mark.parametrize("shape", [(64, 5), (64, 64), (128, 70), (192, 129), (256, 63), (66, 64), (130, 129), (150, 98), (1000, 70), (63, 40), (2, 3)], ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("o", [0, 1, 2, 3, 4, 6])
def test_fast_sat_edge_shapes(gpu, port, shape, o):
    """The single-pass kernels on awkward shapes: one tile column, fewer rows than the halo, heights that overhang the last tile
    row, with and without halo strips, with and without the gradDot view, at 5 and 6 resident workgroups per CU (offsets 1..4 run
    the fast kernel, 0 and 6 the generic one)."""
    capi, ctx, torch = gpu
    w, h = shape
    rng = np.random.default_rng(w * 1000 + h * 10 + o)
    B = 3
    flat = rng.integers(0, 256, (B, h, w, 4), dtype=np.uint8)
    flat[..., 3] = 255
    d_flat = dev(torch, flat)
    want_gd = [port.graddot(flat[i], o) for i in range(B)]
    want_sat = [g.astype(np.int64).cumsum(0).cumsum(1).astype(np.int32) for g in want_gd]
    nbytes = ctx.halo_strip_bytes(w, h, o)
    assert (nbytes > 0) == (w % 64 == 0 and 1 <= o <= 4)  # other widths / offsets run without strips
    d_strips = torch.zeros((B, max(nbytes, 4) // 4), dtype=torch.int32, device="cuda")
    if nbytes:
        ctx.halo_strips(d_flat, w, h, o, B, d_strips)
    for gen, with_strips, with_gd in [(g, s_, d) for g in (5, 6) for s_ in ((False, True) if nbytes else (False,)) for d in (False, True)]:
        ctx.tune("sat_workgroups_per_cu", gen)  # the two register budgets the fast kernel is built for
        try:
            d_sat = torch.full((B, h, w), -7, dtype=torch.int32, device="cuda")
            d_gd = torch.zeros((B, h, w), dtype=torch.float32, device="cuda") if with_gd else None
            if with_strips:
                ctx.graddot_sat_strips(d_flat, d_strips, w, h, o, B, capi.SAT_TILED, d_sat, d_gd)
            else:
                ctx.graddot_sat(d_flat, w, h, o, B, capi.SAT_TILED, d_sat, d_gd)
            ctx.sync()
        finally:
            ctx.tune("sat_workgroups_per_cu", 0)
        for i in range(B):
            assert np.array_equal(host(d_sat[i]), want_sat[i]), (gen, with_strips, with_gd, i)
            if with_gd:
                assert np.array_equal(host(d_gd[i]), want_gd[i])


def test_abi_buffer_and_image_objects(gpu):
    """VpBuffer / VpImage (the objects behind CLArray and the CLImage pool): blocking map / unmap round trips, initial contents,
    and the pool semantics of OpenCL::acquire (opencl.cpp:108-122): an idle image of the same format and size is handed out again."""
    import ctypes as C
    capi, ctx, torch = gpu
    lib = capi.lib()
    lib.vp_buffer_device.restype = C.c_void_p
    lib.vp_image_device.restype = C.c_void_p
    lib.vp_buffer_size.restype = C.c_size_t
    h = ctx.h
    data = np.arange(1000, dtype=np.uint8)
    buf = C.c_void_p()
    assert lib.vp_buffer_create(h, C.c_size_t(1000), data.ctypes.data_as(C.c_void_p), C.byref(buf)) == 0
    assert lib.vp_buffer_size(buf) == 1000
    hp = C.c_void_p()
    assert lib.vp_buffer_map(h, 0, buf, 1, C.byref(hp)) == 0  # read back the initial contents
    view = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(1000,))
    assert np.array_equal(view, data)
    view[:] = 255 - view
    assert lib.vp_buffer_unmap(h, 0, buf, 1) == 0             # write back
    dev = torch.zeros(1000, dtype=torch.uint8, device="cuda")
    assert lib.vp_copy_d2h_sync(h, 0, hp, C.c_void_p(lib.vp_buffer_device(buf)), C.c_size_t(1000)) == 0
    assert np.array_equal(np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(1000,)), 255 - data)
    assert lib.vp_buffer_destroy(h, buf) == 0
    del dev
    # image pool
    a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.vp_image_acquire(h, 2, 64, 48, C.byref(a)) == 0   # VP_FMT_F32
    assert lib.vp_image_acquire(h, 2, 64, 48, C.byref(b)) == 0
    assert a.value != b.value and lib.vp_image_device(a) != lib.vp_image_device(b)
    fmt, w, hh, nb = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
    assert lib.vp_image_info(a, C.byref(fmt), C.byref(w), C.byref(hh), C.byref(nb)) == 0
    assert (fmt.value, w.value, hh.value, nb.value) == (2, 64, 48, 64 * 48 * 4)
    assert lib.vp_image_release(h, a) == 0
    assert lib.vp_image_acquire(h, 2, 64, 48, C.byref(c)) == 0 and c.value == a.value  # the idle one comes back
    d = C.c_void_p()
    assert lib.vp_image_acquire(h, 1, 64, 48, C.byref(d)) == 0 and d.value not in (a.value, b.value)  # other format: a new image
    assert lib.vp_image_map(h, 0, c, 0, C.byref(hp)) == 0
    fv = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_float)), shape=(48, 64))
    fv[:] = np.arange(48 * 64, dtype=np.float32).reshape(48, 64)
    assert lib.vp_image_unmap(h, 0, c, 1) == 0
    fv[:] = 0
    assert lib.vp_image_map(h, 0, c, 1, C.byref(hp)) == 0
    assert np.array_equal(np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_float)), shape=(48, 64)), np.arange(48 * 64, dtype=np.float32).reshape(48, 64))
    assert lib.vp_image_unmap(h, 0, c, 0) == 0
    for im in (b, c, d):
        assert lib.vp_image_release(h, im) == 0
    assert lib.vp_image_acquire(h, 7, 4, 4, C.byref(d)) != 0  # unknown format is refused


def test_full_size_properties(gpu, port):
    """BASELINE.json configs[1] (2048x1536): properties that do not need a full CPU reference run per frame."""
    capi, ctx, torch = gpu
    cfg = cfgs.CFG2
    B = 2
    raws = [cfg.synth(i) for i in range(B)]
    d_raw = dev(torch, np.stack(raws))
    mm = cfg.max_matches
    out = {}
    for name, variant in VARIANTS:
        for order in (capi.ORDER_ATOMIC, capi.ORDER_RASTER):
            p = capi.frame_params(cfg, sat_variant=variant, order=order)
            d = dict(flat=torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
                     matches=torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"), idx=torch.zeros((B, mm), dtype=torch.int32, device="cuda"),
                     counter=torch.zeros((B, 3), dtype=torch.int32, device="cuda"), graddot=torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.float32, device="cuda"))
            ctx.frame(p, d_raw, B, **d)
            ctx.sync()
            out[(name, order)] = {k: host(v) for k, v in d.items()}
    ref = out[(VARIANTS[0][0], capi.ORDER_RASTER)]
    # both SAT variants and both list orders agree on every integer output and on the blob set
    for key, o in out.items():
        assert np.array_equal(o["flat"], ref["flat"]) and np.array_equal(o["sat"], ref["sat"]) and np.array_equal(o["counter"], ref["counter"]), key
        for i in range(B):
            a = capi.sorted_blobs(o["matches"][i], o["idx"][i].view(np.uint32), o["counter"][i], mm)
            b = capi.sorted_blobs(ref["matches"][i], ref["idx"][i].view(np.uint32), ref["counter"][i], mm)
            assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]), key
    # SAT == 2-D cumulative sum of gradDot in int64 arithmetic (mod 2^32); exactness guard of the benchmark inputs
    for i in range(B):
        gd = ref["graddot"][i].astype(np.int64)
        full = gd.cumsum(0).cumsum(1)
        assert np.abs(full).max() < 2 ** 24, "benchmark inputs must satisfy the 2^24 guard (SURVEY H2)"
        assert np.array_equal(ref["sat"][i], full.astype(np.int32))
    # frame 0 against the CPU oracle end to end (a few seconds)
    model_o = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    big = O.Port(threads=8)
    m, idx, cnt = big.frame(raws[0], cfg.layout, cfg.qw, cfg.qh, model_o, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                            cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm)
    assert 50 < len(m) <= mm
    assert_list_equal(capi, (ref["matches"][0], ref["idx"][0].view(np.uint32), ref["counter"][0]), np.frombuffer(m.tobytes(), np.uint8), idx, cnt, mm)
    # the rendered robots are found: 16 robots x 5 blobs + 8 balls, each within 1.5 px of a detected blob of high circularity
    gm, _ = capi.sorted_blobs(ref["matches"][0], ref["idx"][0].view(np.uint32), ref["counter"][0], mm)
    strong = gm[gm["circ"] > 500]
    assert 60 <= len(strong) <= 200


def test_cfg5_high_res_frame_vs_oracle(gpu):
    """BASELINE.json configs[4] (4096x3072, o = 3, r = 9, R = 8): one frame end to end against the CPU oracle, LDS-tiled and
    global-recursive SAT variants (the o = 3 instantiation of the tiled kernel aliases its sub-segment totals over the raw tile)."""
    capi, ctx, torch = gpu
    cfg = cfgs.CFG5
    assert (cfg.grad_offset, cfg.circle_radius, cfg.list_radius) == (3, 9, 8)
    raw = cfg.synth(11)
    mm = cfg.max_matches
    d_raw = dev(torch, raw[None])
    model_o = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    big = O.Port(threads=8)
    m, idx, cnt = big.frame(raw, cfg.layout, cfg.qw, cfg.qh, model_o, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                            cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm)
    assert 50 < len(m) <= mm
    sats = []
    for name, variant in VARIANTS:
        p = capi.frame_params(cfg, sat_variant=variant, order=capi.ORDER_RASTER)
        d = dict(flat=torch.zeros((1, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.zeros((1, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
                 matches=torch.zeros((1, mm * 22), dtype=torch.uint8, device="cuda"), idx=torch.zeros((1, mm), dtype=torch.int32, device="cuda"),
                 counter=torch.zeros((1, 3), dtype=torch.int32, device="cuda"))
        ctx.frame(p, d_raw, 1, **d)
        ctx.sync()
        assert_list_equal(capi, (host(d["matches"][0]), host(d["idx"][0]).view(np.uint32), host(d["counter"][0])), np.frombuffer(m.tobytes(), np.uint8), idx, cnt, mm)
        sats.append(host(d["sat"][0]))
    assert all(np.array_equal(sats[0], s) for s in sats[1:])


@pytest.mark.parametrize("cfg", [cfgs.T_MID, cfgs.T_RGGB], ids=lambda c: c.name)
@pytest.mark.parametrize("o", [1, 2, 3, 4])
def test_halo_strips_path(gpu, port, cfg, o):
    """Single-pass SAT fed by compact halo strips (built standalone here, emitted by the resampling kernels inside vp_frame)."""
    capi, ctx, torch = gpu
    rng = np.random.default_rng(o)
    B, h, w = 2, cfg.fh, cfg.fw
    flat = rng.integers(0, 256, (B, h, w, 4), dtype=np.uint8)
    flat[..., 3] = 255
    d_flat = dev(torch, flat)
    nbytes = ctx.halo_strip_bytes(w, h, o)
    assert nbytes == 2 * (w // 64) * h * o * 4
    d_strips = torch.full((B, nbytes // 4), 0x55AA55AA, dtype=torch.int32, device="cuda")
    ctx.halo_strips(d_flat, w, h, o, B, d_strips)
    d_sat = torch.zeros((B, h, w), dtype=torch.int32, device="cuda")
    d_gd = torch.zeros((B, h, w), dtype=torch.float32, device="cuda")
    ctx.graddot_sat_strips(d_flat, d_strips, w, h, o, B, capi.SAT_TILED, d_sat, d_gd)
    ctx.sync()
    for i in range(B):
        g = port.graddot(flat[i], o)
        assert np.array_equal(host(d_gd[i]), g)
        assert np.array_equal(host(d_sat[i]), g.astype(np.int64).cumsum(0).cumsum(1).astype(np.int32))
    assert ctx.halo_strip_bytes(w + 1, h, o) == 0 and ctx.halo_strip_bytes(w, h, 5) == 0


def test_abi_memory_and_events(gpu):
    """The plain-pointer half of the ABI (no torch): vp_malloc / copies / memset / stage timers."""
    capi, _, torch = gpu
    ctx = capi.Context(0, n_streams=2)
    try:
        assert "gfx950" in ctx.device_name
        g = cases.load_golden("flat_borders")
        h, w = g["flat"].shape[:2]
        o, r, R, mm = [int(v) for v in g["ints"]]
        d_flat, d_sat, d_bc = ctx.malloc(w * h * 4), ctx.malloc(w * h * 4), ctx.malloc(w * h * 4)
        d_m, d_idx, d_cnt = ctx.malloc(mm * 22), ctx.malloc(mm * 4), ctx.malloc(12)
        ctx.h2d(d_flat, g["flat"], stream=1)
        ctx.events_enable(True)
        ctx.graddot_sat(d_flat, w, h, o, 1, capi.SAT_TWOPASS, d_sat, None, stream=1)
        ctx.blobcenter_list(d_flat, d_sat, w, h, 1, r, float(g["scalars"][0]), float(g["scalars"][1]), R, mm, capi.ORDER_RASTER, d_m, d_idx, d_cnt, d_bc, stream=1)
        times = ctx.event_times()
        assert [t[0] for t in times] == ["gradientDot+sat", "satBlobCenter+blobList"] and all(t[1] > 0 for t in times)
        ctx.events_clear()
        assert ctx.event_times() == []
        assert np.array_equal(ctx.d2h(d_sat, (h, w), np.int32, stream=1), g["sat"])
        assert np.array_equal(ctx.d2h(d_bc, (h, w), np.float32, stream=1), g["blobcenter"])
        cnt = ctx.d2h(d_cnt, (3,), np.int32, stream=1)
        assert np.array_equal(cnt, g["counter"])
        n = int(cnt[0])
        assert ctx.d2h(d_m, (n * 22,), np.uint8, stream=1).tobytes() == g["matches"].tobytes()
        ctx.memset(d_cnt, 0, 12, stream=1)
        assert not ctx.d2h(d_cnt, (3,), np.int32, stream=1).any()
        with pytest.raises(capi.VpError):
            ctx.graddot_sat(d_flat, w, h, o, 1, 99, d_sat, None)
        for p in (d_flat, d_sat, d_bc, d_m, d_idx, d_cnt):
            ctx.free(p)
    finally:
        ctx.close()


@pytest.mark.parametrize("mode", [0, 1], ids=["nearest", "bilinear_rtz"])
def test_config0_640x480_bgr_frame(gpu, mode):
    """BASELINE config 0 at its own size (one 640x480 BGR frame): the HIP path against the fixture generated from the reference's own
    compiled kernels -- digests of flat and SAT, counters, pixel indices and blob records."""
    import hashlib
    capi, ctx, torch = gpu
    g = cases.load_golden("config0_640x480_bgr")
    cfg = cfgs.CFG1
    assert (int(g["qw"]), int(g["qh"]), int(g["fw"]), int(g["fh"]), int(g["layout"])) == (cfg.qw, cfg.qh, cfg.fw, cfg.fh, cfg.layout)
    assert np.array_equal(np.frombuffer(bytes(cfg.model()), np.uint8), g["model"].view(np.uint8).reshape(-1))
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
    mm = cfg.max_matches
    for variant, use_map in ((capi.SAT_TILED, True), (capi.SAT_TWOPASS, False)):
        p = capi.frame_params(cfg, sat_variant=variant, order=capi.ORDER_RASTER, sampler_mode=mode, use_coord_map=use_map)
        d = dict(flat=torch.zeros((cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.zeros((cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
                 matches=torch.zeros(mm * 22, dtype=torch.uint8, device="cuda"), idx=torch.zeros(mm, dtype=torch.int32, device="cuda"),
                 counter=torch.zeros(3, dtype=torch.int32, device="cuda"))
        views = dict(planes=torch.zeros((4, cfg.qh, cfg.qw), dtype=torch.uint8, device="cuda"), graddot=torch.zeros((cfg.fh, cfg.fw), dtype=torch.float32, device="cuda"),
                     blobcenter=torch.zeros((cfg.fh, cfg.fw), dtype=torch.float32, device="cuda"))
        ctx.frame(p, dev(torch, g["raw"]), 1, **d, **views)
        ctx.sync()
        assert np.array_equal(sha(host(views["planes"])[:3]), g["sha_planes"])
        assert np.array_equal(sha(host(d["flat"])), g[f"m{mode}_sha_flat"])
        assert np.array_equal(sha(host(views["graddot"]).astype(np.int32)), g[f"m{mode}_sha_graddot"])
        assert np.array_equal(sha(host(d["sat"])), g[f"m{mode}_sha_sat"])
        assert np.array_equal(sha(host(views["blobcenter"])), g[f"m{mode}_sha_blobcenter"])
        assert_list_equal(capi, (host(d["matches"]), host(d["idx"]).view(np.uint32), host(d["counter"])), g[f"m{mode}_matches"], g[f"m{mode}_idx"], g[f"m{mode}_counter"], mm)
