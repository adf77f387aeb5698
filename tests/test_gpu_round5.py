"""GPU tests added in round 5, all through the C ABI.

  * vp_stream_wait_for / vp_stream_release_to: the device-side edges between a context's own (non-blocking) streams and streams the library does not own
    -- the cl::Event of OpenCL::run (opencl.h:76-86) as a dependency instead of a host wait.
  * a refused graph capture is bounded (eight bodies in a row, then the tuple stays on the ordinary path) and leaves no stale vp_last_error behind.
  * the placement search keeps to its footprint cap and says when it ran out of budget.
"""
import ctypes as C

import numpy as np
import pytest

import cases
import oracle as O

pytestmark = pytest.mark.gpu
cfgs = cases.cfgs


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return cases.vp.load_capi(), torch


def oracle_frame(port, cfg, raw, mm):
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    return port.frame(raw, cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                      cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm)


def test_stream_edges_order_a_library_stream_against_foreign_streams(env, port):
    """A producer stream the library does not own is kept busy for tens of milliseconds and only THEN copies the sensor frame into the buffer vp_frame
    reads; a consumer stream reads the results.  With the FENCE_TORCH helper of the tests switched off, the only things ordering the three streams are
    vp_stream_wait_for (producer -> library stream) and vp_stream_release_to (library stream -> consumer): no host synchronisation anywhere between
    the first enqueue and the final wait.  Without the first edge vp_frame would run on the zero-filled buffer, without the second the consumer would
    copy unfinished arrays; the results must be the oracle's.  Repeated with a fresh frame: the edges are per call, not a one-off."""
    capi, torch = env
    cfg, mm = cfgs.T_MID, cfgs.T_MID.max_matches
    capi.FENCE_TORCH = False  # this test orders its streams itself (the autouse fixture restores nothing it did not set: it resets the flag afterwards)
    ctx = capi.Context(0, n_streams=2)
    try:
        prod, cons = torch.cuda.Stream(), torch.cuda.Stream()
        p = capi.frame_params(cfg, sat_variant=capi.SAT_TILED, order=capi.ORDER_RASTER)
        d_raw = torch.zeros(cfg.sensor_shape, dtype=torch.uint8, device="cuda")
        flat = torch.zeros((cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
        sat = torch.zeros((cfg.fh, cfg.fw), dtype=torch.int32, device="cuda")
        m = torch.zeros(mm * 22, dtype=torch.uint8, device="cuda")
        idx = torch.zeros(mm, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(3, dtype=torch.int32, device="cuda")
        busy = torch.randn(4096, 4096, device="cuda")
        torch.cuda.synchronize()
        for rnd, stream in ((0, 1), (1, 0), (2, 1)):
            raw = cfg.synth(7700 + rnd)
            h_raw = torch.from_numpy(raw).pin_memory()
            with torch.cuda.stream(prod):
                acc = busy
                for _ in range(24):  # ~ tens of ms of matrix products ahead of the copy
                    acc = acc @ busy * 1e-3
                d_raw.copy_(h_raw, non_blocking=True)
            ctx._ck(ctx.l.vp_stream_wait_for(ctx.h, stream, C.c_void_p(prod.cuda_stream)))
            ctx.frame(p, d_raw, 1, flat, sat, m, cnt, idx=idx, stream=stream)
            ctx._ck(ctx.l.vp_stream_release_to(ctx.h, stream, C.c_void_p(cons.cuda_stream)))
            with torch.cuda.stream(cons):
                got = [t.to("cpu", non_blocking=False) for t in (cnt, idx, m)]
            # the producer may not overwrite d_raw for the next round before this round's vp_frame has read it: the reverse edge
            ctx._ck(ctx.l.vp_stream_release_to(ctx.h, stream, C.c_void_p(prod.cuda_stream)))
            wm, widx, wcnt = oracle_frame(port, cfg, raw, mm)
            assert np.array_equal(got[0].numpy(), wcnt), (rnd, got[0].numpy(), wcnt)
            n = min(int(wcnt[0]), mm)
            gm, gi = capi.sorted_blobs(got[2].numpy(), got[1].numpy().view(np.uint32), got[0].numpy(), mm)
            assert n >= 20 and np.array_equal(gi, widx) and gm.tobytes() == wm.tobytes(), rnd
            del acc
        # same stream on both sides (an adopted stream is ordered with itself): a no-op, not an error; bad stream indices are refused
        assert ctx.l.vp_stream_wait_for(ctx.h, 5, C.c_void_p(prod.cuda_stream)) != 0
        ctx.sync()
    finally:
        ctx.close()
    a = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
    try:
        a._ck(a.l.vp_stream_wait_for(a.h, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        a._ck(a.l.vp_stream_release_to(a.h, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    finally:
        a.close()


def test_the_test_fence_is_a_device_side_edge(env):
    """capi.FENCE_TORCH (what the autouse fixture of tests/conftest.py switches on) must order torch's stream before a call on a library stream without
    a host wait: a 0xAB fill queued behind a long kernel on torch's stream, handed to vp_memset_device on a library stream at once, must lose to it."""
    capi, torch = env
    assert capi.FENCE_TORCH is True
    ctx = capi.Context(0, n_streams=1)
    try:
        busy = torch.randn(4096, 4096, device="cuda")
        for _ in range(3):
            acc = busy
            for _ in range(16):
                acc = acc @ busy * 1e-3
            t = torch.full((1 << 20,), 0xAB, dtype=torch.uint8, device="cuda")  # queued behind the products on torch's stream
            ctx.memset(t, 0x11, t.numel())                                      # library stream: must come after the fill
            ctx.sync()
            torch.cuda.synchronize()
            assert int(t.min()) == 0x11 and int(t.max()) == 0x11
            del acc
        assert not capi._pending
    finally:
        ctx.close()


def test_a_capture_whose_body_keeps_failing_stops_being_attempted(env, port):
    """frame_graph = 2 (no pre-flight), two launch shapes taking turns on ONE stream: every second sighting of a tuple follows the OTHER shape, so every
    capture meets a SAT-state reset and is refused and undone.  Round 4 ran the launch sequence twice on every such call for ever; now eight refused
    bodies in a row put the tuple on the ordinary path for good (2 x 8 refusals in total, however long the alternation lasts), the ordinary retry that
    follows a refusal leaves vp_last_error empty, and every result is still the oracle's."""
    capi, torch = env
    a = cfgs.T_MID
    b = a.with_(name=a.name + "_704x448", fw=704, fh=448)
    ctx = capi.Context(0, n_streams=1)
    try:
        ctx.tune("frame_graph", 2)
        st = {}
        for k, (c, B) in {"A": (a, 2), "B": (b, 3)}.items():
            mm = c.max_matches
            raws = [c.synth(8100 + 7 * j + (k == "B")) for j in range(B)]
            st[k] = dict(cfg=c, B=B, raws=raws, raw=torch.from_numpy(np.stack(raws)).cuda(), flat=torch.zeros((B, c.fh, c.fw, 4), dtype=torch.uint8, device="cuda"),
                         sat=torch.zeros((B, c.fh, c.fw), dtype=torch.int32, device="cuda"), m=torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"),
                         idx=torch.zeros((B, mm), dtype=torch.int32, device="cuda"), cnt=torch.zeros((B, 3), dtype=torch.int32, device="cuda"),
                         p=capi.frame_params(c, sat_variant=capi.SAT_TILED, order=capi.ORDER_RASTER))
            st[k]["want"] = [oracle_frame(port, c, r, mm) for r in raws]
        torch.cuda.synchronize()
        refused_after = []
        for rnd in range(24):
            s = st["AB"[rnd & 1]]
            s["cnt"].fill_(-7)
            ctx.frame(s["p"], s["raw"], s["B"], s["flat"], s["sat"], s["m"], s["cnt"], idx=s["idx"])
            refused_after.append(ctx.graph_stats()["refused"])
            if refused_after[-1] > (refused_after[-2] if rnd else 0):  # this call's capture was refused and the ordinary retry made it good:
                assert ctx.l.vp_last_error() == b"", (rnd, ctx.l.vp_last_error())  # the refusal's message is not this call's outcome
            ctx.sync()
            cnt, mm = s["cnt"].cpu().numpy(), s["cfg"].max_matches
            for j in range(s["B"]):
                wm, widx, wcnt = s["want"][j]
                assert np.array_equal(cnt[j], wcnt), (rnd, j, cnt[j], wcnt)
                n = min(int(wcnt[0]), mm)
                assert np.array_equal(s["idx"][j].cpu().numpy().view(np.uint32)[:n], widx) and s["m"][j].cpu().numpy()[:22 * n].tobytes() == wm.tobytes(), (rnd, j)
        stats = ctx.graph_stats()
        assert stats["captures"] == 0 and stats["replays"] == 0, stats
        assert stats["refused"] == 16 and refused_after[-1] == refused_after[-5], (stats, refused_after)  # 8 per tuple, none during the last rounds
    finally:
        ctx.close()


def test_placement_search_keeps_to_its_cap_and_reports_an_exhausted_budget(env):
    """vp_malloc_device_placed: with "placement_max_mib" the search holds no more than that at once (candidates <= cap / block size); a search that ends
    without a fast block reports selected = 2 (bench.py and vp_feeds print that as a warning), one that needs no search reports 0.  The default cap is a
    quarter of the device: asked for 1 GiB blocks it can never probe more than total / 4 GiB of them."""
    capi, torch = env
    ctx = capi.Context(0, n_streams=1)
    lib = ctx.l
    try:
        nbytes = 1 << 30
        free0, total = torch.cuda.mem_get_info()

        def placed(count):
            ptrs = (C.c_void_p * count)()
            info = (capi.VpPlacement * count)()
            ctx._ck(lib.vp_malloc_device_placed(ctx.h, C.c_size_t(nbytes), count, ptrs, info))
            out = [info[i].as_dict() | {"raw_selected": int(info[i].selected)} for i in range(count)]
            for q in ptrs:
                ctx.free(q)
            return out

        ctx.tune("placement_max_mib", 3 * 1024)  # room for three 1 GiB candidates
        ctx.tune("placement_candidates", 64)
        r = placed(1)[0]
        assert 1 <= r["candidates"] <= 3, r
        assert r["raw_selected"] in (1, 2) and (("warning" in r) == (r["raw_selected"] == 2)), r
        ctx.tune("placement_max_mib", 0)
        r = placed(2)
        assert all(2 <= x["candidates"] <= max(2, total // 4 // nbytes) for x in r), (r, total)
        ctx.tune("placement_max_mib", -1)  # "the GPU is mine": nine tenths of the free memory, 1 GiB / 2 % reserve (bench.py, vp_feeds)
        r = placed(1)[0]
        assert r["raw_selected"] in (1, 2) and 1 <= r["candidates"] <= 64, r
        ctx.tune("placement_max_mib", 0)
        ctx.tune("placement_candidates", 0)  # take what hipMalloc gives
        r = placed(1)[0]
        assert r["raw_selected"] == 0 and r["candidates"] == 1 and "warning" not in r, r
        # frame slabs: ONE fast block (the table's) ends the search, but two blocks are handed out whenever it ends -- also when the very first candidate is fast
        # (that case once returned a slab whose flat was NULL: bench.py under rocprofv3 --pmc tripped over it)
        ctx.tune("placement_candidates", 96)
        ctx.tune("placement_max_mib", -1)
        cfg = cfgs.CFG2
        p = capi.frame_params(cfg)
        for _ in range(12):
            slab = ctx.frame_slab(p, 64)  # 805 MB blocks: searched
            try:
                assert slab.b.flat and slab.b.sat and slab.b.flat != slab.b.sat, (slab.b.flat, slab.b.sat)
                ps, pf = slab.placement["sat"], slab.placement["flat"]
                assert ps["candidates"] >= 2 and ps["write_gbs"] >= pf["write_gbs"] > 1000 and ps["fast_gbs"] > 1000, slab.placement
            finally:
                slab.destroy()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 >= free0 - (64 << 20), (free0, free1)  # every candidate went back
    finally:
        ctx.close()


@pytest.mark.parametrize("shift", [0.0, -0.45, 0.45], ids=["centred", "past_the_left_border", "past_the_right_border"])
def test_resampling_frame_pairs_and_groups_per_workgroup(env, port, shift):
    """k_resample_staged<., 3> (the default): the footprint box staged as floats of two frames side by side, both frames computed at once, a workgroup
    taking several 16-frame groups with one set-up.  101 frames = 64 (one workgroup's full share: the written-out stream) + groups of 16 + 16 + 5 (an odd
    last group: a half-filled last pair), every
    "resample_groups_per_wg" setting and the older stage formats must give the same flat images, and frames on both sides of every group boundary must be
    the oracle's.  `shift` moves the field window so that the left / right part of flat looks past the sensor border: there a footprint's two columns
    clamp onto one cell (resampling.cl:50 with CLK_ADDRESS_CLAMP_TO_EDGE), which the frame-pair body reads from a copy of the border cell in a virtual
    cell beside the box; the flat window also ends inside a tile there (clamped duplicate lanes)."""
    capi, torch = env
    base = cfgs.T_MID
    cfg = base.with_(name=f"{base.name}_shift{shift}", off_x=base.off_x + shift * base.fw * base.field_scale) if shift else base
    # 101 frames: with the default four groups per workgroup the first 64 take the written-out, software-prefetched pair stream, the other 37 the loop
    B, mm = 101, cfg.max_matches
    raws = np.stack([cfg.synth(6200 + i) for i in range(B)])
    d_raw = torch.from_numpy(raws).cuda()
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    want = {}
    for k in (0, 1, 3, 4, 15, 16, 31, 32, 62, 63, 64, 79, 80, 95, 96, 99, 100):
        s = port.frame_scratch(cfg.qw, cfg.qh, cfg.fw, cfg.fh, mm)
        port.frame(raws[k], cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                   cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm, scratch=s)
        want[k] = s["flat"].copy()
    if shift:  # the window really looks past the border: the outermost flat columns (nearly) repeat -- every footprint there is the border cell, only v still moves
        col = 0 if shift < 0 else cfg.fw - 1
        step = 1 if shift < 0 else -1
        same = np.mean(np.all(want[0][:, col] == want[0][:, col + 8 * step], axis=-1))
        inner = np.mean(np.all(want[0][:, cfg.fw // 2] == want[0][:, cfg.fw // 2 + 8], axis=-1))
        assert same > inner + 0.3, (same, inner)
    ref = None
    for knobs in ({}, {"resample_groups_per_wg": 1}, {"resample_groups_per_wg": 3}, {"resample_groups_per_wg": 64}, {"resample_float_stage": 1}, {"resample_float_stage": 0},
                  {"sat_halo_strips": 0, "resample_groups_per_wg": 2}):
        ctx = capi.Context(0, n_streams=1)
        try:
            for k, v in knobs.items():
                ctx.tune(k, v)
            p = capi.frame_params(cfg, sat_variant=capi.SAT_TILED, order=capi.ORDER_RASTER)
            flat = torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
            sat = torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda")
            m = torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda")
            idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda")
            cnt = torch.full((B, 3), -3, dtype=torch.int32, device="cuda")
            for _ in range(2):
                ctx.frame(p, d_raw, B, flat, sat, m, cnt, idx=idx)
            ctx.sync()
            got = (flat.cpu().numpy(), sat.cpu().numpy(), cnt.cpu().numpy())
        finally:
            ctx.close()
        for k, w in want.items():
            assert np.array_equal(got[0][k], w), (knobs, "flat of frame", k)
        if ref is None:
            ref = got
        else:  # the table (built from the strip records the resampling kernel wrote) and the counters too
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), knobs


def test_resampling_groups_per_workgroup_on_the_unstaged_path(env, port):
    """A geometry whose 64 x 4-pixel tiles have footprint boxes too large for the LDS stage (one flat pixel spans four sensor cells: a box of ~130 x 20
    dwords) sends every workgroup of the default resampling kernel down its per-pixel path -- which must then cover ALL the 16-frame groups the workgroup
    was given (four by default: resample_groups_per_wg 0 = automatic), not only the first: frames 16..36 of a 37-frame launch against the oracle."""
    capi, torch = env
    base = cfgs.T_MID
    fs = base.field_scale * 8.0
    cfg = base.with_(name=base.name + "_coarse", field_scale=fs, off_x=-base.fw / 2 * fs, off_y=-base.fh / 2 * fs)
    B, mm = 37, cfg.max_matches
    raws = np.stack([cfg.synth(6400 + i) for i in range(B)])
    d_raw = torch.from_numpy(raws).cuda()
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    ctx = capi.Context(0, n_streams=1)
    try:
        flat = torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
        for gpw in (0, 1, 3):
            ctx.tune("resample_groups_per_wg", gpw)
            flat.zero_()
            torch.cuda.synchronize()
            ctx.resample_map(cfg.model(), cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh, m := torch.zeros((cfg.fh, cfg.fw, 2), dtype=torch.float32, device="cuda"))
            ctx.resample_raw_mapped(d_raw, cfg.layout, cfg.qw, cfg.qh, m, cfg.sampler_mode, cfg.fw, cfg.fh, B, flat)
            ctx.sync()
            got = flat.cpu().numpy()
            for k in (0, 15, 16, 31, 32, 36):
                s = port.frame_scratch(cfg.qw, cfg.qh, cfg.fw, cfg.fh, mm)
                port.frame(raws[k], cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                           cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm, scratch=s)
                assert np.array_equal(got[k], s["flat"]), (gpw, k)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fh", [3072, 4160], ids=["span_3_2_gb_one_descriptor", "span_4_4_gb_loop_form"])
def test_resampling_workgroup_whose_frames_span_gigabytes_of_flat(env, port, fh):
    """The written-out frame-pair stream addresses a workgroup's 64 frames through ONE buffer descriptor and scalar frame offsets, which must stay inside 32
    bits.  Flat frames of 50 MB (4096 x 3072, configs[4]'s size): 64 of them span 3.2 GB -- a descriptor whose size has bit 31 set and offsets past 2 GiB.
    Flat frames of 68 MB (4096 x 4160): 4.36 GB, so the kernel has to take its loop form (a descriptor per sub-group) instead -- silently, with the same
    result.  66 frames (a workgroup's full 64 + 2): every frame against the byte-staged body of the same launch geometry (other descriptors, another
    arithmetic path to the same integers), frames on both sides of the 64-frame boundary against the oracle's raw2quad + resampling."""
    capi, torch = env
    base = cfgs.T_MID
    fw = 4096
    fs = base.field_scale * base.fw / fw
    cfg = base.with_(name=base.name + "_huge_flat", fw=fw, fh=fh, field_scale=fs)
    assert (64 * fw * fh * 4 > 0xffffffff) == (fh == 4160) and 64 * fw * fh * 4 > 0x7fffffff and 16 * fw * fh * 4 < 0x7fffffff
    B = 66
    raws = np.stack([cfg.synth(6600 + i) for i in range(B)])
    d_raw = torch.from_numpy(raws).cuda()
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    ctx = capi.Context(0, n_streams=1)
    try:
        m = torch.zeros((fh, fw, 2), dtype=torch.float32, device="cuda")
        ctx.resample_map(cfg.model(), cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, fw, fh, m)
        flat = torch.zeros((B, fh, fw, 4), dtype=torch.uint8, device="cuda")
        ctx.resample_raw_mapped(d_raw, cfg.layout, cfg.qw, cfg.qh, m, cfg.sampler_mode, fw, fh, B, flat)
        ctx.sync()
        for k in (0, 63, 64, 65):
            planes = port.raw2quad(raws[k], cfg.layout, cfg.qw, cfg.qh)
            want = port.resample(planes, cfg.layout, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode, fw, fh)
            assert torch.equal(flat[k].cpu(), torch.from_numpy(want)), k
        other = torch.zeros_like(flat)
        ctx.tune("resample_float_stage", 0)
        ctx.resample_raw_mapped(d_raw, cfg.layout, cfg.qw, cfg.qh, m, cfg.sampler_mode, fw, fh, B, other)
        ctx.sync()
        for k in range(B):  # (frame by frame: torch.equal on 4.5 GB would make a 4.5 GB temporary)
            assert torch.equal(flat[k], other[k]), k
    finally:
        ctx.close()


@pytest.mark.gpu
def test_resampling_full_stream_on_the_other_bayer_layout(env, port):
    """The written-out 32-pair stream is a separate instantiation per Bayer layout (which sum is red, which two are the halved greens): GRBG with a
    workgroup's full 64 frames + 6, frames on both sides of the boundary against the oracle, every frame against the byte-staged body."""
    capi, torch = env
    cfg = cfgs.T_MID.with_(name=cfgs.T_MID.name + "_grbg", layout=cfgs.GRBG)
    B = 70
    raws = np.stack([cfg.synth(6700 + i) for i in range(B)])
    d_raw = torch.from_numpy(raws).cuda()
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    ctx = capi.Context(0, n_streams=1)
    try:
        m = torch.zeros((cfg.fh, cfg.fw, 2), dtype=torch.float32, device="cuda")
        ctx.resample_map(cfg.model(), cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh, m)
        flat = torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
        ctx.resample_raw_mapped(d_raw, cfg.layout, cfg.qw, cfg.qh, m, cfg.sampler_mode, cfg.fw, cfg.fh, B, flat)
        ctx.sync()
        for k in (0, 1, 31, 62, 63, 64, 69):
            planes = port.raw2quad(raws[k], cfg.layout, cfg.qw, cfg.qh)
            want = port.resample(planes, cfg.layout, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode, cfg.fw, cfg.fh)
            assert torch.equal(flat[k].cpu(), torch.from_numpy(want)), k
        other = torch.zeros_like(flat)
        ctx.tune("resample_float_stage", 0)
        ctx.resample_raw_mapped(d_raw, cfg.layout, cfg.qw, cfg.qh, m, cfg.sampler_mode, cfg.fw, cfg.fh, B, other)
        ctx.sync()
        assert torch.equal(flat, other)
    finally:
        ctx.close()
