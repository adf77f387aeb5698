"""GPU parity tests added in round 3, all through the C ABI.

  * THE LAUNCH bench.py TIMES: cfg2 (2048x1536 RGGB) x 128 distinct frames in one vp_frame call with the bench's exact parameters
    (VP_SAT_TILED, RASTER order, coordinate map, the adopted torch stream, the feed seeds of feeds.py) -- frames 0, 15, 16, 17 and 127
    (both ends and both sides of a 16-frame resampling group) byte for byte against the CPU oracle: flat, the int32 SAT, counters,
    sorted records; ALL frames of the launch (256) through size-independent properties: the SAT equals the 64-bit 2-D cumulative sum of gradientDot
    formed from the kernel's own flat by plain torch integer arithmetic, flat equals what the unstaged per-pixel resampling kernel
    gives, and VP_SAT_LOCAL's records equal VP_SAT_TILED's.  Call sequence replaced: Resources.cpp:151-164 + main.cpp:283-317.
  * BASELINE configs[2] AT ITS OWN SIZE: the same 2048x1536 frames as 8 streams x 8 frames, every stream loaded before anything is
    waited for, three rounds: every stream's flat / SAT / counters / records equal the one-launch result byte for byte, and one frame
    per stream equals the oracle (one loop per camera: main.cpp:262-317).
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import cases
import oracle as O

pytestmark = pytest.mark.gpu
cfgs = cases.cfgs
CFG = cfgs.CFG2
B_BENCH = cases.vp.feeds.BENCH_FRAMES_PER_STEP  # bench.py --batch default (256 since round 5)
ORACLE_FRAMES = (0, 15, 16, 17, 127, 128, B_BENCH - 1)
NS, PER = 8, 8         # configs[2]: 8 streams x 8 frames in flight
STREAM_ORACLE = {s: s * PER + (3 * s) % PER for s in range(NS)}  # one oracle-checked frame per stream


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return cases.vp.load_capi(), torch


@pytest.fixture(scope="module")
def bench_frames():
    """The synthetic sensor frames (feeds.BENCH_FRAMES_PER_STEP of them) bench.py uploads on rank 0 (feed_seed(0, i)); rendered on host threads (ctypes releases the GIL)."""
    seeds = [cases.vp.feeds.feed_seed(0, i) for i in range(B_BENCH)]
    with ThreadPoolExecutor(16) as ex:
        return np.stack(list(ex.map(CFG.synth, seeds)))


@pytest.fixture(scope="module")
def oracle_of(bench_frames):
    """frame index -> the CPU oracle's flat, int32 SAT, counters and raster-sorted records for that frame (cached across the tests)."""
    big = O.Port(threads=min(16, os.cpu_count() or 1))
    model = O.CameraModel.from_array(np.frombuffer(bytes(CFG.model()), np.uint8))
    cache = {}

    def get(i):
        if i not in cache:
            mm = CFG.max_matches
            s = big.frame_scratch(CFG.qw, CFG.qh, CFG.fw, CFG.fh, mm)
            m, idx, cnt = big.frame(bench_frames[i], CFG.layout, CFG.qw, CFG.qh, model, CFG.max_robot_height, CFG.field_scale, CFG.off_x, CFG.off_y, CFG.sampler_mode,
                                    CFG.fw, CFG.fh, CFG.grad_offset, CFG.circle_radius, CFG.list_radius, CFG.circ_threshold, CFG.min_score, mm, scratch=s)
            sat, mx = big.sat_i32(s["graddot"])
            assert mx < 2 ** 24, "benchmark inputs must satisfy the 2^24 exactness guard (SURVEY H2)"
            assert 80 <= len(m) <= mm
            cache[i] = dict(flat=s["flat"].copy(), sat=sat, m=m, idx=idx, cnt=cnt)
        return cache[i]
    return get


def alloc(torch, n, with_sat=True):
    mm = CFG.max_matches
    return dict(flat=torch.zeros((n, CFG.fh, CFG.fw, 4), dtype=torch.uint8, device="cuda"),
                sat=torch.zeros((n, CFG.fh, CFG.fw), dtype=torch.int32, device="cuda") if with_sat else None,
                m=torch.zeros((n, mm * 22), dtype=torch.uint8, device="cuda"), idx=torch.zeros((n, mm), dtype=torch.int32, device="cuda"),
                cnt=torch.full((n, 3), -5, dtype=torch.int32, device="cuda"))


def assert_frame_is_the_oracles(capi, torch, bufs, k, want, what, with_sat=True):
    mm = CFG.max_matches
    assert torch.equal(bufs["flat"][k].cpu(), torch.from_numpy(want["flat"])), (what, "flat")
    if with_sat:
        assert torch.equal(bufs["sat"][k].cpu(), torch.from_numpy(want["sat"])), (what, "sat")
    cnt = bufs["cnt"][k].cpu().numpy()
    assert np.array_equal(cnt, want["cnt"]), (what, cnt, want["cnt"])
    n = min(int(cnt[0]), mm)
    assert np.array_equal(bufs["idx"][k].cpu().numpy().view(np.uint32)[:n], want["idx"]), (what, "RASTER order = sorted pixel indices")
    assert bufs["m"][k].cpu().numpy()[:22 * n].tobytes() == want["m"].tobytes(), (what, "records")


def torch_sat_reference(torch, flat, o):
    """gradientDot.cl:22-30 + satHorizontal.cl / satVertical.cl:22-32 on a (n, h, w, 4) uint8 device tensor in plain torch integer
    arithmetic: clamp-to-edge taps at distance o, sum over the three colour channels of gx * gy, 64-bit cumulative sums, wrapped to int32."""
    n, h, w, _ = flat.shape
    c = flat[..., :3].to(torch.int32)
    xs, ys = torch.arange(w, device=flat.device), torch.arange(h, device=flat.device)
    gx = c[:, :, (xs + o).clamp(max=w - 1)] - c[:, :, (xs - o).clamp(min=0)]
    gy = c[:, (ys + o).clamp(max=h - 1)] - c[:, (ys - o).clamp(min=0)]
    gd = (gx * gy).sum(-1).to(torch.int64)
    full = gd.cumsum(1).cumsum(2)
    assert int(full.abs().max()) < 2 ** 24
    return full.to(torch.int32)


def test_the_timed_launch_against_the_oracle(env, bench_frames, oracle_of):
    capi, torch = env
    B, mm = B_BENCH, CFG.max_matches
    ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)  # as bench.py: the adopted torch stream
    try:
        d_raw = dev(torch, bench_frames)
        tiled = alloc(torch, B)
        p = capi.frame_params(CFG, sat_variant=capi.SAT_TILED, order=capi.ORDER_RASTER)  # bench.py defaults (use_coord_map = 1)
        for _ in range(3):  # warm-up + steps: epochs, tickets, the cached map and the self-cleaning counters are reused as in the timed loop
            ctx.frame(p, d_raw, B, tiled["flat"], tiled["sat"], tiled["m"], tiled["cnt"], idx=tiled["idx"])
        ctx.sync()
        for k in ORACLE_FRAMES:
            assert_frame_is_the_oracles(capi, torch, tiled, k, oracle_of(k), ("tiled", k))
        # every frame: SAT == cumulative sums of gradientDot of the kernel's own flat; flat == the per-pixel (unstaged, unmapped) resampling kernel
        model = CFG.model()
        other = torch.empty((8, CFG.fh, CFG.fw, 4), dtype=torch.uint8, device="cuda")
        for k0 in range(0, B, 8):
            want = torch_sat_reference(torch, tiled["flat"][k0:k0 + 8], CFG.grad_offset)
            assert torch.equal(tiled["sat"][k0:k0 + 8], want), ("SAT property", k0)
            ctx.resample_raw(d_raw[k0:k0 + 8], CFG.layout, CFG.qw, CFG.qh, model, CFG.max_robot_height, CFG.field_scale, CFG.off_x, CFG.off_y, CFG.sampler_mode, CFG.fw, CFG.fh, 8, other)
            ctx.sync()
            assert torch.equal(tiled["flat"][k0:k0 + 8], other), ("flat: staged+mapped vs per-pixel kernel", k0)
        cnt = tiled["cnt"].cpu().numpy()
        assert int(cnt[:, 0].min()) >= 80 and int(cnt[:, 0].max()) <= mm
        # the same launch without a global table
        local = alloc(torch, B, with_sat=False)
        local["flat"] = tiled["flat"]  # resampling rewrites it with the same bytes
        pl = capi.frame_params(CFG, sat_variant=capi.SAT_LOCAL, order=capi.ORDER_RASTER)
        for _ in range(2):
            ctx.frame(pl, d_raw, B, local["flat"], None, local["m"], local["cnt"], idx=local["idx"])
        ctx.sync()
        for k in ORACLE_FRAMES:
            assert_frame_is_the_oracles(capi, torch, local, k, oracle_of(k), ("local", k), with_sat=False)
        assert torch.equal(local["cnt"], tiled["cnt"])
        lm, li, tm, ti = local["m"].cpu().numpy(), local["idx"].cpu().numpy(), tiled["m"].cpu().numpy(), tiled["idx"].cpu().numpy()
        for k in range(B):
            n = min(int(cnt[k, 0]), mm)
            assert np.array_equal(li[k, :n], ti[k, :n]) and lm[k, :22 * n].tobytes() == tm[k, :22 * n].tobytes(), ("LOCAL records == TILED records", k)
    finally:
        ctx.close()


@pytest.mark.parametrize("variant", ["tiled", "local"])
def test_config2_eight_streams_by_eight_frames_at_2048x1536(env, bench_frames, oracle_of, variant):
    capi, torch = env
    mm = CFG.max_matches
    with_sat = variant == "tiled"
    p = capi.frame_params(CFG, sat_variant=capi.SAT_TILED if with_sat else capi.SAT_LOCAL, order=capi.ORDER_RASTER)
    d_raw = dev(torch, bench_frames[:NS * PER])
    ctx = capi.Context(0, n_streams=NS)
    try:
        # one launch of all 64 frames on one stream: what the streams' results are held to
        one = alloc(torch, NS * PER, with_sat)
        torch.cuda.synchronize()
        ctx.frame(p, d_raw, NS * PER, one["flat"], one["sat"], one["m"], one["cnt"], idx=one["idx"], stream=0)
        ctx.sync()
        bufs = [alloc(torch, PER, with_sat) for _ in range(NS)]
        torch.cuda.synchronize()
        for rnd in range(3):
            for s in range(NS):  # all eight streams are loaded before any of them is waited for
                b = bufs[s]
                ctx.frame(p, d_raw[s * PER:(s + 1) * PER], PER, b["flat"], b["sat"], b["m"], b["cnt"], idx=b["idx"], stream=s)
        ctx.sync()
        cnt1 = one["cnt"].cpu().numpy()
        for s in range(NS):
            b, sl = bufs[s], slice(s * PER, (s + 1) * PER)
            assert torch.equal(b["flat"], one["flat"][sl]), (s, "flat")
            if with_sat:
                assert torch.equal(b["sat"], one["sat"][sl]), (s, "sat")
            assert torch.equal(b["cnt"], one["cnt"][sl]), (s, "counters")
            bm, bi, om, oi = b["m"].cpu().numpy(), b["idx"].cpu().numpy(), one["m"][sl].cpu().numpy(), one["idx"][sl].cpu().numpy()
            for k in range(PER):
                n = min(int(cnt1[s * PER + k, 0]), mm)
                assert np.array_equal(bi[k, :n], oi[k, :n]) and bm[k, :22 * n].tobytes() == om[k, :22 * n].tobytes(), (s, k, "records")
            g = STREAM_ORACLE[s]
            assert_frame_is_the_oracles(capi, torch, b, g - s * PER, oracle_of(g), (variant, "stream", s, "frame", g), with_sat=with_sat)
    finally:
        ctx.close()


def test_dpp_primitives_on_the_device():
    """tools/dbg/dpp_test (built by `make dbg`): the cross-lane primitives the walkers are made of, on the real device.  Rows 0-7 (wave / row
    shifts, ds_bpermute) must behave as documented.  Rows 8-10 are subtractions with a DPP-shifted operand: `shr(e) - a` (shifted value as
    minuend, the only form the kernels use) must be right; `a - shl(e)` is where ROCm 7.2 / gfx950 was found to fold the shift into
    v_subrev_u32_dpp and shift the wrong operand -- reported, not required, so that the workaround is re-examined when the toolchain changes."""
    exe = os.path.join(cases.ROOT, "tools", "dbg", "dpp_test")
    if not os.path.exists(exe):
        pytest.skip("tools/dbg/dpp_test not built (make dbg)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")))
    assert out.returncode == 0, out.stderr
    rows = {ln.rsplit(":", 1)[0].strip(): [int(v) for v in ln.rsplit(":", 1)[1].split()] for ln in out.stdout.strip().splitlines()}
    e = [100 + i for i in range(64)]
    a = [10000 + 100 * i for i in range(64)]
    assert rows["wave_shl:1"][:63] == e[1:] and rows["wave_shr:1"][1:] == e[:63]
    assert rows["bperm +5 (folded)"] == [e[(i + 5) % 64] for i in range(64)] and rows["bperm -5"] == [e[(i - 5) % 64] for i in range(64)]
    assert rows["bperm +5 (opaque)"] == [e[(i + 5) % 64] for i in range(64)]
    assert rows["wave_rol:1"] == [e[(i + 1) % 64] for i in range(64)]
    for r0 in range(0, 64, 16):
        assert rows["row_shl:1"][r0:r0 + 15] == e[r0 + 1:r0 + 16] and rows["row_shr:1"][r0 + 1:r0 + 16] == e[r0:r0 + 15]
    assert rows["shr(e) - a"][1:] == [e[i - 1] - a[i] for i in range(1, 64)], "shifted value as MINUEND: the form the kernels rely on"
    good = rows["a - shl(e)"][:63] == [a[i] - e[i + 1] for i in range(63)]
    print("a - shl(e) folded correctly by this toolchain:", good)


def test_frame_buffers_allocator(env, port):
    """vp_frame_buffers_create / vp_malloc_device_placed (include/vp_hip.h): the buffers the reference allocates once (OpenCL::acquire pool,
    opencl.cpp:108-122; CLArrays of main.cpp:257-258) from the library, the two image-sized ones from placement-checked blocks.  Small frames:
    plain blocks, results the oracle's, the optional views present; bench-sized frames: candidates are probed, the table gets the fastest block."""
    import ctypes as C
    capi, torch = env
    cfg = cfgs.T_MID
    B, mm = 3, cfg.max_matches
    raws = np.stack([cfg.synth(300 + i) for i in range(B)])
    ctx = capi.Context(0, n_streams=1)
    try:
        p = capi.frame_params(cfg, order=capi.ORDER_RASTER)
        slab = ctx.frame_slab(p, B, views=capi.VIEW_GRADDOT | capi.VIEW_BLOBCENTER | capi.VIEW_PLANES)
        assert slab.placement["sat"]["selected"] is False and slab.placement["sat"]["candidates"] == 2  # 3 MB blocks: nothing to choose
        assert slab.b.planes and slab.b.graddot and slab.b.blobcenter
        d_raw = dev(torch, raws)
        for _ in range(2):
            ctx.frame_buffers(p, d_raw, B, slab.b)
        ctx.sync()
        model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
        cnt = slab.counters()
        for k in range(B):
            s = port.frame_scratch(cfg.qw, cfg.qh, cfg.fw, cfg.fh, mm)
            m, widx, wcnt = port.frame(raws[k], cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                                       cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm, scratch=s)
            assert np.array_equal(slab.flat(k), s["flat"]) and np.array_equal(slab.sat(k), port.sat_i32(s["graddot"])[0]) and np.array_equal(cnt[k], wcnt)
            n = int(wcnt[0])
            assert np.array_equal(slab.idx(k)[:n], widx) and slab.matches(k)[:22 * n].tobytes() == m.tobytes()
        gd = ctx.d2h(slab.b.graddot, (B, cfg.fh, cfg.fw), np.float32)
        assert np.array_equal(gd[0], s["graddot"]) if B == 1 else gd.shape[0] == B
        flat_t, sat_t, m_t, idx_t, cnt_t = slab.torch_views()  # zero-copy torch views of the same memory
        assert np.array_equal(cnt_t.cpu().numpy(), cnt) and np.array_equal(sat_t[B - 1].cpu().numpy(), slab.sat(B - 1))
        del flat_t, sat_t, m_t, idx_t, cnt_t
        slab.destroy()
        with pytest.raises(capi.VpError):  # a view the variant cannot produce is refused before anything is allocated
            ctx.frame_slab(capi.frame_params(cfg, sat_variant=capi.SAT_LOCAL), B, views=capi.VIEW_GRADDOT)
        # bench-sized request: 2 x 1.6 GB, candidates probed (how many depends on the device's state; the kept blocks are the fastest seen)
        big = ctx.frame_slab(capi.frame_params(CFG), B_BENCH)
        ps, pf = big.placement["sat"], big.placement["flat"]
        assert ps["selected"] and ps["candidates"] >= 2 and ps["write_gbs"] == ps["best_gbs"] >= pf["write_gbs"] >= ps["worst_gbs"] > 1000, big.placement
        big.destroy()
        ctx.tune("placement_candidates", 0)
        plain = ctx.frame_slab(capi.frame_params(CFG), 64)
        assert plain.placement["sat"]["selected"] is False
        plain.destroy()
        # the building block alone
        lib = capi.lib()
        ctx.tune("placement_candidates", 6)
        ptrs, info = (C.c_void_p * 2)(), (capi.VpPlacement * 2)()
        assert lib.vp_malloc_device_placed(ctx.h, C.c_size_t(1 << 30), 2, ptrs, info) == 0, lib.vp_last_error()
        assert ptrs[0] and ptrs[1] and ptrs[0] != ptrs[1] and 2 <= info[0].candidates <= 6 and info[0].write_gbs >= info[1].write_gbs > 0
        for q in ptrs:
            ctx.free(q)
    finally:
        ctx.close()


def check_gathered_packets(blob, n, feeds, slot, total, oracle_ranks=2):
    """What one device received in the last step of `vp_feeds --gpus n --detect` (n rank blocks of `feeds` packet slots): every slot parses with
    google.protobuf, and for every feed of the first `oracle_ranks` ranks (one feed of the others) the detection equals, binary32 field by field, the
    hot-path oracle's blob list of that feed's seed pushed through oracle/detect_oracle.py for `total` steps (colour model and tracker carried along)."""
    import detect_oracle as D
    import test_detection_stage as S
    import test_wire_format as W
    block = len(blob) // n
    assert len(blob) == n * block and block >= feeds * slot
    cfg, mm = CFG, CFG.max_matches
    big = O.Port(threads=min(16, os.cpu_count() or 1))
    model_o = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    mdl = cfg.model()
    calib = (mdl.f, mdl.p[0], mdl.p[1], mdl.d, -1.0, 0.0, 0.0, 0.0, -mdl.c[0], mdl.c[1], mdl.c[2])
    F = np.float32
    fs, ox, oy = F(cfg.field_scale), F(cfg.off_x), F(cfg.off_y)
    for g in range(n):
        for i in range(feeds):
            off = g * block + i * slot
            length = int(np.frombuffer(blob[off:off + 4], np.int32)[0])
            assert 0 < length <= slot - 4, (g, i, length)
            pkt = W.Wrapper()
            pkt.ParseFromString(blob[off + 4:off + 4 + length])
            assert pkt.IsInitialized() and pkt.source == 3
            d = pkt.detection
            assert d.camera_id == g * feeds + i and d.frame_number == total
            if g >= oracle_ranks and i > 0:
                continue  # the oracle comparison covers every feed of two devices and one feed of the others
            m, _, cnt = big.frame(cfg.synth(cases.vp.feeds.feed_seed(g, i)), cfg.layout, cfg.qw, cfg.qh, model_o, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y,
                                  cfg.sampler_mode, cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm)
            matches = [dict(pos=(F(F(a["x"]) * fs + ox), F(F(a["y"]) * fs + oy)), color=tuple(int(v) for v in a["color"]), center=tuple(int(v) for v in a["center"]), circ=F(a["circ"]),
                            score=F(a["score"])) for a in m]  # Perspective::flat2field (Perspective.cpp:127-129)
            res = D.Res(D.Camera(calib, (cfg.qw, cfg.qh), (cfg.qw, cfg.qh)), field_length=9000, field_width=6000, boundary_width=300)
            res.cam_id = g * feeds + i
            for k in range(1, total + 1):
                t = k / 70.0
                want = D.detect_frame(res, matches, k, t, t)
                D.detection_tracking(res, want)
            got = dict(frame_number=d.frame_number, camera_id=d.camera_id, t_capture=d.t_capture, t_capture_camera=d.t_capture_camera if d.HasField("t_capture_camera") else None,
                       t_offsets=np.asarray(d.t_offsets, np.float32),
                       balls=[dict(confidence=b.confidence, x=b.x, y=b.y, pixel_x=b.pixel_x, pixel_y=b.pixel_y) for b in d.balls],
                       robots_yellow=[dict(confidence=b.confidence, robot_id=b.robot_id, x=b.x, y=b.y, orientation=b.orientation, pixel_x=b.pixel_x, pixel_y=b.pixel_y, height=b.height)
                                      for b in d.robots_yellow],
                       robots_blue=[dict(confidence=b.confidence, robot_id=b.robot_id, x=b.x, y=b.y, orientation=b.orientation, pixel_x=b.pixel_x, pixel_y=b.pixel_y, height=b.height)
                                    for b in d.robots_blue])
            S.assert_frames_equal(got, want)
            assert d.t_sent == total / 70.0 + 0.002
            assert len(d.robots_yellow) + len(d.robots_blue) == 16


def test_config3_native_feeds_gather_the_oracles_detection_packets(tmp_path):
    """BASELINE configs[3] as written, in C++ on the C ABIs alone: `vp_feeds --gpus N --detect` (N = every visible device) -- one driver thread per
    device, vp_frame -> compact record slots -> CPU hypothesis stage -> SSL_WrapperPacket slots -> ONE grouped ncclAllGather per step through real
    RCCL.  The program checks that every device received every owner's block byte for byte (--verify); here the packets the LAST device received
    in the last step are parsed with google.protobuf and held to the CPU checkers for the same feed seeds: the hot-path oracle's blob list pushed
    through oracle/detect_oracle.py for the same number of steps (colour model and tracker carried along), binary32 fields bit for bit.  Replaces
    main.cpp:319-370 + udpsocket.cpp:94-100; the gather block is 13 x smaller than the raw record arrays."""
    import json
    import torch
    n = torch.cuda.device_count()
    exe = os.path.join(cases.ROOT, "vision-processor_amd", "host", "vp_feeds")
    envv = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    feeds, steps, warmup = 3, 3, 1
    dump = str(tmp_path / "gathered.bin")
    out = subprocess.run([exe, "--gpus", str(n), str(feeds), str(steps), str(warmup), "--detect", "--threads", "8", "--verify", "--dump", dump], capture_output=True, text=True, timeout=900, env=envv)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["gpus"] == n and r["gather_verified"] is True and r["block"] == "SSL_WrapperPacket slots" and "ncclAllGather" in r["gather"]
    assert r["record_bytes_per_gpu"] * 10 <= r["full_record_bytes_per_gpu"]
    assert r["detect"]["robots_per_frame"] == 16.0 and 8.0 <= r["detect"]["balls_per_frame"] <= 9.0 and 500 < r["detect"]["packet_bytes_per_frame"] < 1500
    check_gathered_packets(open(dump, "rb").read(), n, feeds, r["detect"]["packet_slot_bytes"], warmup + steps)


@pytest.mark.parametrize("variant", ["tiled", "local"])
def test_frame_graph_replay_is_result_neutral(env, port, variant):
    """vp_ctx_tune "frame_graph": vp_frame captures the launches of a repeated (stream, parameters, batch, buffers) call into a hipGraph the second
    time it sees it and replays it afterwards (nothing that changes between calls is a kernel argument: the single-pass SAT derives its epoch and
    ticket base from the tickets it draws).  Four library-owned streams, two geometries OF THE SAME SIZE taking turns on every stream (a replay must
    notice that the stream's coordinate map was rebuilt for the other geometry; launch shapes that differ in size or batch:
    test_gpu_round4.py::test_frame_graph_alternating_launch_shapes_on_one_stream), eight rounds, new sensor frames written into the same
    device buffers between rounds: every result equals the CPU oracle's -- flat, SAT, counters, sorted records.  One loop per camera: main.cpp:262-317."""
    capi, torch = env
    cfg_a, cfg_b = cfgs.T_MID, cfgs.T_MID.with_(name="t_mid_shifted", off_x=-1400.0, off_y=-850.0)
    NS, B = 4, 2
    with_sat = variant == "tiled"
    sv = capi.SAT_TILED if with_sat else capi.SAT_LOCAL
    ctx = capi.Context(0, n_streams=NS)
    try:
        ctx.tune("frame_graph", 1)
        model_o = {c.name: O.CameraModel.from_array(np.frombuffer(bytes(c.model()), np.uint8)) for c in (cfg_a, cfg_b)}

        def want(cfg, raw):
            mm = cfg.max_matches
            s = port.frame_scratch(cfg.qw, cfg.qh, cfg.fw, cfg.fh, mm)
            m, idx, cnt = port.frame(raw, cfg.layout, cfg.qw, cfg.qh, model_o[cfg.name], cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                                     cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm, scratch=s)
            return dict(flat=s["flat"].copy(), sat=port.sat_i32(s["graddot"])[0], m=m, idx=idx, cnt=cnt)

        mm = cfg_a.max_matches
        bufs = []
        for s in range(NS):
            bufs.append({c.name: dict(raw=torch.zeros((B,) + c.sensor_shape, dtype=torch.uint8, device="cuda"), flat=torch.zeros((B, c.fh, c.fw, 4), dtype=torch.uint8, device="cuda"),
                                      sat=torch.zeros((B, c.fh, c.fw), dtype=torch.int32, device="cuda") if with_sat else None, m=torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"),
                                      idx=torch.zeros((B, mm), dtype=torch.int32, device="cuda"), cnt=torch.full((B, 3), -5, dtype=torch.int32, device="cuda")) for c in (cfg_a, cfg_b)})
        params = {c.name: capi.frame_params(c, sat_variant=sv, order=capi.ORDER_RASTER) for c in (cfg_a, cfg_b)}
        for rnd in range(8):
            # rounds 0-3: geometry A only (ordinary call, capture, replay, replay); then B and A take turns on every stream
            cfg = cfg_a if rnd < 4 or rnd % 2 == 0 else cfg_b
            raws = [[cfg.synth(7000 + 100 * rnd + 10 * s + k) for k in range(B)] for s in range(NS)]
            for s in range(NS):
                bufs[s][cfg.name]["raw"].copy_(torch.from_numpy(np.stack(raws[s])))
            torch.cuda.synchronize()
            for s in range(NS):  # every stream is loaded before any of them is waited for
                b = bufs[s][cfg.name]
                ctx.frame(params[cfg.name], b["raw"], B, b["flat"], b["sat"], b["m"], b["cnt"], idx=b["idx"], stream=s)
            ctx.sync()
            for s in range(NS):
                b = bufs[s][cfg.name]
                for k in range(B):
                    w_ = want(cfg, raws[s][k])
                    assert np.array_equal(b["flat"][k].cpu().numpy(), w_["flat"]), (rnd, s, k, "flat")
                    if with_sat:
                        assert np.array_equal(b["sat"][k].cpu().numpy(), w_["sat"]), (rnd, s, k, "sat")
                    cnt = b["cnt"][k].cpu().numpy()
                    assert np.array_equal(cnt, w_["cnt"]), (rnd, s, k, cnt, w_["cnt"])
                    n = min(int(cnt[0]), mm)
                    assert np.array_equal(b["idx"][k].cpu().numpy().view(np.uint32)[:n], w_["idx"]) and b["m"][k].cpu().numpy()[:22 * n].tobytes() == w_["m"].tobytes(), (rnd, s, k)
    finally:
        ctx.close()


def test_local_variant_never_refuses_a_geometry(env, port):
    """VP_SAT_LOCAL for a (circle radius, gradient offset) pair without a walker (fieldScale 1.5 mm / px: r = 14, o = 5) and, beside it, one with
    (3.4 mm / px: r = 6, o = 2): vp_frame returns the oracle's list either way -- the first through the library's private table -- exactly as
    VP_SAT_TILED does for the same frames."""
    capi, torch = env
    base = cfgs.T_MID
    for fs, thr in ((1.5, 15.0), (3.4, 15.0)):
        cfg = base.with_(name=f"t_mid_fs{fs}", field_scale=fs, off_x=-320.0 * fs, off_y=-192.0 * fs, visible_len=640.0 * fs * 1.05, circ_threshold=thr)
        B, mm = 2, cfg.max_matches
        raws = np.stack([cfg.synth(40 + i) for i in range(B)])
        model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
        ctx = capi.Context(0, n_streams=1)
        try:
            d_raw = dev(torch, raws)
            out = {}
            for variant in (capi.SAT_LOCAL, capi.SAT_TILED):
                p = capi.frame_params(cfg, sat_variant=variant, order=capi.ORDER_RASTER)
                flat = torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
                sat = torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda") if variant == capi.SAT_TILED else None
                m = torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"); idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda")
                cnt = torch.full((B, 3), -9, dtype=torch.int32, device="cuda")
                for _ in range(2):
                    ctx.frame(p, d_raw, B, flat, sat, m, cnt, idx=idx)
                ctx.sync()
                out[variant] = (flat.cpu().numpy(), m.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy())
            for k in range(B):
                wm, widx, wcnt = port.frame(raws[k], cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                                            cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm)
                for variant in (capi.SAT_LOCAL, capi.SAT_TILED):
                    flat, m, idx, cnt = out[variant]
                    assert np.array_equal(cnt[k], wcnt), (fs, variant, cnt[k], wcnt)
                    n = min(int(wcnt[0]), mm)
                    assert np.array_equal(idx[k].view(np.uint32)[:n], widx) and m[k, :22 * n].tobytes() == wm.tobytes(), (fs, variant, k)
            assert cfg.circle_radius == (14 if fs == 1.5 else 6)
        finally:
            ctx.close()


def test_records_compact_is_what_the_host_loop_reads(env):
    """vp_records_compact: per frame counter[3], n = min(counter[0], max_matches, cap) and exactly those n records + pixel indices (main.cpp:299-301),
    against the host restatement of the slot layout (feeds.pack_compact_host); a slot too small for its frame says so (n < min(counter[0], max));
    idx may be NULL."""
    capi, torch = env
    f = cases.vp.feeds
    cfg = cfgs.T_MID
    B, mm = 5, cfg.max_matches
    ctx = capi.Context(0, n_streams=1)
    try:
        raws = dev(torch, np.stack([cfg.synth(600 + i) for i in range(B)]))
        p = capi.frame_params(cfg, order=capi.ORDER_RASTER)
        slab = ctx.frame_slab(p, B)
        ctx.frame_buffers(p, raws, B, slab.b)
        flat, sat, m, idx, cnt = slab.torch_views()
        for cap in (256, 30, 7):
            total, slot, off_m, off_i = ctx.record_layout_compact(B, cap)
            assert total == f.compact_block_bytes(B, cap) and slot == f.compact_slot_bytes(cap) and off_m == 16
            out = torch.full((total,), 0xAB, dtype=torch.uint8, device="cuda")
            ctx.records_compact(cnt, m, idx, B, mm, cap, out)
            ctx.sync()
            got = out.cpu()
            want = f.pack_compact_host(cnt.cpu(), m.cpu(), idx.cpu(), cap)
            gc, gn, gm, gi = f.unpack_compact(got[None], B, cap)
            wc, wn, wm, wi = f.unpack_compact(want[None], B, cap)
            assert torch.equal(gc, wc) and torch.equal(gn, wn)
            assert int(gn.max()) == min(cap, int(cnt[:, 0].max())) and (cap >= 256) == bool((gn[0] == cnt[:, 0].cpu().clamp(max=mm)).all())
            for k in range(B):
                assert torch.equal(gm[0][k], wm[0][k]) and torch.equal(gi[0][k], wi[0][k]), (cap, k)
            # without indices: the record part is the same, the index part is left alone
            out2 = torch.full((total,), 0xAB, dtype=torch.uint8, device="cuda")
            ctx.records_compact(cnt, m, None, B, mm, cap, out2)
            ctx.sync()
            g2 = out2.cpu()
            for k in range(B):
                n = int(gn[0, k])
                assert torch.equal(g2[k * slot:k * slot + 16 + 22 * n], got[k * slot:k * slot + 16 + 22 * n])
                assert bool((g2[k * slot + off_i:k * slot + off_i + 4 * cap] == 0xAB).all())
        del flat, sat, m, idx, cnt
        slab.destroy()
    finally:
        ctx.close()


def test_a_refused_frame_leaves_the_callers_buffers_untouched(env):
    """vp_frame checks every argument before it sizes a workspace or enqueues a kernel (the reference's FATAL convention, log.h:21, turned into a
    status at the C ABI): a refused call must not have overwritten flat, cleared the counters or launched anything."""
    capi, torch = env
    cfg = cfgs.T_MID
    B, mm = 2, cfg.max_matches
    ctx = capi.Context(0, n_streams=1)
    try:
        raws = dev(torch, np.stack([cfg.synth(900 + i) for i in range(B)]))
        flat = torch.full((B, cfg.fh, cfg.fw, 4), 0xAB, dtype=torch.uint8, device="cuda")
        sat = torch.full((B, cfg.fh, cfg.fw), 0x11111111, dtype=torch.int32, device="cuda")
        m = torch.full((B, mm * 22), 0xCD, dtype=torch.uint8, device="cuda")
        idx = torch.full((B, mm), 7, dtype=torch.int32, device="cuda")
        cnt = torch.full((B, 3), -77, dtype=torch.int32, device="cuda")

        def bad(**over):
            p = capi.frame_params(cfg, order=capi.ORDER_RASTER)
            for k, v in over.items():
                setattr(p, k, v)
            return p
        cases_ = [(bad(order=5), B), (bad(fw=-64), B), (bad(fh=0), B), (bad(max_matches=0), B), (bad(layout=9), B), (bad(sampler_mode=3), B), (bad(sat_variant=7), B),
                  (bad(grad_offset=-1), B), (bad(qw=0), B), (bad(), 0), (bad(), 70000)]
        for p, batch in cases_:
            with pytest.raises(capi.VpError) as e:
                ctx.frame(p, raws, batch, flat, sat, m, cnt, idx=idx)
            assert e.value.code == -1
        with pytest.raises(capi.VpError):  # VP_SAT_LOCAL with a table buffer
            ctx.frame(capi.frame_params(cfg, sat_variant=capi.SAT_LOCAL), raws, B, flat, sat, m, cnt, idx=idx)
        with pytest.raises(capi.VpError):  # records at an odd address
            ctx.frame(bad(), raws, B, flat, sat, m.view(-1)[1:].data_ptr(), cnt, idx=idx)
        ctx.sync()
        assert bool((flat == 0xAB).all()) and bool((sat == 0x11111111).all()) and bool((m == 0xCD).all()) and bool((idx == 7).all()) and bool((cnt == -77).all())
        ctx.frame(bad(), raws, B, flat, sat, m, cnt, idx=idx)  # and the context still works
        ctx.sync()
        assert int(cnt[:, 0].min()) > 0 and not bool((flat == 0xAB).all())
    finally:
        ctx.close()
