"""GPU parity tests added in round 2, all through the C ABI:
  * BASELINE.json configs[2]: eight HIP streams of one context with eight different frames in flight before any
    synchronisation (the one-loop-per-camera structure of main.cpp:262-317 on one device) -- every stream's flat, SAT, counters
    and sorted blob records byte-identical to the CPU oracle;
  * the fused circle + detect walker on awkward shapes and every specialised radius, against the oracle's circle + blobList;
  * the bounded candidate list: a capacity smaller than the number of candidates still gives the exact list (rescan path);
  * blocking map / unmap from other host threads while the main thread launches (rtpstreamer.cpp:177-181, snapshotwriter.cpp:52-54).
"""
import ctypes as C
import threading

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
    capi = cases.vp.load_capi()
    return capi, torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def oracle_frame(port, cfg, raw, mm):
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    s = port.frame_scratch(cfg.qw, cfg.qh, cfg.fw, cfg.fh, mm)
    m, idx, cnt = port.frame(raw, cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                             cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm, scratch=s)
    sat, mx = port.sat_i32(s["graddot"])
    assert mx < 2 ** 24
    return dict(flat=s["flat"].copy(), sat=sat, m=m, idx=idx, cnt=cnt)


@pytest.mark.parametrize("variant", [0, 1], ids=["tiled", "twopass"])
@pytest.mark.parametrize("batch", [1, 8])
@pytest.mark.parametrize("cfg", [cfgs.T_MID, cfgs.CFG1], ids=lambda c: c.name)
def test_config2_eight_streams_in_flight(env, port, cfg, batch, variant):
    """One context, n_streams = 8, 8 x batch distinct frames; vp_frame is issued on all eight streams (three rounds, so that the per-stream
    SAT epochs, ticket bases, coordinate-map keys and workspaces are all reused under overlap) before anything is synchronised."""
    capi, torch = env
    NS, mm = 8, cfg.max_matches
    ctx = capi.Context(0, n_streams=NS)
    try:
        raws = [[cfg.synth(9000 + 31 * s + b) for b in range(batch)] for s in range(NS)]
        want = [[oracle_frame(port, cfg, raws[s][b], mm) for b in range(batch)] for s in range(NS)]
        bufs = []
        for s in range(NS):
            bufs.append(dict(raw=dev(torch, np.stack(raws[s])), flat=torch.zeros((batch, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"),
                             sat=torch.zeros((batch, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"), m=torch.zeros((batch, mm * 22), dtype=torch.uint8, device="cuda"),
                             idx=torch.zeros((batch, mm), dtype=torch.int32, device="cuda"), cnt=torch.full((batch, 3), -5, dtype=torch.int32, device="cuda")))
        torch.cuda.synchronize()
        p = capi.frame_params(cfg, sat_variant=variant, order=capi.ORDER_RASTER, use_coord_map=True)
        for rnd in range(3):
            for s in range(NS):  # all eight streams are loaded before any of them is waited for
                b = bufs[s]
                ctx.frame(p, b["raw"], batch, b["flat"], b["sat"], b["m"], b["cnt"], idx=b["idx"], stream=s)
        ctx.sync()
        for s in range(NS):
            b = bufs[s]
            flat, sat, m, idx, cnt = (b[k].cpu().numpy() for k in ("flat", "sat", "m", "idx", "cnt"))
            for i in range(batch):
                w_ = want[s][i]
                assert np.array_equal(flat[i], w_["flat"]), (s, i, "flat")
                assert np.array_equal(sat[i], w_["sat"]), (s, i, "sat")
                assert np.array_equal(cnt[i], w_["cnt"]), (s, i, cnt[i], w_["cnt"])
                n = min(int(cnt[i][0]), mm)
                got = np.frombuffer(m[i].tobytes(), capi.MATCH_DTYPE)[:n]
                assert np.array_equal(idx[i].view(np.uint32)[:n], w_["idx"]), (s, i, "RASTER order = sorted pixel indices")
                assert got.tobytes() == w_["m"].tobytes(), (s, i, "records")
    finally:
        ctx.close()


WALK_SHAPES = [(62, 40), (63, 17), (124, 64), (125, 33), (200, 150), (64, 3), (5, 7), (1, 1), (640, 70)]


@pytest.mark.parametrize("r", [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 1, 13])
def test_circle_walk_edge_shapes(env, port, r):
    """circle + detect straight from the SAT for every specialised radius (2..12: the register walker) and two others (the LDS
    fallback), on shapes around the 62-column strips and short segments, at several rows-per-wavefront settings: the blobCenter
    view, the counters and the sorted records equal the oracle's circle (satBlobCenter.cl) + matches (blobList.cl)."""
    capi, torch = env
    ctx = capi.Context(0, n_streams=1)
    try:
        rng = np.random.default_rng(100 + r)
        for (w, h), seg in zip(WALK_SHAPES, [0, 16, 16, 32, 64, 0, 16, 0, 128]):
            B, mm, R = 2, 300, 2
            gd = rng.integers(-900, 900, (B, h, w)).astype(np.float32)
            # a few strong rings so that there are candidates, including at the borders and at strip boundaries
            for _ in range(6):
                cx, cy = int(rng.integers(0, w)), int(rng.integers(0, h))
                gd[:, max(cy - 2, 0):cy + 3, max(cx - 2, 0):cx + 3] += rng.integers(500, 5000)
            flat = rng.integers(0, 256, (B, h, w, 4), dtype=np.uint8)
            flat[..., 3] = 255
            sats = np.stack([port.sat_i32(gd[i])[0] for i in range(B)])
            thr = 20.0
            d_sat, d_flat = dev(torch, sats), dev(torch, flat)
            ctx.tune("tail_segment_rows", seg)
            for order in (capi.ORDER_RASTER, capi.ORDER_ATOMIC):
                d_m = torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda")
                d_idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda")
                d_cnt = torch.full((B, 3), -3, dtype=torch.int32, device="cuda")
                d_bc = torch.full((B, h, w), np.nan, dtype=torch.float32, device="cuda")
                with_bc = order == capi.ORDER_RASTER
                ctx.blobcenter_list(d_flat, d_sat, w, h, B, r, thr, 0.0, R, mm, order, d_m, d_idx, d_cnt, d_bc if with_bc else None)
                ctx.sync()
                for i in range(B):
                    bc = port.blobcenter_i32(sats[i], r)
                    if with_bc:
                        assert np.array_equal(d_bc[i].cpu().numpy(), bc, equal_nan=True), (w, h, r, "blobCenter view")
                    wm, widx, wcnt = port.bloblist(flat[i], bc, thr, 0.0, R, mm)
                    cnt = d_cnt[i].cpu().numpy()
                    assert np.array_equal(cnt, wcnt), (w, h, r, seg, cnt, wcnt)
                    gm, gi = capi.sorted_blobs(d_m[i].cpu().numpy(), d_idx[i].cpu().numpy().view(np.uint32), cnt, mm)
                    if int(cnt[0]) <= mm:
                        assert np.array_equal(gi, widx) and gm.tobytes() == wm.tobytes(), (w, h, r, seg)
                        if order == capi.ORDER_RASTER:  # already sorted on the device
                            assert np.array_equal(d_idx[i].cpu().numpy().view(np.uint32)[:len(widx)], widx)
    finally:
        ctx.close()


@pytest.mark.parametrize("cap", [1, 7, 64])
def test_candidate_list_overflow_is_exact(env, port, cap):
    """More thresholded local maxima than candidate slots: the emit pass rescans the frame and the result is still the oracle's --
    counters, the blob set, and (RASTER order) the order.  Both the fused path (from the SAT) and the path from a materialised
    blobCenter image; a frame below the capacity travels in the same batch."""
    capi, torch = env
    cfg = cfgs.T_MID
    ctx = capi.Context(0, n_streams=1)
    try:
        mm = 400
        raws = [cfg.synth(77), cfg.synth(78)]
        s = [port.frame_scratch(cfg.qw, cfg.qh, cfg.fw, cfg.fh, mm) for _ in raws]
        model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
        thr = 2.0  # low threshold: hundreds of candidates per frame
        want = [port.frame(r, cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode, cfg.fw, cfg.fh,
                           cfg.grad_offset, cfg.circle_radius, cfg.list_radius, thr, 0.5, mm, scratch=sc) for r, sc in zip(raws, s)]
        assert all(int(w_[2][0]) + int(w_[2][1]) > 64 for w_ in want), "the case must overflow every tested capacity"
        flat = np.stack([sc["flat"] for sc in s])
        sats = np.stack([port.sat_i32(sc["graddot"])[0] for sc in s])
        bcs = np.stack([sc["blobcenter"] for sc in s])
        d_flat, d_sat, d_bc = dev(torch, flat), dev(torch, sats), dev(torch, bcs)
        ctx.tune("candidate_capacity", cap)
        for fused in (True, False):
            for order in (capi.ORDER_RASTER, capi.ORDER_ATOMIC):
                d_m = torch.zeros((2, mm * 22), dtype=torch.uint8, device="cuda")
                d_idx = torch.zeros((2, mm), dtype=torch.int32, device="cuda")
                d_cnt = torch.full((2, 3), -3, dtype=torch.int32, device="cuda")
                for rep in range(2):  # twice: the self-cleaning counters must be back at zero after an overflowing launch
                    if fused:
                        ctx.blobcenter_list(d_flat, d_sat, cfg.fw, cfg.fh, 2, cfg.circle_radius, thr, 0.5, cfg.list_radius, mm, order, d_m, d_idx, d_cnt)
                    else:
                        ctx.bloblist(d_flat, d_bc, cfg.fw, cfg.fh, 2, thr, 0.5, cfg.list_radius, mm, order, d_m, d_idx, d_cnt)
                    ctx.sync()
                    for i in range(2):
                        wm, widx, wcnt = want[i]
                        cnt = d_cnt[i].cpu().numpy()
                        assert np.array_equal(cnt, wcnt), (fused, order, rep, cnt, wcnt)
                        gm, gi = capi.sorted_blobs(d_m[i].cpu().numpy(), d_idx[i].cpu().numpy().view(np.uint32), cnt, mm)
                        if int(cnt[0]) <= mm:
                            assert np.array_equal(gi, widx) and gm.tobytes() == wm.tobytes(), (fused, order, rep)
    finally:
        ctx.close()


def test_map_and_unmap_from_other_threads_while_the_main_thread_launches(env, port):
    """The reference maps NV12 buffers from its RTP encoder thread and images from its snapshot thread while the main loop keeps
    launching (rtpstreamer.cpp:177-181, snapshotwriter.cpp:52-54).  Here: the main thread runs vp_frame in a loop on stream 0; thread A
    fills, writes back and re-reads its own buffer on stream 1; threads B and C map the SAME image concurrently (first map allocates
    the pinned shadow) on streams 2 and 3.  Every map returns what was written; the frames' results stay the oracle's."""
    capi, torch = env
    cfg = cfgs.T_MID
    mm = cfg.max_matches
    ctx = capi.Context(0, n_streams=4)
    lib = capi.lib()
    h = ctx.h
    errors = []
    try:
        raw = cfg.synth(5)
        want = oracle_frame(port, cfg, raw, mm)
        d_raw = dev(torch, raw[None])
        flat = torch.zeros((1, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
        sat = torch.zeros((1, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda")
        m = torch.zeros((1, mm * 22), dtype=torch.uint8, device="cuda")
        idx = torch.zeros((1, mm), dtype=torch.int32, device="cuda")
        cnt = torch.zeros((1, 3), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        nbytes = ctx.nv12_bytes(320, 192)
        buf = C.c_void_p()
        assert lib.vp_buffer_create(h, C.c_size_t(nbytes), None, C.byref(buf)) == 0
        img = C.c_void_p()
        assert lib.vp_image_acquire(h, 2, 96, 64, C.byref(img)) == 0  # F32
        pattern = np.arange(96 * 64, dtype=np.float32)
        hp = C.c_void_p()
        assert lib.vp_image_map(h, 3, img, 0, C.byref(hp)) == 0
        C.memmove(hp, pattern.ctypes.data, pattern.nbytes)
        assert lib.vp_image_unmap(h, 3, img, 1) == 0
        stop = threading.Event()

        def writer():  # thread A: its own NV12 buffer
            try:
                k = 0
                while not stop.is_set():
                    p_ = C.c_void_p()
                    assert lib.vp_buffer_map(h, 1, buf, 0, C.byref(p_)) == 0
                    C.memset(p_, k & 0xFF, nbytes)
                    assert lib.vp_buffer_unmap(h, 1, buf, 1) == 0
                    assert lib.vp_buffer_map(h, 1, buf, 1, C.byref(p_)) == 0
                    got = np.ctypeslib.as_array(C.cast(p_, C.POINTER(C.c_uint8)), (nbytes,))
                    assert int(got.min()) == int(got.max()) == (k & 0xFF), "NV12 buffer round trip"
                    assert lib.vp_buffer_unmap(h, 1, buf, 0) == 0
                    k += 1
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        def reader(stream):  # threads B, C: the same image
            try:
                while not stop.is_set():
                    p_ = C.c_void_p()
                    assert lib.vp_image_map(h, stream, img, 1, C.byref(p_)) == 0
                    got = np.ctypeslib.as_array(C.cast(p_, C.POINTER(C.c_float)), (96 * 64,))
                    assert np.array_equal(got, pattern), "image read-back"
                    assert lib.vp_image_unmap(h, stream, img, 0) == 0
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=writer), threading.Thread(target=reader, args=(2,)), threading.Thread(target=reader, args=(3,))]
        for t in threads:
            t.start()
        try:
            p = capi.frame_params(cfg, order=capi.ORDER_RASTER)
            for it in range(150):
                ctx.frame(p, d_raw, 1, flat, sat, m, cnt, idx=idx, stream=0)
                if it % 10 == 9:
                    ctx.sync(0)
                    c = cnt.cpu().numpy()[0]
                    assert np.array_equal(c, want["cnt"])
        finally:  # the helper threads must be gone before the context is
            stop.set()
            for t in threads:
                t.join(30)
        assert not errors, errors
        ctx.sync()
        n = min(int(want["cnt"][0]), mm)
        assert np.array_equal(idx.cpu().numpy()[0].view(np.uint32)[:n], want["idx"])
        assert np.array_equal(sat.cpu().numpy()[0], want["sat"])
        assert lib.vp_buffer_destroy(h, buf) == 0 and lib.vp_image_release(h, img) == 0
    finally:
        ctx.close()


def test_native_multi_gpu_feeds_gather_through_rccl(port):
    """host/vp_feeds --gpus N (N = every visible device; 1 on the test box): one process, one context per device, vp_frame writing straight
    into each device's record block, one grouped ncclAllGather per step through real RCCL (include/vp_multi.h).  The program checks that
    every device received every owner's block byte for byte; here the gathered counters are additionally held to the CPU oracle's for
    the same feed seeds (vision-processor_amd/feeds.py feed_seed)."""
    import json
    import os
    import subprocess
    import torch
    n = torch.cuda.device_count()
    exe = os.path.join(cases.ROOT, "vision-processor_amd", "host", "vp_feeds")
    envv = dict(os.environ)
    envv["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + envv.get("LD_LIBRARY_PATH", "")
    feeds = 2
    out = subprocess.run([exe, "--gpus", str(n), str(feeds), "3", "1", "--verify"], capture_output=True, text=True, timeout=600, env=envv)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["gpus"] == n and r["gather_verified"] is True and "ncclAllGather" in r["gather"] and r["sat_variant"] == "tiled"
    assert r["block"] == "compact records" and r["record_bytes_per_gpu"] * 10 <= r["full_record_bytes_per_gpu"]  # the gather is sized to its content
    full = subprocess.run([exe, "--gpus", str(n), str(feeds), "2", "1", "--verify", "--full-records"], capture_output=True, text=True, timeout=600, env=envv)
    assert full.returncode == 0, full.stdout + full.stderr
    rf = json.loads(full.stdout.strip().splitlines()[-1])
    assert rf["block"] == "full records" and rf["gather_verified"] is True and rf["counters"] == r["counters"]
    out2 = subprocess.run([exe, "--gpus", str(n), str(feeds), "3", "1", "--verify", "--local"], capture_output=True, text=True, timeout=600, env=envv)
    assert out2.returncode == 0, out2.stdout + out2.stderr
    r2 = json.loads(out2.stdout.strip().splitlines()[-1])
    assert r2["gather_verified"] is True and r2["sat_variant"] == "local" and r2["counters"] == r["counters"]  # the variant without a global table: same lists
    cfg = cfgs.CFG2
    big = O.Port(threads=8)
    for g in range(min(n, 2)):
        for i in range(feeds):
            want = oracle_frame(big, cfg, cfg.synth(cases.vp.feeds.feed_seed(g, i)), cfg.max_matches)
            assert r["counters"][g][i] == [int(v) for v in want["cnt"]], (g, i)


GATHER_ABI_CHILD = r"""
import ctypes as C, os, sys
import numpy as np
root = sys.argv[1]
hip = C.CDLL(os.path.join(root, "vision-processor_amd", "libvp_hip.so"), mode=C.RTLD_GLOBAL)
multi = C.CDLL(os.path.join(root, "vision-processor_amd", "libvp_multi.so"))
hip.vp_last_error.restype = C.c_char_p
multi.vp_gather_send.restype = C.c_void_p
multi.vp_gather_recv.restype = C.c_void_p
multi.vp_record_layout.restype = C.c_size_t
multi.vp_packet_layout.restype = C.c_size_t
hip.vp_record_layout_compact.restype = C.c_size_t
n = hip.vp_device_count()
assert n >= 1
nbytes = multi.vp_record_layout(2, 64, None, None, None)
assert nbytes > 0 and nbytes % 256 == 0
assert multi.vp_packet_layout(3, 4096) == 12288 and multi.vp_packet_layout(3, 4) == 0 and multi.vp_packet_layout(0, 4096) == 0
slot = C.c_size_t()
assert hip.vp_record_layout_compact(128, 192, C.byref(slot), None, None) * 10 <= multi.vp_record_layout(128, 2000, None, None, None) and slot.value % 16 == 0
ctxs = []
for d in range(n):
    h = C.c_void_p()
    assert hip.vp_ctx_create(d, 2, C.byref(h)) == 0, hip.vp_last_error()
    ctxs.append(h)
g = C.c_void_p()
devs = (C.c_int * n)(*range(n))
assert multi.vp_gather_create(n, devs, C.c_size_t(nbytes), 2, C.byref(g)) == 0, hip.vp_last_error()
harr = (C.c_void_p * n)(*[c.value for c in ctxs])
for slot_i in (0, 1, 0):
    blocks = [np.random.default_rng(17 * slot_i + r).integers(0, 256, nbytes, dtype=np.uint8) for r in range(n)]
    for r in range(n):
        assert hip.vp_copy_h2d(ctxs[r], slot_i, C.c_void_p(multi.vp_gather_send(g, r, slot_i)), blocks[r].ctypes.data_as(C.c_void_p), C.c_size_t(nbytes)) == 0
        assert hip.vp_sync(ctxs[r], slot_i) == 0
    streams = (C.c_int * n)(*([slot_i] * n))
    assert multi.vp_gather_start(g, slot_i, harr, streams) == 0, hip.vp_last_error()
    assert multi.vp_gather_wait(g, slot_i) == 0
    for r in range(n):
        got = np.empty((n, nbytes), np.uint8)
        assert hip.vp_copy_d2h_sync(ctxs[r], slot_i, got.ctypes.data_as(C.c_void_p), C.c_void_p(multi.vp_gather_recv(g, r, slot_i)), C.c_size_t(got.nbytes)) == 0
        for q in range(n):
            assert np.array_equal(got[q], blocks[q]), (slot_i, r, q)
if n >= 2:  # a context on the wrong device is refused
    swapped = (C.c_void_p * n)(*[c.value for c in reversed(ctxs)])
    assert multi.vp_gather_start(g, 0, swapped, streams) != 0
multi.vp_gather_destroy(g)
for c in ctxs:
    hip.vp_ctx_destroy(c)
print("GATHER_ABI_OK", n)
"""


def test_gather_abi_in_a_child_process():
    """include/vp_multi.h through ctypes, in a child process WITHOUT torch (libvp_multi.so brings the system HIP runtime and RCCL, which must not meet
    PyTorch's bundled copies in one process): N = visible devices, two slots, blocks filled by device copies on the contexts' streams; after
    vp_gather_start / vp_gather_wait every device's receive area holds all blocks in rank order; the three block layouts report their sizes."""
    import os
    import subprocess
    import sys
    envv = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([sys.executable, "-c", GATHER_ABI_CHILD, cases.ROOT], capture_output=True, text=True, timeout=600, env=envv)
    assert out.returncode == 0 and "GATHER_ABI_OK" in out.stdout, out.stdout + out.stderr


# ---- VP_SAT_LOCAL: no global table ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,mode,seed", cases.CHAIN_CASES, ids=[cases.chain_case_name(*c) for c in cases.CHAIN_CASES])
def test_local_table_variant_on_the_goldens(env, cfg, mode, seed):
    """vp_frame with sat_variant = VP_SAT_LOCAL (buffers.sat = NULL): flat, counters and the sorted records equal the golden vectors, i.e.
    what the global-table path and the reference-derived fixtures give, bit for bit."""
    capi, torch = env
    g = cases.load_golden(cases.chain_case_name(cfg, mode, seed))
    ctx = capi.Context(0, n_streams=1)
    try:
        mm = cfg.max_matches
        for order in (capi.ORDER_RASTER, capi.ORDER_ATOMIC):
            for seg in (0, 16, 64):
                ctx.tune("tail_segment_rows", seg)
                p = capi.frame_params(cfg, sat_variant=capi.SAT_LOCAL, order=order, sampler_mode=mode)
                flat = torch.zeros((cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
                m = torch.zeros(mm * 22, dtype=torch.uint8, device="cuda")
                idx = torch.zeros(mm, dtype=torch.int32, device="cuda")
                cnt = torch.full((3,), -9, dtype=torch.int32, device="cuda")
                ctx.frame(p, dev(torch, g["raw"]), 1, flat, None, m, cnt, idx=idx)
                ctx.sync()
                assert np.array_equal(flat.cpu().numpy(), g["flat"])
                c = cnt.cpu().numpy()
                assert np.array_equal(c, g["counter"]), (order, seg, c, g["counter"])
                gm, gi = capi.sorted_blobs(m.cpu().numpy(), idx.cpu().numpy().view(np.uint32), c, mm)
                wm = np.frombuffer(np.ascontiguousarray(g["matches"]).tobytes(), capi.MATCH_DTYPE)
                assert np.array_equal(gi, g["idx"]) and gm.tobytes() == wm.tobytes(), (order, seg)
    finally:
        ctx.close()


@pytest.mark.parametrize("name", sorted(cases.flat_cases()))
def test_local_table_variant_on_the_flat_edge_cases(env, name):
    """The hand-made flat images (blobs on every border, plateaus, uniform discs with infinite score, list overflow, the 2^24-guard violation, o = 0)
    through vp_flat_bloblist: the walker where the (r, o) pair has one, the private-table route otherwise -- never refused."""
    capi, torch = env
    case = cases.flat_cases()[name]
    g = cases.load_golden(name)
    if "blobcenter" in case:
        pytest.skip("case starts from a hand-made blobCenter image")
    ctx = capi.Context(0, n_streams=1)
    try:
        flat = case["flat"]
        h, w = flat.shape[:2]
        mm = case["max_matches"]
        m = torch.zeros(mm * 22, dtype=torch.uint8, device="cuda")
        idx = torch.zeros(mm, dtype=torch.int32, device="cuda")
        cnt = torch.full((3,), -9, dtype=torch.int32, device="cuda")
        # (r, o) pairs no geometry forms (r = 6 with o = 1, o = 0, ...) have no walker: the same list through the library's private table
        ctx.flat_bloblist(dev(torch, flat), w, h, 1, case["o"], case["r"], case["thr"], case["min_score"], case["R"], mm, capi.ORDER_RASTER, m, idx, cnt)
        ctx.sync()
        c = cnt.cpu().numpy()
        assert np.array_equal(c, g["counter"]), (c, g["counter"])
        if int(c[0]) <= mm:
            gm, gi = capi.sorted_blobs(m.cpu().numpy(), idx.cpu().numpy().view(np.uint32), c, mm)
            wm = np.frombuffer(np.ascontiguousarray(g["matches"]).tobytes(), capi.MATCH_DTYPE)
            assert np.array_equal(gi, g["idx"]) and gm.tobytes() == wm.tobytes()
    finally:
        ctx.close()


@pytest.mark.parametrize("cfg", [cfgs.T_MID, cfgs.CFG1, cfgs.CFG2, cfgs.CFG5], ids=lambda c: c.name)
def test_local_table_variant_equals_the_global_table_path(env, cfg):
    """Batches at the mid and the BASELINE sizes (2048x1536 r = 5 o = 2, 4096x3072 r = 9 o = 3, 640x480 BGR r = 4 o = 1): VP_SAT_LOCAL and
    VP_SAT_TILED produce the same counters and the same records in the same (RASTER) order, byte for byte."""
    capi, torch = env
    B = 3 if cfg.n_flat > 4e6 else 5
    mm = cfg.max_matches
    raws = dev(torch, np.stack([cfg.synth(4000 + i) for i in range(B)]))
    ctx = capi.Context(0, n_streams=1)
    try:
        outs = []
        for variant in (capi.SAT_TILED, capi.SAT_LOCAL):
            p = capi.frame_params(cfg, sat_variant=variant, order=capi.ORDER_RASTER)
            flat = torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
            sat = torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda") if variant == capi.SAT_TILED else None
            m = torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda")
            idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda")
            cnt = torch.full((B, 3), -9, dtype=torch.int32, device="cuda")
            for rep in range(2):
                ctx.frame(p, raws, B, flat, sat, m, cnt, idx=idx)
            ctx.sync()
            outs.append((flat.cpu().numpy(), m.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()))
        a, b = outs
        assert np.array_equal(a[3], b[3]), (a[3], b[3])
        assert int(a[3][:, 0].min()) > 10
        assert np.array_equal(a[0], b[0])
        for i in range(B):
            nrec = min(int(a[3][i, 0]), mm)
            assert np.array_equal(a[2][i, :nrec], b[2][i, :nrec]) and a[1][i, :nrec * 22].tobytes() == b[1][i, :nrec * 22].tobytes(), i
    finally:
        ctx.close()


@pytest.mark.parametrize("cfg", [cfgs.T_MID, cfgs.T_RAGGED.with_(name="t_ragged_even", qw=68, qh=46), cfgs.CFG2], ids=lambda c: c.name)
def test_tuning_knobs_are_result_neutral(env, cfg):
    """vp_ctx_tune promises bit-identical output for every value: halo strip records on / off, 2..6 resident SAT workgroups per CU, both
    resampling stage formats and tile mappings, three rows-per-wavefront settings of the tail -- flat, SAT, counters, records all equal."""
    capi, torch = env
    B, mm = 5, cfg.max_matches
    raws = dev(torch, np.stack([cfg.synth(5100 + i) for i in range(B)]))
    p = capi.frame_params(cfg, order=capi.ORDER_RASTER)
    want = None
    for knobs in ({}, {"sat_halo_strips": 0}, {"sat_workgroups_per_cu": 2}, {"sat_workgroups_per_cu": 6}, {"resample_float_stage": 0}, {"resample_float_stage": 1}, {"resample_groups_per_wg": 1}, {"resample_groups_per_wg": 5}, {"resample_xcd_bands": 1},
                  {"resample_xcd_bands": 0, "tail_segment_rows": 16}, {"tail_segment_rows": 192, "sat_halo_strips": 0}, {"candidate_capacity": 8}):
        ctx = capi.Context(0, n_streams=1)
        try:
            for k, v in knobs.items():
                ctx.tune(k, v)
            flat = torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
            sat = torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda")
            m = torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda")
            idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda")
            cnt = torch.full((B, 3), -9, dtype=torch.int32, device="cuda")
            for rep in range(2):
                ctx.frame(p, raws, B, flat, sat, m, cnt, idx=idx)
            ctx.sync()
            got = (flat.cpu().numpy(), sat.cpu().numpy(), cnt.cpu().numpy(), idx.cpu().numpy(), m.cpu().numpy())
        finally:
            ctx.close()
        if want is None:
            want = got
            assert int(got[2][:, 0].min()) > 0
            continue
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), knobs
        for i in range(B):
            nrec = min(int(want[2][i, 0]), mm)
            assert np.array_equal(got[3][i, :nrec], want[3][i, :nrec]) and got[4][i, :nrec * 22].tobytes() == want[4][i, :nrec * 22].tobytes(), (knobs, i)
    with pytest.raises(Exception):
        ctx = capi.Context(0, n_streams=1)
        try:
            ctx.tune("no_such_knob", 1)
        finally:
            ctx.close()


@pytest.mark.parametrize("r,o", [(4, 1), (5, 2), (9, 3)])
def test_local_variant_ignores_the_alpha_byte(env, r, o):
    """gradientDot is formed from the three colour bytes only (gradientDot.cl:25-29).  The local walker computes it with byte dot products on
    alpha-masked pixels: a flat image with RANDOM alpha bytes must give the records of the global-table path (vp_graddot_sat + vp_blobcenter_list),
    whose gradient code never touches byte 3, and the records of the same image with alpha = 255."""
    capi, torch = env
    rng = np.random.default_rng(77 + r)
    h, w, mm, B = 200, 320, 6000, 2
    flat = rng.integers(0, 256, (B, h, w, 4), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(40):  # discs on noise: plenty of ring-like maxima
        cy, cx = rng.integers(0, h), rng.integers(0, w)
        flat[:, (yy - cy) ** 2 + (xx - cx) ** 2 <= (r + 1) ** 2, :3] = rng.integers(0, 256, 3, dtype=np.uint8)
    opaque = flat.copy(); opaque[..., 3] = 255
    thr = 40.0
    ctx = capi.Context(0, n_streams=1)
    try:
        def run(img, local):
            d = dev(torch, img)
            m = torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"); idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda")
            cnt = torch.full((B, 3), -9, dtype=torch.int32, device="cuda")
            if local:
                ctx.flat_bloblist(d, w, h, B, o, r, thr, 0.0, 3, mm, capi.ORDER_RASTER, m, idx, cnt)
            else:
                sat = torch.zeros((B, h, w), dtype=torch.int32, device="cuda")
                ctx.graddot_sat(d, w, h, o, B, capi.SAT_TILED, sat)
                cnt.zero_()
                ctx.blobcenter_list(d, sat, w, h, B, r, thr, 0.0, 3, mm, capi.ORDER_RASTER, m, idx, cnt)
            ctx.sync()
            return cnt.cpu().numpy(), idx.cpu().numpy(), m.cpu().numpy()
        want = run(flat, False)
        assert int(want[0][:, 0].min()) > 5 and int(want[0][:, 0].max()) <= mm  # (beyond max_matches the kept subset is the atomics' order)
        for got in (run(flat, True), run(opaque, True)):
            assert np.array_equal(got[0][:, :2], want[0][:, :2]), (got[0], want[0])
            for i in range(B):
                n = min(int(want[0][i, 0]), mm)
                assert np.array_equal(got[1][i, :n], want[1][i, :n])
                # the record's colour / centre fields come from flat itself (alpha is not part of a record): compare everything
                assert got[2][i, :22 * n].tobytes() == want[2][i, :22 * n].tobytes()
    finally:
        ctx.close()
