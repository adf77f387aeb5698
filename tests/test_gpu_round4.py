"""GPU tests added in round 4, all through the C ABI or the C++ programs built on it.

  * THE N > 1 HOST PIPELINE EXECUTES: `vp_feeds --gpus 2 --same-device` -- two driver threads, two contexts, two detector sets, the step rendezvous, the
    gather issued by whichever thread arrives last, slot reuse two steps later and the end-of-run drain (host/feeds_multi.cpp), with the all-gather done
    by the TEST-ONLY loop-back constructor of include/vp_multi.h (both ranks on device 0, device-to-device copies; real RCCL needs two devices).  Both
    ranks' packets / records are held to the CPU oracle for feed_seed(0, .) and feed_seed(1, .).  Replaces one process per camera, main.cpp:262-370 +
    udpsocket.cpp:94-100, 147-163.
"""
import json
import os
import subprocess

import numpy as np
import pytest

import cases
import oracle as O
from test_gpu_round3 import check_gathered_packets

pytestmark = pytest.mark.gpu
cfgs = cases.cfgs
CFG = cfgs.CFG2
EXE = os.path.join(cases.ROOT, "vision-processor_amd", "host", "vp_feeds")


def run_feeds(*args, timeout=900):
    envv = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([EXE, *[str(a) for a in args]], capture_output=True, text=True, timeout=timeout, env=envv)
    assert out.returncode == 0, out.stdout + out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1]), out.stderr


@pytest.mark.parametrize("local", [False, True])
def test_two_ranks_on_one_device_gather_the_oracles_detection_packets(tmp_path, local):
    """`vp_feeds --gpus 2 --same-device --detect --verify --dump`: 1 warm-up + 4 timed steps, so every slot of the double-buffered gather is reused
    twice and the last step's CPU stage runs with no GPU step behind it.  --verify: both ranks received both blocks byte for byte; the dump of rank 1's
    receive area parses (google.protobuf) to the oracle's detections for BOTH ranks' feeds."""
    feeds, steps, warmup = 3, 4, 1
    dump = str(tmp_path / "gathered.bin")
    r, _ = run_feeds("--gpus", 2, "--same-device", feeds, steps, warmup, "--detect", "--threads", 8, "--verify", "--dump", dump, *(["--local"] if local else []))
    assert r["gpus"] == 2 and r["same_device"] is True and "loop-back" in r["gather"] and "RCCL" not in r["gather"].replace("no RCCL", "")
    assert r["gather_verified"] is True and r["block"] == "SSL_WrapperPacket slots" and r["sat_variant"] == ("local" if local else "tiled")
    assert r["detect"]["robots_per_frame"] == 16.0 and r["detect"]["steps_reread_at_full_capacity"] == 0
    check_gathered_packets(open(dump, "rb").read(), 2, feeds, r["detect"]["packet_slot_bytes"], warmup + steps)


@pytest.mark.parametrize("ranks,feeds", [(2, 3), (4, 2)])
def test_two_ranks_on_one_device_gather_the_oracles_records(tmp_path, ranks, feeds):
    """The same pipeline without the CPU stage, with two and with four ranks (driver threads) on the one device: every rank's compact record block
    (vp_records_compact straight into the gather block) arrives at every rank; the last rank's receive area holds, for every feed of the first two
    ranks, the oracle's counters and raster-sorted records + pixel indices, and the program's --verify covers all ranks byte for byte."""
    f = cases.vp.feeds
    steps, warmup, cap = 5, 2, 192
    dump = str(tmp_path / "records.bin")
    r, _ = run_feeds("--gpus", ranks, "--same-device", feeds, steps, warmup, "--verify", "--dump", dump, "--cap", cap)
    assert r["gpus"] == ranks and r["gather_verified"] is True and r["block"] == "compact records" and r["truncated_slots_last_step"] == 0 and r["same_device"] is True
    blob = np.frombuffer(open(dump, "rb").read(), np.uint8)
    block = r["record_bytes_per_gpu"]
    assert blob.size == ranks * block
    slot = f.compact_slot_bytes(cap)
    off_i = (16 + 22 * cap + 3) & ~3
    cfg, mm = CFG, CFG.max_matches
    big = O.Port(threads=min(16, os.cpu_count() or 1))
    model_o = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    for g in range(2):
        for i in range(feeds):
            m, idx, cnt = big.frame(cfg.synth(f.feed_seed(g, i)), cfg.layout, cfg.qw, cfg.qh, model_o, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y,
                                    cfg.sampler_mode, cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm)
            s = blob[g * block + i * slot:g * block + (i + 1) * slot]
            head = s[:16].view(np.int32)
            n = int(cnt[0])
            assert head[:3].tolist() == [int(v) for v in cnt] and head[3] == n and r["counters"][g][i] == head[:3].tolist(), (g, i)
            assert s[16:16 + 22 * n].tobytes() == m.tobytes() and np.array_equal(s[off_i:off_i + 4 * n].view(np.uint32), idx), (g, i)


def test_a_frame_with_more_blobs_than_the_slot_is_reread_not_dropped(tmp_path):
    """--cap 8 (every cfg2 frame has ~130 blobs): the compact slots cannot hold what the host loop reads (main.cpp:301: min(counter[0], maxBlobs)
    records).  With --detect every step is read once more at full capacity and the packets are still the oracle's; without --detect the truncated
    slots are counted in the report."""
    feeds, steps, warmup = 2, 2, 1
    dump = str(tmp_path / "gathered.bin")
    r, _ = run_feeds("--gpus", 2, "--same-device", feeds, steps, warmup, "--detect", "--threads", 4, "--verify", "--dump", dump, "--cap", 8)
    assert r["gather_verified"] is True and r["detect"]["steps_reread_at_full_capacity"] == 2 * (steps + warmup) and r["detect"]["robots_per_frame"] == 16.0
    check_gathered_packets(open(dump, "rb").read(), 2, feeds, r["detect"]["packet_slot_bytes"], warmup + steps, oracle_ranks=1)
    r2, err = run_feeds("--gpus", 2, "--same-device", feeds, steps, warmup, "--verify", "--cap", 8)
    assert r2["truncated_slots_last_step"] == 2 * feeds and "raise --cap" in err


# ---- vp_ctx_tune "frame_graph": a refused capture must leave no host state behind ---------------------------------------------------------------
def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return cases.vp.load_capi(), torch


def _no_walker(cfg, fs=1.5):
    """a geometry whose (circle radius, gradient offset) pair has no k_flat_walk instance (r = 14, o = 5): VP_SAT_LOCAL goes through the private table"""
    return cfg.with_(name=cfg.name + f"_fs{fs}", field_scale=fs, off_x=-cfg.fw / 2 * fs, off_y=-cfg.fh / 2 * fs, visible_len=cfg.fw * fs * 1.05)


GRAPH_SEQ = "ABABACAAABBBCCABCA"  # second sightings right after another shape (A B A), after another geometry (A C A), runs of replays, and mixes


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("variant", ["tiled", "local", "local_private_table", "twopass"])
def test_frame_graph_alternating_launch_shapes_on_one_stream(env, port, variant, mode):
    """frame_graph = 1 on ONE stream, three (parameters, batch) tuples taking turns: A = 640x384 x 2 frames, B = 704x448 x 3 frames (another tile grid AND
    batch: the single-pass SAT's ticket counters, tile order and carry granules must be reset whenever the shape changes), C = A's shape with a shifted
    geometry (the stream's coordinate map must be rebuilt).  mode 1: a tuple is captured only when it follows itself (pre-flight: the stream's state is
    the tuple's own); mode 2: every repeated tuple attempts a capture, a capture that meets either need is refused, and the refusal must leave no trace --
    the call that follows it on the ordinary path has to see the stream as it was (round 3 left sat_shape / sat_tickets / map_key behind: the retry then
    skipped the reset and the map).  Every call's flat, table, counters and sorted records equal the CPU oracle's; vp_sync never reports a look-back timeout.
    "local_private_table": VP_SAT_LOCAL for an (r, o) pair without a walker uses the same single-pass state through a private table.  main.cpp:262-289."""
    capi, torch = env
    base = cfgs.T_MID
    if variant == "local_private_table":
        base = _no_walker(base)
        assert (base.circle_radius, base.grad_offset) == (14, 5)
    big = base.with_(name=base.name + "_704x448", fw=704, fh=448)
    shifted = base.with_(name=base.name + "_shifted", off_x=base.off_x + 100.0, off_y=base.off_y + 50.0)
    tuples = {"A": (base, 2), "B": (big, 3), "C": (shifted, 2)}
    sv = {"tiled": capi.SAT_TILED, "twopass": capi.SAT_TWOPASS}.get(variant, capi.SAT_LOCAL)
    with_sat = sv != capi.SAT_LOCAL
    ctx = capi.Context(0, n_streams=1)
    try:
        ctx.tune("frame_graph", mode)
        bufs, params, models = {}, {}, {}
        for k, (c, B) in tuples.items():
            mm = c.max_matches
            bufs[k] = dict(raw=torch.zeros((B,) + c.sensor_shape, dtype=torch.uint8, device="cuda"), flat=torch.zeros((B, c.fh, c.fw, 4), dtype=torch.uint8, device="cuda"),
                           sat=torch.zeros((B, c.fh, c.fw), dtype=torch.int32, device="cuda") if with_sat else None, m=torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"),
                           idx=torch.zeros((B, mm), dtype=torch.int32, device="cuda"), cnt=torch.full((B, 3), -5, dtype=torch.int32, device="cuda"))
            params[k] = capi.frame_params(c, sat_variant=sv, order=capi.ORDER_RASTER)
            models[k] = O.CameraModel.from_array(np.frombuffer(bytes(c.model()), np.uint8))
        for rnd, k in enumerate(GRAPH_SEQ):
            cfg, B = tuples[k]
            mm = cfg.max_matches
            raws = [cfg.synth(9000 + 10 * rnd + j) for j in range(B)]
            b = bufs[k]
            b["raw"].copy_(torch.from_numpy(np.stack(raws)))
            torch.cuda.synchronize()
            ctx.frame(params[k], b["raw"], B, b["flat"], b["sat"], b["m"], b["cnt"], idx=b["idx"])
            ctx.sync()  # raises on VP_ERR_TIMEOUT
            for j in range(B):
                s = port.frame_scratch(cfg.qw, cfg.qh, cfg.fw, cfg.fh, mm)
                wm, widx, wcnt = port.frame(raws[j], cfg.layout, cfg.qw, cfg.qh, models[k], cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                                            cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm, scratch=s)
                assert np.array_equal(b["flat"][j].cpu().numpy(), s["flat"]), (rnd, k, j, "flat")
                if with_sat:
                    assert np.array_equal(b["sat"][j].cpu().numpy(), port.sat_i32(s["graddot"])[0]), (rnd, k, j, "sat")
                cnt = b["cnt"][j].cpu().numpy()
                assert np.array_equal(cnt, wcnt), (rnd, k, j, cnt, wcnt)
                n = min(int(cnt[0]), mm)
                assert np.array_equal(b["idx"][j].cpu().numpy().view(np.uint32)[:n], widx) and b["m"][j].cpu().numpy()[:22 * n].tobytes() == wm.tobytes(), (rnd, k, j)
        stats = ctx.graph_stats()
        # the sequence really exercised the outcomes: captures, replays, graphs dropped because another tuple had used the stream -- and, without the
        # pre-flight, captures refused and undone
        assert stats["captures"] >= 2 and stats["replays"] >= 2 and stats["dropped"] >= 2 and (stats["refused"] >= 3 if mode == 2 else stats["refused"] == 0), stats
    finally:
        ctx.close()


# ---- BASELINE configs[4] at the launch profiles/ quotes: 4096x3072 x 32 frames, LDS-tiled vs global-recursive SAT -------------------------------------
CFG5 = cfgs.CFG5
B5 = 32
ORACLE5 = (0, 15, 16, 31)


def test_config4_high_res_launch_tiled_twopass_local_against_the_oracle(env):
    """cfg5 (4096x3072 RGGB, o = 3, r = 9, R = 8) x 32 frames through vp_frame_buffers_create, the launch the tiled-vs-global-recursive comparison of
    BASELINE configs[4] is quoted on (bench.py extras.cfg5_sat_variants, profiles/): VP_SAT_TILED, VP_SAT_TWOPASS and VP_SAT_LOCAL.  Frames 0, 15, 16, 31
    byte for byte against the CPU oracle (flat, int32 SAT, counters, raster-sorted records); all 32 through the size-independent property SAT == 64-bit
    cumulative sums of gradientDot of the kernel's own flat in plain torch integer arithmetic; the three variants' records equal each other on all 32.
    Replaces satHorizontal.cl:22-32 + satVertical.cl:22-32 at the size SURVEY section 5 "long-context" names."""
    from concurrent.futures import ThreadPoolExecutor
    from test_gpu_round3 import torch_sat_reference
    capi, torch = env
    cfg, B, mm = CFG5, B5, CFG5.max_matches
    with ThreadPoolExecutor(16) as ex:
        raws = np.stack(list(ex.map(cfg.synth, [cases.vp.feeds.feed_seed(5, i) for i in range(B)])))
    big = O.Port(threads=min(16, os.cpu_count() or 1))
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    want = {}
    for k in ORACLE5:
        s = big.frame_scratch(cfg.qw, cfg.qh, cfg.fw, cfg.fh, mm)
        m, idx, cnt = big.frame(raws[k], cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode,
                                cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm, scratch=s)
        sat, mx = big.sat_i32(s["graddot"])
        assert mx < 2 ** 24 and len(m) >= 80
        want[k] = dict(flat=s["flat"].copy(), sat=sat, m=m, idx=idx, cnt=cnt)
    ctx = capi.Context(0, n_streams=1)
    try:
        d_raw = dev(torch, raws)
        results = {}
        for name, sv in (("tiled", capi.SAT_TILED), ("twopass", capi.SAT_TWOPASS), ("local", capi.SAT_LOCAL)):
            p = capi.frame_params(cfg, sat_variant=sv, order=capi.ORDER_RASTER)
            slab = ctx.frame_slab(p, B)
            for _ in range(2):
                ctx.frame_buffers(p, d_raw, B, slab.b)
            ctx.sync()
            flat, sat, m_t, idx_t, cnt_t = slab.torch_views()
            cnt = cnt_t.cpu().numpy()
            for k in ORACLE5:
                w_ = want[k]
                assert torch.equal(flat[k].cpu(), torch.from_numpy(w_["flat"])), (name, k, "flat")
                if sat is not None:
                    assert torch.equal(sat[k].cpu(), torch.from_numpy(w_["sat"])), (name, k, "sat")
                assert np.array_equal(cnt[k], w_["cnt"]), (name, k, cnt[k], w_["cnt"])
                n = min(int(cnt[k, 0]), mm)
                assert np.array_equal(idx_t[k].cpu().numpy().view(np.uint32)[:n], w_["idx"]) and m_t[k].cpu().numpy()[:22 * n].tobytes() == w_["m"].tobytes(), (name, k, "records")
            if sat is not None:
                for k0 in range(0, B, 4):
                    assert torch.equal(sat[k0:k0 + 4], torch_sat_reference(torch, flat[k0:k0 + 4], cfg.grad_offset)), (name, "SAT property", k0)
            results[name] = (cnt.copy(), m_t.cpu().numpy().copy(), idx_t.cpu().numpy().copy())
            del flat, sat, m_t, idx_t, cnt_t
            slab.destroy()
        ct, mt, it = results["tiled"]
        for name in ("twopass", "local"):
            c2, m2, i2 = results[name]
            assert np.array_equal(c2, ct), name
            for k in range(B):
                n = min(int(ct[k, 0]), mm)
                assert np.array_equal(i2[k, :n], it[k, :n]) and m2[k, :22 * n].tobytes() == mt[k, :22 * n].tobytes(), (name, k)
    finally:
        ctx.close()


# ---- the walkers' survivor queues: records inside the walker, by the second launch, and every overflow path between them -------------------------------
@pytest.mark.parametrize("emit_kernel,cand_cap", [(0, 0), (1, 0), (1, 64), (1, 8), (1, 7)])
@pytest.mark.parametrize("walker", ["table", "local"])
def test_walker_survivor_queue_paths(env, port, walker, emit_kernel, cand_cap):
    """k_circle_walk / k_flat_walk queue their survivors in LDS and make records after the row loop: inside the walker (tail_emit_kernel 0) or, through
    the frames' queues in global memory, by k_emit_queue (1).  Inputs that push every limit: a frame in which EVERY pixel is a plateau peak (a wavefront
    finds 58 survivors per row: the LDS queue stops the walk after every row and the wavefront walks again from the next one; the frame's queue --
    candidate_capacity / 8 entries: default, 8, 1 and 0 -- overflows and the rest is drained inside the walker; counter[0] runs far past max_matches),
    the same frame beside ordinary ones in one launch, and noise with many rings.  Counters, blob sets and (RASTER) order equal the oracle's
    gradientDot -> SAT -> circle -> matches.  blobList.cl:36-103, main.cpp:283-317."""
    capi, torch = env
    rng = np.random.default_rng(4242)
    h, w, B = 150, 200, 4
    o, r, R = (2, 5, 4)
    flat = np.zeros((B, h, w, 4), np.uint8)
    flat[..., 3] = 255
    flat[0, ..., :3] = (90, 90, 90)                                  # constant: gradDot == 0, every pixel a plateau peak at threshold 0
    flat[1, ..., :3] = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)  # noise
    flat[2, ..., :3] = (100, 140, 120)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(60):                                              # many small discs, some clustered in one 52-column strip
        cy, cx = int(rng.integers(0, h)), int(rng.integers(20, 70))
        flat[2, (yy - cy) ** 2 + (xx - cx) ** 2 <= 16, :3] = rng.integers(0, 256, 3, dtype=np.uint8)
    flat[3] = flat[0]
    flat[3, 40:90, 50:150, :3] = rng.integers(80, 100, (50, 100, 3), dtype=np.uint8)
    for thr, mm in ((0.0, 50), (0.0, 40000), (15.0, 2000)):
        want = []
        for i in range(B):
            gd = port.graddot(flat[i], o)
            sat, _ = port.sat_i32(gd)
            bc = port.blobcenter_i32(sat, r)
            want.append(port.bloblist(flat[i], bc, thr, 0.0, R, mm))
        ctx = capi.Context(0, n_streams=1)
        try:
            ctx.tune("tail_emit_kernel", emit_kernel)
            if cand_cap:
                ctx.tune("candidate_capacity", cand_cap)
            d_flat = dev(torch, flat)
            d_sat = torch.zeros((B, h, w), dtype=torch.int32, device="cuda")
            if walker == "table":
                ctx.graddot_sat(d_flat, w, h, o, B, capi.SAT_TILED, d_sat)
            for order in (capi.ORDER_RASTER, capi.ORDER_ATOMIC):
                d_m = torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda")
                d_idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda")
                d_cnt = torch.full((B, 3), -3, dtype=torch.int32, device="cuda")
                for _ in range(2):  # twice: the self-cleaning queue counters must be back at zero
                    if walker == "table":
                        ctx.blobcenter_list(d_flat, d_sat, w, h, B, r, thr, 0.0, R, mm, order, d_m, d_idx, d_cnt)
                    else:
                        ctx.flat_bloblist(d_flat, w, h, B, o, r, thr, 0.0, R, mm, order, d_m, d_idx, d_cnt)
                ctx.sync()
                for i in range(B):
                    wm, widx, wcnt = want[i]
                    cnt = d_cnt[i].cpu().numpy()
                    assert np.array_equal(cnt, wcnt), (walker, thr, mm, order, i, cnt, wcnt)
                    gm, gi = capi.sorted_blobs(d_m[i].cpu().numpy(), d_idx[i].cpu().numpy().view(np.uint32), cnt, mm)
                    if int(cnt[0]) <= mm:
                        assert np.array_equal(gi, widx) and gm.tobytes() == wm.tobytes(), (walker, thr, mm, order, i)
                        if order == capi.ORDER_RASTER:
                            assert np.array_equal(d_idx[i].cpu().numpy().view(np.uint32)[:len(widx)], widx)
                    else:  # list overflow: which records survive is run-dependent in the reference too (blobList.cl:87-89); every survivor is genuine and unique
                        assert len(gm) == mm and len(np.unique(gi)) == mm
            if thr == 0.0:
                assert int(want[0][2][0]) == h * w  # the constant frame: every pixel was appended
        finally:
            ctx.close()
