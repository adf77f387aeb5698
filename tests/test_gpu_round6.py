"""GPU tests added in round 6, all through the C ABI.

  * vp_records_compact_signal / vp_wait_word / vp_flush: the read-back of main.cpp:297-301 without a copy command or a stream synchronisation on the way.
  * the library's one recovery path, executed: a look-back wait of the single-pass SAT that never ends -> bounded launch -> VP_ERR_TIMEOUT from vp_sync
    -> the stream's next launch resets the carry state and is correct again; other streams of the context never notice.
"""
import time

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


class Buffers:
    def __init__(self, torch, cfg, batch, mm):
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
        self.flat, self.sat = z((batch, cfg.fh, cfg.fw, 4), torch.uint8), z((batch, cfg.fh, cfg.fw), torch.int32)
        self.m, self.idx, self.cnt = z((batch, mm * 22), torch.uint8), z((batch, mm), torch.int32), z((batch, 3), torch.int32)

    def kw(self):
        return dict(flat=self.flat, sat=self.sat, matches=self.m, counter=self.cnt, idx=self.idx)


def check(capi, b, f, want, mm):
    wm, widx, wcnt = want
    cnt = b.cnt[f].cpu().numpy()
    assert np.array_equal(cnt, wcnt), (cnt, wcnt)
    gm, gi = capi.sorted_blobs(b.m[f].cpu().numpy(), b.idx[f].cpu().numpy().view(np.uint32), cnt, mm)
    assert np.array_equal(gi, widx) and gm.tobytes() == wm.tobytes()


def test_a_lookback_wait_that_never_ends_is_bounded_reported_and_recovered_from(env, port):
    """The only way a carry hand-off bug of the single-pass SAT (vp_sat_tiled.hip) would ever reach a caller, run once on purpose.

    vp_ctx_tune("sat_debug_drop_tile", t) makes the launch withhold the column carry of tile t -- neither the aggregate nor the inclusive granule is
    published -- so the tile below it polls until "sat_spin_limit" runs out (set small here; seconds by default), raises the device abort word and the
    pinned host flag, every other waiting wavefront sees the abort and leaves, and the launch DRAINS (a kernel that spun for ever would take the GPU
    with it).  Then, as opencl.cpp:87-92 turns a failed wait into FATAL: vp_sync returns VP_ERR_TIMEOUT with a message (the facade's VP_OR_FATAL exits
    on it); the table of that launch is wrong, as documented.  The caller's NEXT vp_frame on the stream finds the flag, resets the carry state
    (tickets, epochs, granules: vp_sat_tiled.hip `reset`) and returns the oracle's frame; a launch on another stream of the same context, issued
    between the two, never notices.  Three frames per launch so that frames beside the poisoned one are seen to be caught by the abort, too."""
    capi, torch = env
    cfg, mm, B = cfgs.T_MID, cfgs.T_MID.max_matches, 3
    raws = [cfg.synth(9100 + i) for i in range(B)]
    want = [oracle_frame(port, cfg, r, mm) for r in raws]
    assert min(int(w[2][0]) for w in want) >= 20
    d_raw = torch.from_numpy(np.stack(raws)).cuda()
    ctx = capi.Context(0, n_streams=2)
    try:
        ctx.tune("sat_spin_limit", 4000)
        p = capi.frame_params(cfg, sat_variant=capi.SAT_TILED, order=capi.ORDER_RASTER)
        b0, b1 = Buffers(torch, cfg, B, mm), Buffers(torch, cfg, B, mm)
        # a healthy launch first: the stream has live carry state (epoch 1) when the fault comes
        ctx.frame(p, d_raw, B, stream=0, **b0.kw())
        ctx.sync(0)
        for f in range(B): check(capi, b0, f, want[f], mm)
        good_sat = b0.sat.clone()

        tiles_x = (cfg.fw + 63) // 64
        ctx.tune("sat_debug_drop_tile", 1 * tiles_x * ((cfg.fh + 63) // 64) + 3)   # frame 1, tile row 0, tile column 3: the tiles below it starve
        ctx.frame(p, d_raw, B, stream=0, **b0.kw())
        ctx.tune("sat_debug_drop_tile", -1)
        ctx.frame(p, d_raw, B, stream=1, **b1.kw())                                    # another stream of the same context, same time
        t0 = time.perf_counter()
        ctx.sync(1)                                                                    # healthy: no error, the oracle's frames
        for f in range(B): check(capi, b1, f, want[f], mm)
        with pytest.raises(capi.VpError) as e:
            ctx.sync(0)
        waited = time.perf_counter() - t0
        assert e.value.code == -4 and "look-back" in str(e.value), str(e.value)         # VP_ERR_TIMEOUT (include/vp_hip.h)
        assert waited < 20.0, f"the poisoned launch took {waited:.1f} s to drain"
        assert not torch.equal(b0.sat[1], good_sat[1]), "the fault was not injected: the table of the poisoned frame is right"
        with pytest.raises(capi.VpError):                                               # the flag stays up until a launch has repaired the stream
            ctx.sync(0)
        ctx.sync(1)

        # recovery: the next call on the poisoned stream resets and is right again -- twice, so that the state after the reset is seen to be live
        for _ in range(2):
            b0.sat.zero_(); b0.cnt.fill_(77)
            ctx.frame(p, d_raw, B, stream=0, **b0.kw())
            ctx.sync(0)
            assert torch.equal(b0.sat, good_sat)
            for f in range(B): check(capi, b0, f, want[f], mm)
        ctx.sync()
    finally:
        ctx.close()


def test_records_reach_the_host_through_a_completion_word(env, port):
    """vp_records_compact_signal stores the compact slots straight into pinned host memory and its last workgroup then stores a step number into a completion
    word; the host learns that the records are readable by polling that word (vp_wait_word) -- no copy command, no vp_sync between the launch and the read.
    Three steps with different frames on one stream, each waited for by its own value; the slots must be byte for byte what vp_records_compact + a blocking copy
    deliver, and what the oracle says.  vp_flush (clFlush) is a no-op for results and must not fail on a busy stream; a word that never comes is a
    VP_ERR_TIMEOUT, not a hang."""
    import ctypes as C
    capi, torch = env
    cfg, mm, B, cap = cfgs.T_MID, cfgs.T_MID.max_matches, 5, 192
    ctx = capi.Context(0, n_streams=1)
    try:
        capi.lib().vp_record_layout_compact.restype = C.c_size_t
        slot, offm, offi = C.c_size_t(), C.c_size_t(), C.c_size_t()
        total = capi.lib().vp_record_layout_compact(B, cap, C.byref(slot), C.byref(offm), C.byref(offi))
        h_slots = ctx.malloc_host(total)
        h_word = ctx.malloc_host(64)
        word = (C.c_uint32 * 16).from_address(h_word)
        word[0] = 0
        host = np.ctypeslib.as_array((C.c_uint8 * total).from_address(h_slots))
        p = capi.frame_params(cfg, sat_variant=capi.SAT_TILED, order=capi.ORDER_RASTER)
        b = Buffers(torch, cfg, B, mm)
        d_slots = torch.zeros(total, dtype=torch.uint8, device="cuda")
        for step in range(1, 4):
            raws = [cfg.synth(9300 + 10 * step + i) for i in range(B)]
            d_raw = torch.from_numpy(np.stack(raws)).cuda()
            host[:] = 0xEE
            ctx.frame(p, d_raw, B, stream=0, **b.kw())
            ctx.records_compact_signal(b.cnt, b.m, b.idx, B, mm, cap, h_slots, h_word, step)
            ctx.flush(0)
            ctx.wait_word(h_word, step, 20.0)          # no vp_sync: the word is the only thing the host waits for
            got = host.copy()
            ctx.sync(0)
            ctx.records_compact(b.cnt, b.m, b.idx, B, mm, cap, d_slots)
            ctx.sync(0)
            want = d_slots.cpu().numpy()
            for f in range(B):
                wm, widx, wcnt = oracle_frame(port, cfg, raws[f], mm)
                s0 = f * slot.value
                head = got[s0:s0 + 16].view(np.int32)
                n = min(int(wcnt[0]), mm, cap)
                assert head[:3].tolist() == wcnt.tolist() and head[3] == n and n >= 20
                # the bytes of the slot that are defined (header, n records, n indices) equal the device-block form
                assert np.array_equal(got[s0:s0 + 16 + 22 * n], want[s0:s0 + 16 + 22 * n])
                assert np.array_equal(got[s0 + offi.value:s0 + offi.value + 4 * n], want[s0 + offi.value:s0 + offi.value + 4 * n])
                assert got[s0 + 16:s0 + 16 + 22 * n].tobytes() == wm[:n].tobytes() and np.array_equal(got[s0 + offi.value:s0 + offi.value + 4 * n].view(np.uint32), widx[:n])
        with pytest.raises(capi.VpError) as e:
            ctx.wait_word(h_word, 99, 0.05)
        assert e.value.code == -4
        ctx.free_host(h_slots)
        ctx.free_host(h_word)
    finally:
        ctx.close()


def test_completion_words_of_two_streams_and_truncated_slots(env, port):
    """Two library streams signal into two words at once (each stream has its own arrival counter), a single-frame launch signals too (one workgroup is the last
    one), and a slot too small for a frame's blobs says so (n < min(counter[0], max_matches)) exactly as vp_records_compact does."""
    import ctypes as C
    capi, torch = env
    cfg, mm, cap = cfgs.T_MID, cfgs.T_MID.max_matches, 16
    ctx = capi.Context(0, n_streams=2)
    try:
        capi.lib().vp_record_layout_compact.restype = C.c_size_t
        slot, offm, offi = C.c_size_t(), C.c_size_t(), C.c_size_t()
        B = [4, 1]
        totals = [capi.lib().vp_record_layout_compact(b, cap, C.byref(slot), C.byref(offm), C.byref(offi)) for b in B]
        h_slots = [ctx.malloc_host(t) for t in totals]
        h_word = ctx.malloc_host(256)
        words = (C.c_uint32 * 64).from_address(h_word)
        for i in range(64): words[i] = 0
        p = capi.frame_params(cfg, sat_variant=capi.SAT_TILED, order=capi.ORDER_RASTER)
        bufs = [Buffers(torch, cfg, b, mm) for b in B]
        raws = [[cfg.synth(9500 + 10 * s + i) for i in range(B[s])] for s in range(2)]
        d_raw = [torch.from_numpy(np.stack(r)).cuda() for r in raws]
        torch.cuda.synchronize()
        for step in (1, 2):
            for s in range(2):
                ctx.frame(p, d_raw[s], B[s], stream=s, **bufs[s].kw())
                ctx.records_compact_signal(bufs[s].cnt, bufs[s].m, bufs[s].idx, B[s], mm, cap, h_slots[s], h_word + 64 * s, 100 * step + s, stream=s)
            for s in (1, 0):
                ctx.wait_word(h_word + 64 * s, 100 * step + s, 20.0)
                host = np.ctypeslib.as_array((C.c_uint8 * totals[s]).from_address(h_slots[s])).copy()
                for f in range(B[s]):
                    wm, widx, wcnt = oracle_frame(port, cfg, raws[s][f], mm)
                    head = host[f * slot.value:f * slot.value + 16].view(np.int32)
                    assert head[:3].tolist() == wcnt.tolist() and int(wcnt[0]) > cap
                    assert head[3] == cap                                                  # truncated: the slot holds cap of the counter[0] records
                    s0 = f * slot.value
                    assert host[s0 + 16:s0 + 16 + 22 * cap].tobytes() == wm[:cap].tobytes()
                    assert np.array_equal(host[s0 + offi.value:s0 + offi.value + 4 * cap].view(np.uint32), widx[:cap])
        ctx.sync()
        ctx.flush()
        for hs in h_slots: ctx.free_host(hs)
        ctx.free_host(h_word)
    finally:
        ctx.close()


def test_the_generic_tiled_kernel_times_out_and_recovers_too(env):
    """The same recovery path through the unfused entry (vp_sat on a float gradDot image: the generic k_sat_tiled, not the benchmark's k_sat_fast2), on a ragged
    shape, against 64-bit cumulative sums."""
    capi, torch = env
    w, h, B = 200, 150, 2      # 4 x 3 tiles, the last column and row partial
    rng = np.random.default_rng(66)
    g = rng.integers(-40000, 40000, (B, h, w)).astype(np.float32)
    want = np.cumsum(np.cumsum(g.astype(np.int64), axis=1), axis=2).astype(np.int32)
    d_g = torch.from_numpy(g).cuda()
    d_sat = torch.zeros((B, h, w), dtype=torch.int32, device="cuda")
    ctx = capi.Context(0, n_streams=1)
    try:
        ctx.tune("sat_spin_limit", 3000)
        ctx.sat(d_g, w, h, B, capi.SAT_TILED, d_sat)
        ctx.sync(0)
        assert np.array_equal(d_sat.cpu().numpy(), want)
        ctx.tune("sat_debug_drop_tile", 1)          # frame 0, tile row 0, tile column 1: (1, 1) and (2, 1) starve
        ctx.sat(d_g, w, h, B, capi.SAT_TILED, d_sat)
        ctx.tune("sat_debug_drop_tile", -1)
        with pytest.raises(capi.VpError) as e:
            ctx.sync(0)
        assert e.value.code == -4
        got = d_sat.cpu().numpy()
        assert not np.array_equal(got[0], want[0])  # the poisoned frame's table is wrong ...
        d_sat.zero_()
        ctx.sat(d_g, w, h, B, capi.SAT_TILED, d_sat)  # ... and the next launch on the stream resets the carry state
        ctx.sync(0)
        assert np.array_equal(d_sat.cpu().numpy(), want)
    finally:
        ctx.close()
