"""Randomised differential soak of the whole hot path against the CPU oracle (tools only; GPU).  For each seed a synthetic frame of
a small configuration (every layout, both sampler modes, ragged sizes, varying batch sizes and both list orders) is pushed through
vp_frame (all three SAT variants, the local-table one without a SAT buffer) and through oracle/vp_oracle.c; flat, SAT, counters and the sorted blob records must be byte-identical.
usage: parity_soak.py [SECONDS]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
from _vp_pkg import load
import oracle as O
vp = load(); capi = vp.load_capi(); cfgs = vp.configs
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
O.build(ref=False)
port = O.Port(threads=8)
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
gctx = capi.Context(0, n_streams=2)  # library-owned streams with vp_ctx_tune "frame_graph": ordinary call, capture, replay of the same call
gctx.tune("frame_graph", 1)
rng = np.random.default_rng(2024)
pool = [cfgs.T_MID, cfgs.T_RGGB, cfgs.T_GRBG, cfgs.T_BGR, cfgs.T_RAGGED, cfgs.T_RAGGED.with_(name="t_ragged_even", qw=68, qh=46), cfgs.CFG1]
t0, frames, launches, seed = time.time(), 0, 0, 10_000
t_note = t0
while time.time() - t0 < budget:
    if time.time() - t_note > 60:  # a line a minute: a silent GPU job is taken to be hung
        t_note = time.time()
        print(f"... {frames} frames in {launches} launches so far, all identical", flush=True)
    cfg = pool[int(rng.integers(len(pool)))]
    mode = int(rng.integers(2))
    B = int(rng.choice([1, 1, 2, 3, 5, 17]))
    order = capi.ORDER_RASTER if rng.integers(2) else capi.ORDER_ATOMIC
    variant = [capi.SAT_TILED, capi.SAT_TILED, capi.SAT_LOCAL, capi.SAT_LOCAL, capi.SAT_TWOPASS][int(rng.integers(5))]
    ctx.tune("tail_segment_rows", int(rng.choice([0, 0, 16, 32, 128])))
    ctx.tune("sat_workgroups_per_cu", int(rng.choice([0, 0, 2, 3, 6])))
    ctx.tune("tail_emit_kernel", int(rng.choice([1, 1, 1, 0])))
    mm = cfg.max_matches
    raws = [cfg.synth(seed + i) for i in range(B)]
    seed += B
    p = capi.frame_params(cfg, sat_variant=variant, order=order, sampler_mode=mode, use_coord_map=bool(rng.integers(2)))
    d_raw = torch.from_numpy(np.stack(raws)).cuda()
    flat = torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
    sat = torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda") if variant != capi.SAT_LOCAL else None
    matches = torch.zeros((B, 22 * mm), dtype=torch.uint8, device="cuda")
    idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda")
    counter = torch.zeros((B, 3), dtype=torch.int32, device="cuda")
    ctx.frame(p, d_raw, B, flat, sat, matches, counter, idx=idx)
    ctx.sync()
    launches += 1
    if rng.integers(3) == 0:  # the same call three times through the graph path on a library stream: the replay must leave the same bytes
        st = int(rng.integers(2))
        g = dict(flat=torch.zeros_like(flat), sat=torch.zeros_like(sat) if sat is not None else None, m=torch.zeros_like(matches), i=torch.zeros_like(idx), c=torch.zeros_like(counter))
        for rep in range(3):
            gctx.frame(p, d_raw, B, g["flat"], g["sat"], g["m"], g["c"], idx=g["i"], stream=st)
        gctx.sync()
        launches += 3
        assert torch.equal(g["flat"], flat) and (sat is None or torch.equal(g["sat"], sat)) and torch.equal(g["c"], counter), "graph replay differs (flat / SAT / counters)"
        if order == capi.ORDER_RASTER:
            for i in range(B):
                n = min(int(counter[i, 0]), mm)
                assert torch.equal(g["i"][i, :n], idx[i, :n]) and torch.equal(g["m"][i, :22 * n], matches[i, :22 * n]), "graph replay differs (records)"
    hf, hs, hm, hi, hc = flat.cpu().numpy(), (sat.cpu().numpy() if sat is not None else None), matches.cpu().numpy(), idx.cpu().numpy().view(np.uint32), counter.cpu().numpy()
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    for i in range(B):
        planes = port.raw2quad(raws[i], cfg.layout, cfg.qw, cfg.qh)
        wflat = port.resample(planes, cfg.layout, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, mode, cfg.fw, cfg.fh)
        wsat = port.sat_i32(port.graddot(wflat, cfg.grad_offset))[0]
        m, wi, wc = port.frame(raws[i], cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, mode, cfg.fw, cfg.fh,
                               cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm)
        tag = f"{cfg.name} mode {mode} batch {B} frame {i} seed {seed - B + i} order {order} sat {variant}"
        assert np.array_equal(hf[i], wflat), "flat: " + tag
        assert hs is None or np.array_equal(hs[i], wsat), "SAT: " + tag
        assert np.array_equal(hc[i], wc), "counter: " + tag
        gm, gi = capi.sorted_blobs(hm[i], hi[i], hc[i], mm)
        assert np.array_equal(gi, wi) and gm.tobytes() == m.tobytes(), "blob records: " + tag
        frames += 1
print(f"parity soak: {frames} frames in {launches} launches over {len(pool)} configurations, {time.time() - t0:.0f} s: flat, SAT, counters and blob records byte-identical to the CPU oracle")
