"""RASTER-order records must be identical launch after launch (the last workgroup of a frame sorts what other CUs wrote in the same launch with
write-through stores and reads with sc1 loads, no fence): repeat vp_frame at full size and compare every launch with the first (tools only; GPU).
usage: sortstress.py [LAUNCHES] [BATCH]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL[os.environ.get("VP_CFG", "cfg2_2048x1536_rggb")]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300; B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mm = cfg.max_matches
raw = torch.from_numpy(np.stack([cfg.synth(7000 + i) for i in range(B)])).cuda()
ctx = capi.Context(0, n_streams=2)
bad = 0
for variant in (capi.SAT_TILED, capi.SAT_LOCAL):
    p = capi.frame_params(cfg, sat_variant=variant, order=capi.ORDER_RASTER)
    flat = torch.empty((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"); sat = torch.empty((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda") if variant == capi.SAT_TILED else None
    m = torch.zeros((B, 22 * mm), dtype=torch.uint8, device="cuda"); idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda"); c = torch.zeros((B, 3), dtype=torch.int32, device="cuda")
    ref = None
    for it in range(N):
        m.fill_(0xEE); idx.fill_(-1)
        ctx.frame(p, raw, B, flat, sat, m, c, idx=idx, stream=it & 1); ctx.sync()
        cnt = c.cpu().numpy(); n = np.minimum(cnt[:, 0], mm)
        got = [(idx[i, :n[i]].cpu().numpy().copy(), m[i, :22 * n[i]].cpu().numpy().copy()) for i in range(B)]
        for i in range(B):
            assert np.all(np.diff(got[i][0].view(np.uint32).astype(np.int64)) > 0), f"variant {variant} launch {it} frame {i}: indices not strictly increasing"
        if ref is None: ref = (cnt.copy(), got); continue
        if not np.array_equal(cnt, ref[0]) or any(not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])) for a, b in zip(got, ref[1])):
            bad += 1; print(f"variant {variant}: launch {it} differs from launch 0", flush=True)
    print(f"variant {variant}: {N} launches of {B} frames, {int(ref[0][:, 0].sum())} records per launch, mismatching launches so far: {bad}", flush=True)
sys.exit(1 if bad else 0)
