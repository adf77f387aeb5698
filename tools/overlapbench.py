"""Does the hot path gain from running launch groups of different streams side by side (resampling is VALU/LDS-bound, the SAT pass HBM-bound)?
s streams x b frames per launch on each, every combination with several settings of resident SAT workgroups (tools only; GPU).
usage: overlapbench.py [TOTAL_FRAMES]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL[os.environ.get("VP_CFG", "cfg2_2048x1536_rggb")]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mm = cfg.max_matches
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
raw = raw4.repeat((T + 3) // 4, 1, 1)[:T].contiguous()
ctx = capi.Context(0, n_streams=8)
p = capi.frame_params(cfg)
flat = torch.empty((T, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"); sat = torch.empty((T, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda")
m = torch.zeros((T, 22 * mm), dtype=torch.uint8, device="cuda"); idx = torch.zeros((T, mm), dtype=torch.int32, device="cuda"); c = torch.zeros((T, 3), dtype=torch.int32, device="cuda")
def rate(fn, n):
    for _ in range(3): fn()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    ctx.sync(); return T * n / (time.perf_counter() - t0)
for wpc in (0, 2, 3, 4):
    ctx.tune("sat_workgroups_per_cu", wpc)
    row = []
    for ns in (1, 2, 4, 8):
        b = T // ns
        def go():
            for st in range(ns):
                o = st * b
                ctx.frame(p, raw[o:], b, flat[o:], sat[o:], m[o:], c[o:], idx=idx[o:], stream=st)
        row.append(f"{ns}x{b}: {rate(go, 20) / 1e3:6.1f} k")
    print(f"SAT workgroups/CU {wpc or 'auto'}:  " + "   ".join(row), flush=True)
