"""Does splitting a batch over several streams overlap the three stage kernels?  (tools only)  usage: streambench.py TOTAL_BATCH"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.CFG2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mm = cfg.max_matches
NS = 4
ctx = capi.Context(0, n_streams=NS)
p = capi.frame_params(cfg)
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
raw = raw4.repeat((B + 3) // 4, 1, 1)[:B].contiguous()
d = dict(flat=torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
         matches=torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"), idx=torch.zeros((B, mm), dtype=torch.int32, device="cuda"),
         counter=torch.zeros((B, 3), dtype=torch.int32, device="cuda"))
torch.cuda.synchronize()
def step(ns):
    per = B // ns
    for s in range(ns):
        sl = slice(s * per, (s + 1) * per)
        ctx.frame(p, raw[sl], per, d["flat"][sl], d["sat"][sl], d["matches"][sl], d["counter"][sl], idx=d["idx"][sl], stream=s)
for ns in (1, 2, 4):
    for _ in range(3): step(ns)
    ctx.sync()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(10): step(ns)
        ctx.sync()
        ts.append((time.perf_counter() - t0) / 10)
    t = sorted(ts)[len(ts) // 2]
    print(f"{ns} stream(s) x {B // ns} frames: {t*1e3:.3f} ms per {B} frames = {B/t:.0f} frames/s")
