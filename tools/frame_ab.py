"""A/B of an environment switch on the whole vp_frame path in ONE process on ONE set of buffers (tools only).
usage: frame_ab.py ENVVAR VALUE_A VALUE_B [BATCH]   -> per-stage and whole-step times, interleaved rounds"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 128
cfg = vp.configs.ALL[os.environ.get("VP_CFG", "cfg2_2048x1536_rggb")]
mm = cfg.max_matches
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
params = capi.frame_params(cfg, sat_variant=capi.SAT_TILED, order=capi.ORDER_RASTER)
raws = np.stack([cfg.synth(i) for i in range(min(B, 8))])
d_raw = torch.from_numpy(np.concatenate([raws] * ((B + len(raws) - 1) // len(raws)))[:B]).cuda()
flat = torch.empty((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
sat = torch.empty((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda")
matches = torch.zeros((B, 22 * mm), dtype=torch.uint8, device="cuda")
idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda")
counter = torch.zeros((B, 3), dtype=torch.int32, device="cuda")
res = {va: [], vb: []}
stages = {va: {}, vb: {}}
for rnd in range(9):
    for v in (va, vb):
        os.environ[var] = v
        for _ in range(2): ctx.frame(params, d_raw, B, flat, sat, matches, counter, idx=idx)
        torch.cuda.synchronize()
        ctx.events_clear(); ctx.events_enable(True)
        t0 = time.perf_counter()
        for _ in range(10): ctx.frame(params, d_raw, B, flat, sat, matches, counter, idx=idx)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        ctx.events_enable(False)
        if rnd:
            res[v].append(dt * 1e3)
            for name, ms in ctx.event_times(): stages[v].setdefault(name, []).append(ms)
for v in (va, vb):
    print(f"{var}={v}: step {np.median(res[v]):.4f} ms ({B / np.median(res[v]) * 1e3:.0f} frames/s)  " + "  ".join(f"{k} {np.median(x):.4f}" for k, x in stages[v].items()))
