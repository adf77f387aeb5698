"""Whole-path stage times on flat sizes that are not multiples of 64 / 4 (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
B = 32
base = vp.configs.CFG2
raw4 = torch.from_numpy(np.stack([base.synth(i) for i in range(4)])).cuda()
raw = raw4.repeat(B // 4, 1, 1).contiguous()
for (fw, fh) in [(2048, 1536), (2040, 1530), (1994, 1472)]:
    cfg = base.with_(name=f"cfg_{fw}x{fh}", fw=fw, fh=fh)
    mm = cfg.max_matches
    d = dict(flat=torch.zeros((B, fh, fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.zeros((B, fh, fw), dtype=torch.int32, device="cuda"),
             matches=torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"), idx=torch.zeros((B, mm), dtype=torch.int32, device="cuda"),
             counter=torch.zeros((B, 3), dtype=torch.int32, device="cuda"))
    p = capi.frame_params(cfg)
    for _ in range(3): ctx.frame(p, raw, B, **d)
    ctx.sync(); ctx.events_clear(); ctx.events_enable(True)
    for _ in range(10): ctx.frame(p, raw, B, **d)
    ctx.sync(); ctx.events_enable(False)
    st = {}
    for name, ms in ctx.event_times(): st.setdefault(name, []).append(ms)
    print(fw, fh, {k: round(float(np.mean(v)) * 1e3) for k, v in st.items()}, "us per", B, "frames; blobs", int(d["counter"][0, 0]))
