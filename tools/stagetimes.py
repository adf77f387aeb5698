"""Per-stage times of vp_frame (the library's own hipEvent list) under vp_ctx_tune settings (tools only; GPU).
usage: stagetimes.py BATCH [key=value ...]   each key=value is one run; 'default' runs untouched"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL[os.environ.get("VP_CFG", "cfg2_2048x1536_rggb")]
B = int(sys.argv[1]); mm = cfg.max_matches
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
raw = raw4.repeat((B + 3) // 4, 1, 1)[:B].contiguous()
p = capi.frame_params(cfg)
_ctx0 = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
slab = _ctx0.frame_slab(p, B)  # the library's allocator (placement-checked blocks for large launch groups)
flat, sat, m, idx, c = slab.torch_views()
for setting in sys.argv[2:] or ["default"]:
    ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
    if setting != "default":
        for kv in setting.split(","):
            ctx.tune(kv.split("=")[0], int(kv.split("=")[1]))
    for _ in range(3): ctx.frame(p, raw, B, flat, sat, m, c, idx=idx)
    ctx.sync(); ctx.events_clear(); ctx.events_enable(True)
    for _ in range(20): ctx.frame(p, raw, B, flat, sat, m, c, idx=idx)
    ctx.sync(); ctx.events_enable(False)
    acc = {}
    for name, ms in ctx.event_times(): acc.setdefault(name, []).append(ms)
    print(f"{setting:28s} " + "  ".join(f"{k}: {np.median(v) * 1e3:7.1f} us" for k, v in acc.items()) + f"   sum {sum(np.median(v) for v in acc.values()) * 1e3:7.1f} us", flush=True)
    ctx.close() if hasattr(ctx, "close") else None
