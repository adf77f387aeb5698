import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.CFG2; mm = cfg.max_matches
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
d_raw = raw4.repeat(16, 1, 1).contiguous()
per = 8
bufs = [dict(flat=torch.empty((per, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.empty((per, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
             m=torch.zeros((per, 22 * mm), dtype=torch.uint8, device="cuda"), i=torch.zeros((per, mm), dtype=torch.int32, device="cuda"), c=torch.zeros((per, 3), dtype=torch.int32, device="cuda")) for _ in range(8)]
for wpc in (0, 2, 3):
  for variant in ("tiled", "local"):
    ctx = capi.Context(0, n_streams=8)
    ctx.tune("sat_workgroups_per_cu", wpc)
    p = capi.frame_params(cfg, sat_variant=capi.SAT_TILED if variant == "tiled" else capi.SAT_LOCAL)
    def go(ns, b):
        for st in range(ns):
            u = bufs[st]
            ctx.frame(p, d_raw[st * b:], b, u["flat"], u["sat"] if variant == "tiled" else None, u["m"], u["c"], idx=u["i"], stream=st)
    out = {}
    for ns, b in ((1, 8), (2, 8), (4, 8), (8, 8), (8, 1), (4, 1)):
        for _ in range(3): go(ns, b)
        ctx.sync(); t0 = time.perf_counter()
        n = 30 if b == 8 else 100
        for _ in range(n): go(ns, b)
        ctx.sync(); out[f"{ns}x{b}"] = round(ns * b * n / (time.perf_counter() - t0))
    print(variant, "wpc", wpc, out, flush=True)
    ctx.close()
