"""Small launches (BASELINE configs[2]; the regime main.cpp:262-317 runs in: one frame per loop iteration): single-frame latency, host-synchronised, and
8 streams x 8 frames, with vp_ctx_tune knobs from VP_TUNE="key=value,...".  MODE=lat|8x8|both (default both), EVENTS=1 adds the per-stage hipEvent times of
one-frame launches.  GPU; tools only."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.CFG2
mm = cfg.max_matches
mode = os.environ.get("MODE", "both")
raw = torch.from_numpy(np.stack([cfg.synth(i) for i in range(8)])).cuda()
d_raw = raw.repeat(8, 1, 1).contiguous()
ctx = capi.Context(0, n_streams=8)
for kv in filter(None, os.environ.get("VP_TUNE", "").split(",")):
    ctx.tune(kv.split("=")[0], int(kv.split("=")[1]))
p = capi.frame_params(cfg)
per = 8
bufs = [dict(flat=torch.empty((per, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.empty((per, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
             matches=torch.zeros((per, 22 * mm), dtype=torch.uint8, device="cuda"), idx=torch.zeros((per, mm), dtype=torch.int32, device="cuda"),
             counter=torch.zeros((per, 3), dtype=torch.int32, device="cuda")) for _ in range(8)]
torch.cuda.synchronize()
out = {"tune": os.environ.get("VP_TUNE", "")}
u = bufs[0]
if mode in ("lat", "both"):
    for _ in range(20): ctx.frame(p, d_raw, 1, stream=0, **u); ctx.sync(0)
    lat = []
    for rnd in range(5):
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.frame(p, d_raw, 1, stream=0, **u); ctx.sync(0)
        lat.append(round(1e6 * (time.perf_counter() - t0) / n, 1))
    out["single_frame_latency_us"] = lat
    if os.environ.get("EVENTS"):
        ctx.events_clear(); ctx.events_enable(True)
        for _ in range(50):
            ctx.frame(p, d_raw, 1, stream=0, **u); ctx.sync(0)
        ctx.events_enable(False)
        st = {}
        for name, ms in ctx.event_times(): st.setdefault(name, []).append(ms)
        ctx.events_clear()
        out["stage_us"] = {k: round(1e3 * float(np.median(v)), 1) for k, v in st.items()}
if mode in ("8x8", "both"):
    def go(ns, b):
        for s in range(ns): ctx.frame(p, d_raw[s * b:], b, stream=s, **bufs[s])
    res = {}
    for ns, b in ((1, 1), (8, 1), (1, 8), (8, 8)):
        for _ in range(5): go(ns, b)
        ctx.sync()
        r = []
        for rnd in range(3):
            n = 100 if b == 1 else 40
            t0 = time.perf_counter()
            for _ in range(n): go(ns, b)
            ctx.sync()
            r.append(round(ns * b * n / (time.perf_counter() - t0)))
        res[f"{ns}x{b}"] = r
    out["frames_per_s_streams_x_frames"] = res
print(json.dumps(out), flush=True)
ctx.close()
