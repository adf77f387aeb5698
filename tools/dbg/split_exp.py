"""One launch group of N frames against the same frames split over s library streams of ONE context (slabs from vp_frame_buffers_create), with caps on the
SAT pass's resident workgroups: does the VALU-bound resampling of one part run beside the HBM-bound SAT / tail of another?  GPU; tools only."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.CFG2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
d_raw = raw4.repeat(N // 4, 1, 1).contiguous()
res = {}
for wpc in (0, 3, 4):
    for parts in (1, 2, 4):
        ctx = capi.Context(0, n_streams=max(parts, 1))
        ctx.tune("placement_max_mib", -1)
        if wpc: ctx.tune("sat_workgroups_per_cu", wpc)
        p = capi.frame_params(cfg)
        b = N // parts
        slabs = [ctx.frame_slab(p, b) for _ in range(parts)]
        def go():
            for s in range(parts):
                ctx.frame_buffers(p, d_raw[s * b:], b, slabs[s].b, stream=s)
        for _ in range(3): go()
        ctx.sync()
        ts = []
        for rnd in range(3):
            t0 = time.perf_counter()
            for _ in range(10): go()
            ctx.sync(); ts.append(N * 10 / (time.perf_counter() - t0))
        res[f"wpc{wpc}_{parts}x{b}"] = [round(t) for t in ts]
        print(f"wpc {wpc} {parts} x {b}: {[round(t) for t in ts]} frames/s  sat block {[s.placement['sat']['write_gbs'] for s in slabs]}", flush=True)
        for s in slabs: s.destroy()
        ctx.close()
print(json.dumps(res))
