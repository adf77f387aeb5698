"""Context-to-context spread of vp_frame at the benchmark launch: ten contexts with identical settings (sat_workgroups_per_cu 5; the even ones with
"placement_candidates" 0, which only matters to builds that place their workspaces) in one process on one set of torch buffers, interleaved.  What differs
between them is where their internal workspaces landed (profiles/r05_workspace_block_class.txt).  Edit the tuple below to compare values of the knob
instead.  GPU; tools only.  usage: sat_wpc.py [frames=256]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = vp.configs.ALL["cfg2_2048x1536_rggb"]; mm = cfg.max_matches
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
raw = raw4.repeat((B + 3) // 4, 1, 1)[:B].contiguous()
d = dict(flat=torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
         matches=torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"), idx=torch.zeros((B, mm), dtype=torch.int32, device="cuda"), counter=torch.zeros((B, 3), dtype=torch.int32, device="cuda"))
runs = []
for wpc in (5, 5, 5, 5, 5, 5, 5, 5, 5, 5):
    ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
    ctx.tune("sat_workgroups_per_cu", wpc)
    p = capi.frame_params(cfg)
    ctx.frame(p, raw, B, **d); ctx.sync()
    runs.append((len(runs), ctx, p))
res = {w: [] for w, _, _ in runs}
for rnd in range(8):
    for wpc, ctx, p in runs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ctx.frame(p, raw, B, **d)
        e1.record(); torch.cuda.synchronize()
        res[wpc].append(e0.elapsed_time(e1) * 100)
# per-stage times of each context (the library's own stage events), after the interleaved whole-frame timing
stage = {}
for wpc, ctx, p in runs:
    ctx.events_enable(True); ctx.events_clear()
    for _ in range(6): ctx.frame(p, raw, B, **d)
    ctx.sync()
    acc = {}
    for name, ms in ctx.event_times(): acc.setdefault(name, []).append(ms)
    stage[wpc] = {k: round(float(np.median(v)) * 1000, 1) for k, v in acc.items()}
    ctx.events_enable(False)
for wpc in res: print(f"   stages of context {wpc}: {stage[wpc]}")
for wpc in res: print(f"context {wpc}: vp_frame median {np.median(res[wpc]):8.1f} us per {B} frames (min {min(res[wpc]):.1f})")
