"""Timing of the fused tail (circle+detect, emit + sort) on real SAT tables, swept over a vp_ctx_tune knob, interleaved in one process (tools only).
usage: tailbench.py BATCH [KEY=tail_segment_rows] [VALUE ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL[os.environ.get("VP_CFG", "cfg2_2048x1536_rggb")]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
key = sys.argv[2] if len(sys.argv) > 2 else "tail_segment_rows"
variants = [int(v) for v in sys.argv[3:]] or [0]
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
raw = raw4.repeat((B + 3) // 4, *([1] * (raw4.dim() - 1)))[:B].contiguous()
mm = cfg.max_matches
d = dict(flat=torch.zeros((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.zeros((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
         matches=torch.zeros((B, mm * 22), dtype=torch.uint8, device="cuda"), idx=torch.zeros((B, mm), dtype=torch.int32, device="cuda"),
         counter=torch.zeros((B, 3), dtype=torch.int32, device="cuda"))
ORDER = capi.ORDER_ATOMIC if os.environ.get("VP_TB_ORDER") == "atomic" else capi.ORDER_RASTER  # VP_TB_ORDER=atomic: no sort by the last workgroup
p = capi.frame_params(cfg, sat_variant=capi.SAT_TILED, order=ORDER)
ctx.frame(p, raw, B, **d); ctx.sync()
res = {v: [] for v in variants}
def run(N=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        ctx.blobcenter_list(d["flat"], d["sat"], cfg.fw, cfg.fh, B, cfg.circle_radius, cfg.circ_threshold, cfg.min_score, cfg.list_radius, mm, ORDER,
                            d["matches"], d["idx"], d["counter"])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1000
for rnd in range(int(os.environ.get("VP_TB_ROUNDS", "6"))):
    for v in variants:
        ctx.tune(key, v)
        t = run(3) if rnd == 0 else run()
        if rnd: res[v].append(t)
for v, ts in res.items():
    med = float(np.median(ts))
    print(f"tail {cfg.name} B={B} {key}={v}: median {med:.1f} us/launch (min {min(ts):.1f}), {med/B:.2f} us/frame", flush=True)
