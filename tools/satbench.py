"""A/B timing of the gradientDot+SAT pass (tools only).  usage: satbench.py BATCH [debug values ...]
Variants (VP_SAT_DEBUG values, see vp_sat_tiled.hip) are interleaved round-robin in ONE process (cdna guide rule 24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL[os.environ.get("VP_CFG", "cfg2_2048x1536_rggb")]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
variants = [int(v) for v in sys.argv[2:]] or [0]
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
flat = torch.empty((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device='cuda')
raw = torch.from_numpy(np.stack([cfg.synth(i) for i in range(min(B, 4))])).cuda()
m = cfg.model()
for i in range(B):
    ctx.resample_raw(raw[i % raw.shape[0]], cfg.layout, cfg.qw, cfg.qh, m, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh, 1, flat[i])
sat = torch.empty((B, cfg.fh, cfg.fw), dtype=torch.int32, device='cuda')
USE_STRIPS = bool(int(os.environ.get('VP_STRIPS', '1')))
sb = ctx.halo_strip_bytes(cfg.fw, cfg.fh, cfg.grad_offset)
strips = torch.zeros((B, max(sb, 4) // 4), dtype=torch.int32, device='cuda')
if sb: ctx.halo_strips(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, strips)
res = {v: [] for v in variants}
res['twopass'] = []
def run(variant, N=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        if USE_STRIPS and sb and variant == 0: ctx.graddot_sat_strips(flat, strips, cfg.fw, cfg.fh, cfg.grad_offset, B, variant, sat)
        else: ctx.graddot_sat(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, variant, sat)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1000
for rnd in range(6):
    for v in variants:
        if v: os.environ['VP_SAT_DEBUG'] = str(v)
        else: os.environ.pop('VP_SAT_DEBUG', None)  # 0 = the shipped path (k_sat_fast2): any VP_SAT_DEBUG value selects the instrumented first-generation kernel
        t = run(0, 3) if rnd == 0 else run(0)
        if rnd: res[v].append(t)
os.environ.pop('VP_SAT_DEBUG', None)
if os.environ.get("VP_TWOPASS"):
    for _ in range(3): res['twopass'].append(run(1))
for v, ts in res.items():
    if not ts: continue
    med = float(np.median(ts))
    print(f"{'tiled' if v != 'twopass' else 'twopass'} strips={int(USE_STRIPS and sb > 0)} {cfg.name} B={B} debug={v}: median {med:.1f} us/launch (min {min(ts):.1f}), {med/B:.2f} us/frame, {8*cfg.n_flat*B/med/1e3:.0f} GB/s algorithmic")
