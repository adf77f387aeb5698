"""A/B timing of the coordinate-map resampling kernel over the values of one environment switch, interleaved in one process, with a
bit-for-bit comparison of the outputs (tools only).  usage: rsab.py BATCH VALUE [VALUE ...]   (switch: $VP_AB_VAR, default VP_RS_FLT;
e.g. VP_AB_VAR=VP_RS_XCD rsab.py 128 0 1)"""
import os, sys
VAR = os.environ.get('VP_AB_VAR', 'VP_RS_FLT')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL[os.environ.get("VP_CFG", "cfg2_2048x1536_rggb")]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
variants = [int(v) for v in sys.argv[2:]] or [0]
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
raw = raw4.repeat((B + 3) // 4, *([1] * (raw4.dim() - 1)))[:B].contiguous()
flat = torch.empty((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device='cuda')
d_map = torch.zeros((cfg.fh, cfg.fw, 2), dtype=torch.float32, device='cuda')
ctx.resample_map(cfg.model(), cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh, d_map)
res = {v: [] for v in variants}
def run(N=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        ctx.resample_raw_mapped(raw, cfg.layout, cfg.qw, cfg.qh, d_map, int(os.environ.get("VP_MODE", "1")), cfg.fw, cfg.fh, B, flat)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1000
for rnd in range(6):
    for v in variants:
        os.environ[VAR] = str(v)
        t = run(3) if rnd == 0 else run()
        if rnd: res[v].append(t)
for v, ts in res.items():
    med = float(np.median(ts))
    print(f"resample {cfg.name} B={B} {VAR}={v}: median {med:.1f} us/launch (min {min(ts):.1f}), {med/B:.2f} us/frame")
# the variants must agree bit for bit
outs = {}
for v in variants:
    os.environ[VAR] = str(v)
    flat.zero_()
    ctx.resample_raw_mapped(raw, cfg.layout, cfg.qw, cfg.qh, d_map, int(os.environ.get("VP_MODE", "1")), cfg.fw, cfg.fh, B, flat)
    torch.cuda.synchronize()
    outs[v] = flat.clone()
ref = outs[variants[0]]
for v in variants[1:]:
    print(f"{VAR}={v} equals {VAR}={variants[0]}: {bool(torch.equal(outs[v], ref))}")
