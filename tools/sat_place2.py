"""gradientDot+SAT pass vs the distance between its input (flat) and output (SAT) inside ONE arena (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL["cfg2_2048x1536_rggb"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEP = int(sys.argv[2]) << 20 if len(sys.argv) > 2 else 64 << 20
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
raw = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
m = cfg.model()
N = cfg.fh * cfg.fw
fbytes = B * N * 4
arena = torch.empty(2 * fbytes + K * STEP + (4 << 20), dtype=torch.uint8, device='cuda')
base = (-arena.data_ptr()) % (2 << 20)  # 2 MB aligned start
flat = arena[base: base + fbytes].view(B, cfg.fh, cfg.fw, 4)
for i in range(B):
    ctx.resample_raw(raw[i % 4], cfg.layout, cfg.qw, cfg.qh, m, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh, 1, flat[i])
sb = ctx.halo_strip_bytes(cfg.fw, cfg.fh, cfg.grad_offset)
strips = torch.zeros((B, sb // 4), dtype=torch.int32, device='cuda')
ctx.halo_strips(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, strips)

def timeit(sat, n=10):
    for _ in range(3): ctx.graddot_sat_strips(flat, strips, cfg.fw, cfg.fh, cfg.grad_offset, B, 0, sat)
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): ctx.graddot_sat_strips(flat, strips, cfg.fw, cfg.fh, cfg.grad_offset, B, 0, sat)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1000)
    return float(np.median(ts))

print(f"arena @ {arena.data_ptr():#x} flat @ {flat.data_ptr():#x} strips @ {strips.data_ptr():#x} step {STEP >> 20} MB")
for k in list(range(K + 1)) + [0]:
    off = base + fbytes + k * STEP
    sat = arena[off: off + fbytes].view(torch.int32).view(B, cfg.fh, cfg.fw)
    print(f"sat - flat = {(off - base) >> 20:5d} MB: {timeit(sat):7.1f} us")
