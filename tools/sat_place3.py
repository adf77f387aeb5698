"""Which buffer's physical placement moves the gradientDot+SAT pass?  (tools only)  One process; K distinct live allocations of the
output (input fixed), then K distinct live allocations of the input (output fixed), then of the strips."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL["cfg2_2048x1536_rggb"]
B, K = 128, 6
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
raw = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
m = cfg.model()
sb = ctx.halo_strip_bytes(cfg.fw, cfg.fh, cfg.grad_offset)
flat0 = torch.empty((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device='cuda')
for i in range(B):
    ctx.resample_raw(raw[i % 4], cfg.layout, cfg.qw, cfg.qh, m, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh, 1, flat0[i])
strips0 = torch.zeros((B, sb // 4), dtype=torch.int32, device='cuda')
ctx.halo_strips(flat0, cfg.fw, cfg.fh, cfg.grad_offset, B, strips0)

def timeit(flat, strips, sat, n=10):
    for _ in range(3): ctx.graddot_sat_strips(flat, strips, cfg.fw, cfg.fh, cfg.grad_offset, B, 0, sat)
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): ctx.graddot_sat_strips(flat, strips, cfg.fw, cfg.fh, cfg.grad_offset, B, 0, sat)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1000)
    return float(np.median(ts))

sats = [torch.empty((B, cfg.fh, cfg.fw), dtype=torch.int32, device='cuda') for _ in range(K)]
print("outputs (input fixed):   " + "  ".join(f"{timeit(flat0, strips0, s):6.1f}" for s in sats))
print("outputs again:           " + "  ".join(f"{timeit(flat0, strips0, s):6.1f}" for s in sats))
flats = [flat0] + [flat0.clone() for _ in range(K - 1)]
print("inputs (output 0 fixed): " + "  ".join(f"{timeit(f, strips0, sats[0]):6.1f}" for f in flats))
print("inputs (output 1 fixed): " + "  ".join(f"{timeit(f, strips0, sats[1]):6.1f}" for f in flats))
strs = [strips0] + [strips0.clone() for _ in range(K - 1)]
print("strips (rest fixed):     " + "  ".join(f"{timeit(flat0, s, sats[0]):6.1f}" for s in strs))
print("pairs (flat i, sat i):   " + "  ".join(f"{timeit(flats[i], strs[i], sats[i]):6.1f}" for i in range(K)))
