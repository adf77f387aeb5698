import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
B = 32
for (w, h) in [(2048, 1536), (2040, 1530), (1994, 1472)]:
    flat = torch.randint(0, 255, (B, h, w, 4), dtype=torch.uint8, device='cuda'); flat[..., 3] = 255
    sat = torch.empty((B, h, w), dtype=torch.int32, device='cuda')
    for _ in range(3): ctx.graddot_sat(flat, w, h, 2, B, capi.SAT_TILED, sat)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ctx.graddot_sat(flat, w, h, 2, B, capi.SAT_TILED, sat)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"{w}x{h} B={B}: {t*1e3:.0f} us/launch, {8*w*h*B/t/1e6:.0f} GB/s algorithmic")
