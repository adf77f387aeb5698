"""Does the tiled SAT pass depend on the RELATIVE placement of its input and output?  One pool allocation; flat at its start, the table at a
chosen byte offset behind it; time the pass for each offset (tools only; GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL["cfg2_2048x1536_rggb"]; B = 128
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
n = B * cfg.fh * cfg.fw * 4
pool = torch.empty(2 * n + (512 << 20), dtype=torch.uint8, device="cuda")
flat = pool[:n].view(B, cfg.fh, cfg.fw, 4)
flat.random_(0, 255)
print(f"pool at {pool.data_ptr():#x}", flush=True)
def t(sat):
    for _ in range(3): ctx.graddot_sat(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, capi.SAT_TILED, sat)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ctx.graddot_sat(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, capi.SAT_TILED, sat)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100
for rnd in range(2):
    for d in (0, 4 << 10, 64 << 10, 256 << 10, 1 << 20, (2 << 20) + 4096, 8 << 20, (16 << 20) + (64 << 10), 64 << 20, 100 << 20, 256 << 20, 400 << 20):
        sat = pool[n + d: n + d + n].view(torch.int32).view(B, cfg.fh, cfg.fw)
        print(f"round {rnd} table offset behind flat +{d >> 10:7d} KiB: {t(sat):6.1f} us", flush=True)
# the same pair of buffers from separate allocations, several times
for i in range(4):
    f2 = torch.empty((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda").random_(0, 255); s2 = torch.empty((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda")
    flat_old = flat; flat = f2
    print(f"separate allocations {i}: flat {f2.data_ptr():#x} sat {s2.data_ptr():#x}: {t(s2):6.1f} us", flush=True)
    flat = flat_old
    keep = (f2, s2) if i % 2 == 0 else None
