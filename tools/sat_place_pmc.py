"""Fresh-allocation loop of sat_place.py with few launches, to run under rocprofv3 --pmc (tools only).  Prints the time of every
allocation; the k_sat_fast2 dispatches come in groups of 8 per allocation (3 warm-up + 5 timed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL["cfg2_2048x1536_rggb"]
B = 128
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
raw = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
m = cfg.model()
sb = ctx.halo_strip_bytes(cfg.fw, cfg.fh, cfg.grad_offset)
for r in range(6):
    torch.cuda.empty_cache()
    junk = torch.empty(((r * 37 + 5) << 20,), dtype=torch.uint8, device='cuda')
    flat = torch.empty((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device='cuda')
    for i in range(B):
        ctx.resample_raw(raw[i % 4], cfg.layout, cfg.qw, cfg.qh, m, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh, 1, flat[i])
    strips = torch.zeros((B, sb // 4), dtype=torch.int32, device='cuda')
    ctx.halo_strips(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, strips)
    sat = torch.empty((B, cfg.fh, cfg.fw), dtype=torch.int32, device='cuda')
    for _ in range(3): ctx.graddot_sat_strips(flat, strips, cfg.fw, cfg.fh, cfg.grad_offset, B, 0, sat)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ctx.graddot_sat_strips(flat, strips, cfg.fw, cfg.fh, cfg.grad_offset, B, 0, sat)
    e1.record(); torch.cuda.synchronize()
    print(f"fresh allocation {r}: flat @ {flat.data_ptr():#x} strips @ {strips.data_ptr():#x} sat @ {sat.data_ptr():#x}: {e0.elapsed_time(e1) / 5 * 1000:7.1f} us", flush=True)
    del flat, strips, sat, junk
