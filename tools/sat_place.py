"""Does the gradientDot+SAT pass depend on WHERE its buffers lie?  (tools only)
1. one process, fixed flat: the SAT output placed at different byte offsets inside one big allocation;
2. everything re-allocated several times (fresh hipMalloc blocks each time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL["cfg2_2048x1536_rggb"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
raw = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
m = cfg.model()
N = cfg.fh * cfg.fw
sb = ctx.halo_strip_bytes(cfg.fw, cfg.fh, cfg.grad_offset)

def make_inputs():
    flat = torch.empty((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device='cuda')
    for i in range(B):
        ctx.resample_raw(raw[i % 4], cfg.layout, cfg.qw, cfg.qh, m, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh, 1, flat[i])
    strips = torch.zeros((B, sb // 4), dtype=torch.int32, device='cuda')
    ctx.halo_strips(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, strips)
    return flat, strips

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

flat, strips = make_inputs()
pad = 64 << 20
big = torch.empty(B * N + pad // 4, dtype=torch.int32, device='cuda')
print(f"flat @ {flat.data_ptr():#x}  strips @ {strips.data_ptr():#x}  sat block @ {big.data_ptr():#x}")
for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 8 << 20, 32 << 20, 0):
    sat = big[off // 4: off // 4 + B * N].view(B, cfg.fh, cfg.fw)
    print(f"sat offset {off:>9d} B: {timeit(flat, strips, sat):7.1f} us")
del big, flat, strips
for r in range(6):
    torch.cuda.empty_cache()
    junk = torch.empty(((r * 37 + 5) << 20,), dtype=torch.uint8, device='cuda')  # shifts where the next blocks land
    flat, strips = make_inputs()
    sat = torch.empty((B, cfg.fh, cfg.fw), dtype=torch.int32, device='cuda')
    print(f"fresh allocation {r}: flat @ {flat.data_ptr():#x} sat @ {sat.data_ptr():#x}: {timeit(flat, strips, sat):7.1f} us")
    del flat, strips, sat, junk
