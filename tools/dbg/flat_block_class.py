"""Does the write-rate class of flat's hipMalloc block matter to the resampling kernel?  Allocates N candidate blocks for flat (256 cfg2
frames each), probes each block's zero-fill rate (vp_probe_stream) and times vp_resample_raw_mapped into each, interleaved.  GPU; tools only.
usage: flat_block_class.py [N=8]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load()
capi = vp.load_capi()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = vp.configs.ALL["cfg2_2048x1536_rggb"]
B = 256
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
lib = ctx.l
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
raw = raw4.repeat(B // 4, 1, 1).contiguous()
d_map = torch.zeros((cfg.fh, cfg.fw, 2), dtype=torch.float32, device="cuda")
ctx.resample_map(cfg.model(), cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh, d_map)
nbytes = B * cfg.fh * cfg.fw * 4
blocks = [ctx.malloc(nbytes) for _ in range(N)]
def probe(ptr):
    us = C.c_float()
    assert lib.vp_probe_stream(ctx.h, 0, C.c_void_p(ptr), C.c_size_t(nbytes), 1, 5, C.byref(us)) == 0, lib.vp_last_error()
    return nbytes / us.value / 1e3  # GB/s
rates = [probe(p) for p in blocks]
res = [[] for _ in blocks]
for rnd in range(6):
    for i, p in enumerate(blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ctx.resample_raw_mapped(raw, cfg.layout, cfg.qw, cfg.qh, d_map, cfg.sampler_mode, cfg.fw, cfg.fh, B, p)
        e1.record(); torch.cuda.synchronize()
        res[i].append(e0.elapsed_time(e1) * 1000 / 5)
for i, p in enumerate(blocks):
    print(f"block {i} {hex(p)}: zero-fill {rates[i]:7.1f} GB/s, resample median {np.median(res[i]):7.1f} us per 256 frames")
