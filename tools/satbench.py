import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.CFG2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
flat = torch.randint(0, 255, (B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device='cuda')
raw = torch.from_numpy(np.stack([cfg.synth(i) for i in range(min(B,4))])).cuda()
m = cfg.model()
for i in range(B):
    ctx.resample_raw(raw[i % raw.shape[0]], cfg.layout, cfg.qw, cfg.qh, m, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh, 1, flat[i])
sat = torch.empty((B, cfg.fh, cfg.fw), dtype=torch.int32, device='cuda')
for variant, name in ((0,'tiled'),(1,'twopass')):
    for _ in range(3): ctx.graddot_sat(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, variant, sat)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    N = 20
    for _ in range(N): ctx.graddot_sat(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, variant, sat)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/N
    print(f"{name} B={B} debug={os.environ.get('VP_SAT_DEBUG','0')}: {ms*1000:.1f} us/launch, {ms*1000/B:.2f} us/frame, {8*cfg.n_flat*B/ms/1e6:.0f} GB/s algorithmic")
