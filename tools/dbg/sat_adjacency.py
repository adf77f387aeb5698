"""The SAT pass inside vp_frame (right behind the resampling kernel) against the same launch alone on the same buffers (placement-checked slab).  Tools only."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.CFG2; B = 128
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
d_raw = raw4.repeat(B // 4, 1, 1).contiguous()
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
p = capi.frame_params(cfg)
slab = ctx.frame_slab(p, B)
flat, sat, m, idx, cnt = slab.torch_views()
strips = torch.zeros(ctx.halo_strip_bytes(cfg.fw, cfg.fh, cfg.grad_offset) * B, dtype=torch.uint8, device="cuda")
for _ in range(3): ctx.frame_buffers(p, d_raw, B, slab.b)
ctx.halo_strips(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, strips); ctx.sync()
def stage(fn, n=10):
    ctx.events_clear(); ctx.events_enable(True)
    for _ in range(n): fn()
    ctx.sync(); ctx.events_enable(False)
    acc = {}
    for name, ms in ctx.event_times(): acc.setdefault(name, []).append(ms)
    return {k: round(float(np.median(v)) * 1e3, 1) for k, v in acc.items()}
for rnd in range(3):
    print("in vp_frame      :", stage(lambda: ctx.frame_buffers(p, d_raw, B, slab.b)), flush=True)
    print("SAT alone        :", stage(lambda: ctx.graddot_sat_strips(flat, strips, cfg.fw, cfg.fh, cfg.grad_offset, B, capi.SAT_TILED, sat)), flush=True)
    def both():
        ctx.graddot_sat_strips(flat, strips, cfg.fw, cfg.fh, cfg.grad_offset, B, capi.SAT_TILED, sat)
        ctx.blobcenter_list(flat, sat, cfg.fw, cfg.fh, B, cfg.circle_radius, cfg.circ_threshold, cfg.min_score, cfg.list_radius, cfg.max_matches, capi.ORDER_RASTER, m, idx, cnt)
    print("SAT + tail       :", stage(both), flush=True)
print(slab.placement)
