"""One stage of the hot path alone, timed with stream events (tools only).
usage: stagebench.py STAGE BATCH [--cfg NAME] [--sat tiled|twopass] [--iters N]     STAGE = resample | sat | tail | frame
Prints  "<stage> <cfg> B=<batch> sat=<variant>: median <us> us/launch, <GB/s> GB/s algorithmic"  (algorithmic bytes: DESIGN.md section 3)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
ap = argparse.ArgumentParser()
ap.add_argument("stage"); ap.add_argument("batch", type=int)
ap.add_argument("--cfg", default="cfg2_2048x1536_rggb"); ap.add_argument("--sat", default="tiled"); ap.add_argument("--iters", type=int, default=10); ap.add_argument("--tune", action="append", default=[], help="key=value for vp_ctx_tune"); ap.add_argument("--grad-offset", type=int, default=0, help="override the configuration's gradient offset o (timing only)"); ap.add_argument("--no-strips", action="store_true", help="sat stage: the pass reads its halo columns from flat instead of the halo strips")
a = ap.parse_args()
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL[a.cfg]; B = a.batch; mm = cfg.max_matches
go = a.grad_offset or cfg.grad_offset
variant = capi.SAT_TILED if a.sat == "tiled" else capi.SAT_TWOPASS
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
for kv in a.tune:
    ctx.tune(kv.split("=")[0], int(kv.split("=")[1]))
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
raw = raw4.repeat((B + 3) // 4, *([1] * (raw4.dim() - 1)))[:B].contiguous()
p = capi.frame_params(cfg, sat_variant=variant)
p.grad_offset = go
slab = ctx.frame_slab(capi.frame_params(cfg), B)  # the library's allocator: the table and flat from placement-checked blocks when they are large
_flat, _sat, _m, _idx, _cnt = slab.torch_views()
d = dict(flat=_flat, sat=_sat, matches=_m, idx=_idx, counter=_cnt)
print("placement:", slab.placement["sat"], flush=True)
ctx.frame(p, raw, B, **d); ctx.sync()
d_map = torch.zeros((cfg.fh, cfg.fw, 2), dtype=torch.float32, device="cuda")
ctx.resample_map(cfg.model(), cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh, d_map)
strips = None
nb = ctx.halo_strip_bytes(cfg.fw, cfg.fh, go)
if nb and not a.no_strips:
    strips = torch.zeros((B, nb // 4), dtype=torch.int32, device="cuda")
    ctx.halo_strips(d["flat"], cfg.fw, cfg.fh, go, B, strips)
def fn():
    if a.stage == "resample": ctx.resample_raw_mapped(raw, cfg.layout, cfg.qw, cfg.qh, d_map, cfg.sampler_mode, cfg.fw, cfg.fh, B, d["flat"])
    elif a.stage == "sat":
        if strips is not None and variant == capi.SAT_TILED: ctx.graddot_sat_strips(d["flat"], strips, cfg.fw, cfg.fh, go, B, variant, d["sat"])
        else: ctx.graddot_sat(d["flat"], cfg.fw, cfg.fh, go, B, variant, d["sat"])
    elif a.stage == "tail": ctx.blobcenter_list(d["flat"], d["sat"], cfg.fw, cfg.fh, B, cfg.circle_radius, cfg.circ_threshold, cfg.min_score, cfg.list_radius, mm, capi.ORDER_RASTER, d["matches"], d["idx"], d["counter"])
    else: ctx.frame(p, raw, B, **d)
ts = []
for rnd in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    if rnd: ts.append(e0.elapsed_time(e1) / a.iters * 1000)
ctx.sync()
per_px = {"resample": 1 + 4, "sat": 8, "tail": 4, "frame": 17}[a.stage]  # algorithmic bytes per flat pixel: raw 1 + flat 4 | flat 4 + SAT 4 | SAT 4 | raw 1 + flat 4+4 + SAT 4+4
med = float(np.median(ts))
tunes = " ".join(a.tune)
print(f"{a.stage} {cfg.name} B={B} sat={a.sat} {tunes}: median {med:.1f} us/launch (min {min(ts):.1f}), {med / B:.2f} us/frame, {per_px * cfg.n_flat * B / med / 1e3:.0f} GB/s algorithmic", flush=True)
