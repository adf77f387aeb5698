"""Phase timeline of the tiled SAT kernel from a TRACE build (temporary instrumentation; not part of the product)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL["cfg2_2048x1536_rggb"]
B = int(sys.argv[1]); wpc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
if wpc: ctx.tune("sat_workgroups_per_cu", wpc)
L = capi.lib()
flat = torch.randint(0, 255, (B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"); sat = torch.empty((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda")
NB = 8192 * 16 * 8 * 8
for _ in range(3): ctx.graddot_sat(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, capi.SAT_TILED, sat)
ctx.sync()
L.vp_dbg_trace_clear.argtypes = [C.c_size_t]; L.vp_dbg_trace_copy.argtypes = [C.c_void_p, C.c_size_t]
assert L.vp_dbg_trace_clear(NB) == 0
ctx.graddot_sat(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, capi.SAT_TILED, sat); ctx.sync()
ctx.graddot_sat(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, capi.SAT_TILED, sat); ctx.sync()
h = np.zeros(NB // 8, dtype=np.uint64)
assert L.vp_dbg_trace_copy(h.ctypes.data, NB) == 0
t = h.reshape(-1, 16, 8).astype(np.int64)
used = t[:, :, 1] > 0
t0 = t[:, 0, 0][t[:, 0, 0] > 0].min()
us = lambda x: (x - t0) / 100.0   # s_memrealtime: 100 MHz
wg = np.where(used[:, 0])[0]
print(f"B={B}: {len(wg)} workgroups, {used.sum()} tiles traced; kernel span {us(t[:, :, 5][used].max()):.1f} us")
start = us(t[wg, 0, 0]); print(f"workgroup start: min {start.min():.1f} p50 {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f} us")
for name, a, b in (("A  (commit prefetch -> LDS)", 1, 2), ("B  (local SAT)", 2, 3), ("CD (publish + look-back)", 3, 4), ("E0 (K publish + prefetch issue)", 4, 6), ("E1 (adds + stores)", 6, 5)):
    d = (t[:, :, b] - t[:, :, a])[used] / 100.0
    print(f"  {name:30s} mean {d.mean():6.2f}  p50 {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
first = (t[:, 0, 1] - t[:, 0, 0])[used[:, 0]] / 100.0
print(f"  start -> first A              mean {first.mean():6.2f} p90 {np.percentile(first, 90):6.2f}")
tile = t[:, :, 7][used]; ti, tj = tile >> 16, tile & 0xFFFF
endt = us(t[:, :, 4][used]); d = ti + tj
for dd in (0, 5, 10, 20, 30, 40, 50, 54):
    m = d == dd
    if m.any(): print(f"  diagonal {dd:2d}: look-back resolved at {endt[m].min():6.1f} .. {endt[m].max():6.1f} us ({m.sum()} tiles)")
