"""Footprint-box statistics of the staged resampling kernel (tools only; GPU for the coordinate map): per 64x4 flat tile the sensor box
(dword columns x rows) and the 128-byte lines it touches.  usage: boxstats.py [CFG] [TW TH]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL[sys.argv[1] if len(sys.argv) > 1 else "cfg2_2048x1536_rggb"]
TW, TH = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 4)
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
d_map = torch.zeros((cfg.fh, cfg.fw, 2), dtype=torch.float32, device="cuda")
ctx.resample_map(cfg.model(), cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh, d_map); ctx.sync()
m = d_map.cpu().numpy()
px, py = m[..., 0], m[..., 1]
qw, qh = cfg.qw, cfg.qh
def lin(u, n):
    f = np.floor(u - 0.5).astype(np.int64)
    return np.clip(f, 0, n - 1), np.clip(f + 1, 0, n - 1)
xa0, xa1 = lin(px + 0.25, qw); xb0, xb1 = lin(px - 0.25, qw)
ya0, ya1 = lin(py + 0.25, qh); yb0, yb1 = lin(py - 0.25, qh)
cmin_b = 2 * np.minimum(xa0, xb0); cmax_b = 2 * np.maximum(xa1, xb1) + 1   # byte columns
rmin = np.minimum(2 * ya0, 2 * yb0 + 1); rmax = np.maximum(2 * ya1, 2 * yb1 + 1)
H, W = (cfg.fh // TH) * TH, (cfg.fw // TW) * TW
def tiles(a, fn): return fn(fn(a[:H, :W].reshape(H // TH, TH, W // TW, TW), axis=3), axis=1)
c0, c1, r0, r1 = tiles(cmin_b, np.min), tiles(cmax_b, np.max), tiles(rmin, np.min), tiles(rmax, np.max)
d0, d1 = c0 // 4, c1 // 4 + 1
wdw, hr = d1 - d0, r1 - r0 + 1
lines = (c1 // 128 - c0 // 128 + 1) * hr
print(f"{cfg.name}: sensor {2*qw}x{2*qh} bytes, flat {cfg.fw}x{cfg.fh}, tile {TW}x{TH}")
for name, a in (("wdw (dwords)", wdw), ("rows", hr), ("box dwords", wdw * hr), ("128-B lines", lines)):
    print(f"  {name:14s} mean {a.mean():7.2f}  p50 {np.median(a):5.0f}  p99 {np.percentile(a, 99):5.0f}  max {a.max():5d}")
print(f"  useful bytes per tile-frame {4 * (wdw * hr).mean():.0f}, line bytes {128 * lines.mean():.0f}, lines per flat pixel {lines.mean() / (TW * TH):.4f}")
print(f"  clamped columns (ia1 != ia0+1 anywhere in the tile): {np.mean(tiles((xa1 != xa0 + 1) | (xb1 != xb0 + 1), np.max)) * 100:.2f} % of tiles")
