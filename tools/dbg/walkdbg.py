import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np, torch
from _vp_pkg import load
import oracle as O
vp = load(); capi = vp.load_capi()
port = O.Port(threads=4)
ctx = capi.Context(0, n_streams=1)
r = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w, h = 192, 128
rng = np.random.default_rng(1)
gd = rng.integers(-900, 900, (h, w)).astype(np.float32)
sat = port.sat_i32(gd)[0]
flat = rng.integers(0, 256, (1, h, w, 4), dtype=np.uint8)
d_sat = torch.from_numpy(sat[None]).cuda(); d_flat = torch.from_numpy(flat).cuda()
mm = 100
d_m = torch.zeros((1, mm * 22), dtype=torch.uint8, device="cuda"); d_idx = torch.zeros((1, mm), dtype=torch.int32, device="cuda"); d_cnt = torch.zeros((1, 3), dtype=torch.int32, device="cuda")
d_bc = torch.full((1, h, w), np.nan, dtype=torch.float32, device="cuda")
ctx.blobcenter_list(d_flat, d_sat, w, h, 1, r, 1e9, 0.0, 2, mm, capi.ORDER_RASTER, d_m, d_idx, d_cnt, d_bc)
ctx.sync()
got = d_bc[0].cpu().numpy(); want = port.blobcenter_i32(sat, r)
bad = np.argwhere(~(got == want))
print("r", r, "mismatches", len(bad), "of", w * h)
if len(bad):
    cols = np.unique(bad[:, 1]); rows = np.unique(bad[:, 0])
    print("cols", cols[:80]); print("rows", rows[:40], "...", len(rows))
    y, x = bad[0]; print("first", y, x, got[y, x], want[y, x])
if len(bad):
    for yy in (0, 1, 2, 3, 10, 11, 12, 40, 127):
        print("row", yy, "bad x:", np.nonzero(~(got[yy] == want[yy]))[0][:70])
    print("per-row bad counts:", [(int(yy), int((~(got[yy] == want[yy])).sum())) for yy in range(0, 40)])
os.makedirs("gpurun_out/dbg", exist_ok=True)
np.savez_compressed(f"gpurun_out/dbg/walk_r{r}.npz", got=got, want=want, sat=sat)
