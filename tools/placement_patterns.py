"""Does a block's write class (fast / slow zero-fill) depend on the ACCESS PATTERN of the writer?  For 14 hipMalloc blocks of 1.6 GB: the library's probe (4096
workgroups, grid-stride 16 MB, nt 16-byte stores), hipMemsetAsync (the runtime's fill kernel), and the single-pass SAT kernel writing its table into the block
(128 frames, flat in one fixed other block).  GPU; tools only.  Prints JSON."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.CFG2; B = 128
ctx = capi.Context(0, n_streams=1)
lib = capi.lib()
Q = B * cfg.fw * cfg.fh * 4


def probe(ptr):
    us = C.c_float()
    assert lib.vp_probe_stream(ctx.h, 0, C.c_void_p(ptr), C.c_size_t(Q), 1, 3, C.byref(us)) == 0
    return round(Q / us.value * 1e-3)


def memset_rate(ptr):
    ctx.memset(ptr, 0, Q); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3): ctx.memset(ptr, 0, Q)
    ctx.sync()
    return round(Q / ((time.perf_counter() - t0) / 3) * 1e-9)


raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
d_raw = raw4.repeat(B // 4, 1, 1).contiguous()
flat = ctx.malloc(Q)
strips = ctx.malloc(ctx.halo_strip_bytes(cfg.fw, cfg.fh, cfg.grad_offset) * B)
tmp_sat = ctx.malloc(Q)
mm = cfg.max_matches
m = torch.zeros((B, 22 * mm), dtype=torch.uint8, device="cuda"); idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda"); cnt = torch.zeros((B, 3), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
p = capi.frame_params(cfg)
ctx.frame(p, d_raw, B, flat, tmp_sat, m, cnt, idx=idx); ctx.sync()   # flat is now a real frame group
ctx.halo_strips(flat, cfg.fw, cfg.fh, cfg.grad_offset, B, strips); ctx.sync()


def sat_us(ptr):
    for _ in range(2): ctx.graddot_sat_strips(flat, strips, cfg.fw, cfg.fh, cfg.grad_offset, B, capi.SAT_TILED, ptr)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(5): ctx.graddot_sat_strips(flat, strips, cfg.fw, cfg.fh, cfg.grad_offset, B, capi.SAT_TILED, ptr)
    ctx.sync()
    return round((time.perf_counter() - t0) / 5 * 1e6, 1)


rows, held = [], []
for _ in range(14):
    b = ctx.malloc(Q); held.append(b)
    rows.append({"probe_GBs": probe(b), "hipMemset_GBs": memset_rate(b), "sat_pass_us": sat_us(b), "probe_again_GBs": probe(b)})
print(json.dumps({"blocks": rows}))
pr = np.array([r["probe_GBs"] for r in rows]); ms = np.array([r["hipMemset_GBs"] for r in rows]); st = np.array([r["sat_pass_us"] for r in rows])
print("corr(probe, memset) %.3f  corr(probe, -sat_us) %.3f  corr(memset, -sat_us) %.3f" % (np.corrcoef(pr, ms)[0, 1], np.corrcoef(pr, -st)[0, 1], np.corrcoef(ms, -st)[0, 1]))
