"""Full-size soak of the timed launch shape: feeds.BENCH_FRAMES_PER_STEP (256) resident cfg2 frames, their ORDER in the batch reshuffled every iteration (so every frame meets every
slot, frame-group position and neighbour), vp_frame with the bench's parameters on placement-checked buffers; per-frame checksums of flat, the table,
counters and records must not depend on the position, and a few frames are compared with the CPU oracle outright.  GPU; tools only.
usage: fullsize_soak.py [SECONDS=120] [tiled|local]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
from _vp_pkg import load
import oracle as O
vp = load(); capi = vp.load_capi()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
local = len(sys.argv) > 2 and sys.argv[2] == "local"
cfg = vp.configs.ALL["cfg2_2048x1536_rggb"]; B, mm = vp.feeds.BENCH_FRAMES_PER_STEP, cfg.max_matches
with ThreadPoolExecutor(16) as ex:
    raws = np.stack(list(ex.map(cfg.synth, [vp.feeds.feed_seed(3, i) for i in range(B)])))
base = torch.from_numpy(raws).cuda()
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
p = capi.frame_params(cfg, sat_variant=capi.SAT_LOCAL if local else capi.SAT_TILED, order=capi.ORDER_RASTER)
slab = ctx.frame_slab(p, B)
flat, sat, m, idx, cnt = slab.torch_views()
w64 = torch.arange(1, cfg.fh * cfg.fw + 1, device="cuda", dtype=torch.int64)

def sums():
    f = (flat.view(B, -1, 4).to(torch.int64).sum(-1) * w64).sum(1)
    s = (sat.view(B, -1).to(torch.int64) * w64).sum(1) if sat is not None else torch.zeros(B, dtype=torch.int64, device="cuda")
    n = cnt[:, 0].clamp(max=mm).to(torch.int64)
    mask = (torch.arange(22 * mm, device="cuda")[None, :] < 22 * n[:, None])
    r = (m.to(torch.int64) * mask * torch.arange(1, 22 * mm + 1, device="cuda", dtype=torch.int64)).sum(1)
    return torch.stack([f, s, r, cnt[:, 0].to(torch.int64), cnt[:, 1].to(torch.int64), cnt[:, 2].to(torch.int64)], 1)

port = O.Port(threads=16)
model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
checked = set()
def oracle_check(pos, orig):
    s = port.frame_scratch(cfg.qw, cfg.qh, cfg.fw, cfg.fh, mm)
    wm, wi, wc = port.frame(raws[orig], cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.sampler_mode, cfg.fw, cfg.fh,
                            cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, mm, scratch=s)
    assert np.array_equal(flat[pos].cpu().numpy(), s["flat"]) and np.array_equal(cnt[pos].cpu().numpy(), wc), ("oracle", pos, orig)
    if sat is not None:
        assert np.array_equal(sat[pos].cpu().numpy(), port.sat_i32(s["graddot"])[0]), ("oracle SAT", pos, orig)
    n = int(wc[0])
    assert np.array_equal(idx[pos, :n].cpu().numpy().view(np.uint32), wi) and m[pos, :22 * n].cpu().numpy().tobytes() == wm.tobytes(), ("oracle records", pos, orig)
    checked.add(orig)

rng = np.random.default_rng(7)
ref, it, t0, t_note = None, 0, time.time(), time.time()
while time.time() - t0 < budget:
    perm = torch.from_numpy(rng.permutation(B)).cuda()
    d_raw = base[perm].contiguous()
    ctx.frame(p, d_raw, B, flat, sat, m, cnt, idx=idx)
    ctx.sync()
    got = sums()
    if ref is None:
        ref = torch.empty_like(got); ref[perm] = got  # per ORIGINAL frame
    else:
        assert torch.equal(got, ref[perm]), ("a frame's result depends on its position", it, (got != ref[perm]).nonzero()[:4].tolist())
    if it % 25 == 0:
        pos = int(rng.integers(B)); oracle_check(pos, int(perm[pos]))
    it += 1
    if time.time() - t_note > 60:
        t_note = time.time(); print(f"... {it} launches, {len(checked)} oracle checks, all consistent", flush=True)
print(f"full-size soak ({'local' if local else 'tiled'}): {it} launches of {B} reshuffled frames = {it * B} frames, position-independent checksums of flat / table / counters / records, "
      f"{len(checked)} distinct frames equal to the CPU oracle byte for byte, {time.time() - t0:.0f} s")
