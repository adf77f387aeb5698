"""Does the issue-bound resampling of step k+1 run beside the HBM-bound kernels of step k?  (VERDICT r05, next #1a.)
Stage entries of the C ABI on adopted torch streams, ordered by events only where the data demands it; each step works on its own buffer set
(2 or 3 sets taking turns), so the launches keep their full size.  Modes:
  seq      R S T of a step back to back on one stream (today's order; stage entries, no strips)
  rt       R(k+1) beside T(k); the SAT pass runs alone (S(k+1) waits for T(k))
  rt_free  as rt without that wait: S(k+1) may start beside T(k)'s end
  free     three streams, three buffer sets, data dependencies only: R(k+2), S(k+1), T(k) all in flight
  frame    vp_frame on one stream, sets taking turns (reference point: the shipped path with strips)
usage: overlap_exp.py [frames per launch = 256] [hi = which stream gets high priority: none|t|r]      GPU; tools only."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.CFG2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
HI = sys.argv[2] if len(sys.argv) > 2 else "none"
K = 12
mm = cfg.max_matches
raw4 = torch.from_numpy(np.stack([cfg.synth(i) for i in range(4)])).cuda()
d_raw = raw4.repeat(N // 4, 1, 1).contiguous()
NSETS = 3
sets = [dict(flat=torch.zeros((N, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda"), sat=torch.zeros((N, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda"),
             matches=torch.zeros((N, mm * 22), dtype=torch.uint8, device="cuda"), idx=torch.zeros((N, mm), dtype=torch.int32, device="cuda"),
             counter=torch.zeros((N, 3), dtype=torch.int32, device="cuda")) for _ in range(NSETS)]
d_map = torch.zeros((cfg.fh, cfg.fw, 2), dtype=torch.float32, device="cuda")
pr = lambda who: -1 if HI == who else 0
sR, sS, sT = torch.cuda.Stream(priority=pr("r")), torch.cuda.Stream(priority=pr("s")), torch.cuda.Stream(priority=pr("t"))
cR, cS, cT = (capi.Context(0, stream_ptr=s.cuda_stream) for s in (sR, sS, sT))
p = capi.frame_params(cfg)
cR.resample_map(cfg.model(), cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh, d_map); cR.sync()
torch.cuda.synchronize()

def R(c, d): c.resample_raw_mapped(d_raw, cfg.layout, cfg.qw, cfg.qh, d_map, cfg.sampler_mode, cfg.fw, cfg.fh, N, d["flat"])
def S(c, d): c.graddot_sat(d["flat"], cfg.fw, cfg.fh, cfg.grad_offset, N, capi.SAT_TILED, d["sat"])
def T(c, d): c.blobcenter_list(d["flat"], d["sat"], cfg.fw, cfg.fh, N, cfg.circle_radius, cfg.circ_threshold, cfg.min_score, cfg.list_radius, mm, capi.ORDER_RASTER, d["matches"], d["idx"], d["counter"])

def run_seq(k0, steps):
    for k in range(k0, k0 + steps):
        d = sets[k % 2]; R(cR, d); S(cR, d); T(cR, d)

def run_frame(k0, steps):
    for k in range(k0, k0 + steps): cR.frame(p, d_raw, N, **sets[k % 2])

def run_rt(k0, steps, gate=True):
    evT = {}
    for k in range(k0, k0 + steps):
        d = sets[k % 2]
        R(cR, d)                                   # beside T(k-1) on sT
        if gate and k - 1 in evT: sR.wait_event(evT[k - 1])
        elif k - 2 in evT: sR.wait_event(evT[k - 2])   # (R(k+1) will overwrite the set T(k-1) reads: ordered through S(k) when gated, explicitly here otherwise)
        S(cR, d)
        e = torch.cuda.Event(); e.record(sR); sT.wait_event(e)
        T(cT, d)
        evT[k] = torch.cuda.Event(); evT[k].record(sT)
    sR.wait_event(evT[k0 + steps - 1])

def run_rt_free(k0, steps):
    # R(k) must not overwrite flat of set k % 2 before T(k-2) has read it
    evT = {}
    for k in range(k0, k0 + steps):
        d = sets[k % 2]
        if k - 2 in evT: sR.wait_event(evT[k - 2])
        R(cR, d); S(cR, d)
        e = torch.cuda.Event(); e.record(sR); sT.wait_event(e)
        T(cT, d)
        evT[k] = torch.cuda.Event(); evT[k].record(sT)
    sR.wait_event(evT[k0 + steps - 1])

def run_free(k0, steps):
    evT = {}
    for k in range(k0, k0 + steps):
        d = sets[k % 3]
        if k - 3 in evT: sR.wait_event(evT[k - 3])
        R(cR, d)
        e = torch.cuda.Event(); e.record(sR); sS.wait_event(e)
        S(cS, d)
        e = torch.cuda.Event(); e.record(sS); sT.wait_event(e)
        T(cT, d)
        evT[k] = torch.cuda.Event(); evT[k].record(sT)
    sR.wait_event(evT[k0 + steps - 1])

modes = dict(seq=run_seq, frame=run_frame, rt=run_rt, rt_free=run_rt_free, free=run_free)
# reference results from the sequential order
run_seq(0, 2); torch.cuda.synchronize()
ref = [{k: v.clone() for k, v in sets[i].items() if k in ("counter", "sat")} for i in range(2)]
res = {m: [] for m in modes}
for rnd in range(4):
    for m, fn in modes.items():
        fn(0, 3); torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(0, K); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rnd: res[m].append(round(dt / K * 1e3, 4))
        for i in range(2):
            for k in ref[i]: assert torch.equal(ref[i][k], sets[i][k]), (m, k)
for m, ts in res.items():
    print(f"N={N} hi={HI} {m:8s}: ms per step {ts}  -> {N / min(ts):.1f} k frames/s", flush=True)
print(json.dumps({"frames_per_launch": N, "high_priority": HI, "ms_per_step": res}))
