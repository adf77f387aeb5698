"""Where do the SAT pass's 0.60-0.72 ms come from?  K allocation sets per allocation mode in ONE process; for each set the stage times of the
bench launch (cfg2 x 128 frames) and the pure read / write stream times of its flat and SAT buffers (vp_probe_stream).  GPU; tools only.
usage: python tools/placement_probe.py [sets=6] [modes=0,1,2]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL["cfg2_2048x1536_rggb"]; B = 128
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
modes = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2").split(",")]
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
lib = capi.lib()
n = B * cfg.n_flat * 4
raws = np.stack([cfg.synth(i % 4) for i in range(B)])
d_raw = torch.from_numpy(raws).cuda()
mm = cfg.max_matches
m = torch.zeros((B, 22 * mm), dtype=torch.uint8, device="cuda"); idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda"); cnt = torch.zeros((B, 3), dtype=torch.int32, device="cuda")
p = capi.frame_params(cfg)

def alloc(mode):
    q = C.c_void_p()
    rc = lib.vp_malloc_device_ex(ctx.h, C.c_size_t(n), mode, C.byref(q))
    assert rc == 0, lib.vp_last_error()
    return q.value

def probe(ptr, write):
    us = C.c_float()
    assert lib.vp_probe_stream(ctx.h, 0, C.c_void_p(ptr), C.c_size_t(n), write, 5, C.byref(us)) == 0, lib.vp_last_error()
    return us.value

def stages(flat, sat):
    for _ in range(3):
        ctx.frame(p, d_raw, B, flat, sat, m, cnt, idx=idx)
    ctx.sync(); ctx.events_clear(); ctx.events_enable(True)
    for _ in range(10):
        ctx.frame(p, d_raw, B, flat, sat, m, cnt, idx=idx)
    ctx.events_enable(False)
    out = {}
    for name, ms in ctx.event_times():
        out.setdefault(name, []).append(ms)
    ctx.events_clear()
    return {k: round(float(np.mean(v)) * 1e3, 1) for k, v in out.items()}

rows = []
for mode in modes:
    held = []
    for k in range(K):
        f, s = alloc(mode), alloc(mode)
        st = stages(f, s)
        row = dict(mode=mode, set=k, flat=hex(f), sat=hex(s), read_flat_us=round(probe(f, 0), 1), read_sat_us=round(probe(s, 0), 1), stage_us=st)
        st2 = stages(f, s)
        row["stage_us_again"] = st2
        row["write_sat_us"] = round(probe(s, 1), 1)
        row["write_flat_us"] = round(probe(f, 1), 1)
        # swapped roles: is it the buffer or its role?
        row["stage_us_swapped"] = stages(s, f)
        print(json.dumps(row), flush=True)
        rows.append(row)
        if k % 2 == 0:
            held.append((f, s))  # keep every other set alive so that later sets land elsewhere
        else:
            ctx.free(f); ctx.free(s)
    for f, s in held:
        ctx.free(f); ctx.free(s)
sat_key = "gradientDot+sat"
for mode in modes:
    v = [r["stage_us"][sat_key] for r in rows if r["mode"] == mode]
    print(f"mode {mode}: SAT pass {min(v):.1f} .. {max(v):.1f} us over {len(v)} sets; read(flat) {min(r['read_flat_us'] for r in rows if r['mode'] == mode):.1f} .. {max(r['read_flat_us'] for r in rows if r['mode'] == mode):.1f}")
ctx.close()
