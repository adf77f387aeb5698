"""What differs between a fast-writing and a slow-writing hipMalloc block at the L2 -> fabric interface?  Finds one block of each kind (zero-fill probe),
then runs 6 zero-fill sweeps on the fast one, 6 on the slow one, 6 on the fast one again; run under `rocprofv3 --kernel-trace --pmc <counters>`
and read the per-dispatch counters of k_probe_write in dispatch order.  GPU; tools only.  Prints the order as JSON on the last line."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
ctx = capi.Context(0, n_streams=1)
lib = capi.lib()
nb = 128 * 2048 * 1536 * 4

def probe(ptr, reps):
    us = C.c_float()
    assert lib.vp_probe_stream(ctx.h, 0, C.c_void_p(ptr), C.c_size_t(nb), 1, reps, C.byref(us)) == 0, lib.vp_last_error()
    return us.value

held, search = [], []
for _ in range(16):  # under counter collection every dispatch is slower: classify relative to each other, not by absolute time
    p = ctx.malloc(nb)
    held.append(p)
    search.append(probe(p, 2))  # 3 dispatches each
lo, hi = min(search), max(search)
assert hi > 1.12 * lo, ("no slow block among 16", search)
fast, slow = (held[search.index(lo)], lo), (held[search.index(hi)], hi)
seq = []
for name, (p, _) in (("fast", fast), ("slow", slow), ("fast", fast)):
    us = probe(p, 5)  # 6 dispatches
    seq.append({"block": name, "us": round(us, 1), "dispatches": 6})
print(json.dumps({"search_dispatches": 3 * len(search), "search_us": [round(v) for v in search], "sequence": seq}))
for p in held:
    ctx.free(p)
ctx.close()
