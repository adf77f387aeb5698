"""Write-stream speed of 1.5 GB hipMalloc buffers against allocation order, up to a total; persistence; variation inside a buffer.  GPU; tools only."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
lib = capi.lib()
nb = 128 * 2048 * 1536 * 4
total_gb = int(sys.argv[1]) if len(sys.argv) > 1 else 150
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0

def alloc(nbytes):
    q = C.c_void_p()
    assert lib.vp_malloc_device_ex(ctx.h, C.c_size_t(nbytes), mode, C.byref(q)) == 0, lib.vp_last_error()
    return q.value

def probe(ptr, nbytes, write, reps=3):
    us = C.c_float()
    assert lib.vp_probe_stream(ctx.h, 0, C.c_void_p(ptr), C.c_size_t(nbytes), write, reps, C.byref(us)) == 0, lib.vp_last_error()
    return us.value

ptrs, w1 = [], []
while (len(ptrs) + 1) * nb <= total_gb << 30:
    p = alloc(nb)
    ptrs.append(p)
    w1.append(round(probe(p, nb, 1)))
print("write us at allocation:", w1, flush=True)
w2 = [round(probe(p, nb, 1)) for p in ptrs]
print("write us, second sweep:", w2, flush=True)
r = [round(probe(p, nb, 0)) for p in ptrs[:16]]
print("read us, first 16:", r, flush=True)
order = np.argsort(w2)
q = nb // 4
for tag, sel in (("fastest", order[:3]), ("slowest", order[-3:])):
    for i in sel:
        print(tag, int(i), hex(ptrs[i]), w2[i], "quarters:", [round(probe(ptrs[i] + k * q, q, 1, 4)) for k in range(4)], flush=True)
# free every second buffer and allocate again: do the new ones differ?
for i in range(0, len(ptrs), 2):
    ctx.free(ptrs[i])
w3 = []
for i in range(0, len(ptrs), 2):
    ptrs[i] = alloc(nb)
    w3.append(round(probe(ptrs[i], nb, 1)))
print("write us, re-allocated even ones:", w3, flush=True)
print("same VAs:", sum(1 for i in range(0, len(ptrs), 2)), flush=True)
for p in ptrs:
    ctx.free(p)
ctx.close()
