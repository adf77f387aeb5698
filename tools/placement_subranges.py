"""Is the write rate a property of a whole hipMalloc allocation or of the physical region behind each part of it?  Allocates a few LARGE blocks (4 x 1.6 GB
each), probes the zero-fill rate of every 1.6 GB quarter separately (vp_probe_stream: a pure 16-byte-per-lane nt write stream, far larger than the 256 MiB
Infinity Cache), then the same for 1.6 GB blocks allocated one by one.  GPU; tools only.  Prints JSON."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
ctx = capi.Context(0, n_streams=1)
lib = capi.lib()
Q = 128 * 2048 * 1536 * 4  # 1.6 GB: the table of the bench launch


def rate(ptr, nbytes, write=1):
    us = C.c_float()
    assert lib.vp_probe_stream(ctx.h, 0, C.c_void_p(ptr), C.c_size_t(nbytes), write, 3, C.byref(us)) == 0, lib.vp_last_error()
    return round(nbytes / us.value * 1e-3)


out = {"quarter_bytes": Q, "large_blocks_4_quarters_each_GBs": [], "whole_large_block_GBs": [], "single_blocks_GBs": []}
big = []
for _ in range(6):
    p = ctx.malloc(4 * Q)
    big.append(p)
    out["large_blocks_4_quarters_each_GBs"].append([rate(p + k * Q, Q) for k in range(4)])
    out["whole_large_block_GBs"].append(rate(p, 4 * Q))
small = []
for _ in range(12):
    p = ctx.malloc(Q)
    small.append(p)
    out["single_blocks_GBs"].append(rate(p, Q))
out["read_GBs_first_large_block_quarters"] = [rate(big[0] + k * Q, Q, 0) for k in range(4)]
# second look at the first large block: is the figure stable per quarter?
out["large_block_0_again"] = [rate(big[0] + k * Q, Q) for k in range(4)]
for p in big + small:
    ctx.free(p)
ctx.close()
print(json.dumps(out))
