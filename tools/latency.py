"""Latency / stream-concurrency / PCIe measurements (tools only; BASELINE.json configs 2-3 and the PCIe-inclusive note of DESIGN.md).
  * single-frame latency: one 2048x1536 frame through vp_frame, host-synchronised per frame
  * n streams in flight: n contexts streams each processing its own frame queue (throughput with batch 1 per stream)
  * PCIe-inclusive: pinned host raw frame -> H2D copy -> vp_frame -> D2H of the 52 KB record, per frame, double-buffered over 2 streams"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.CFG2
mm = cfg.max_matches
NS = 8
ctx = capi.Context(0, n_streams=NS)
p = capi.frame_params(cfg)
raw_h = cfg.synth(0)
bufs = []
for s in range(NS):
    d = dict(raw=ctx.malloc(cfg.raw_bytes), flat=ctx.malloc(cfg.n_flat * 4), sat=ctx.malloc(cfg.n_flat * 4), m=ctx.malloc(22 * mm), idx=ctx.malloc(4 * mm), cnt=ctx.malloc(16))
    ctx.h2d(d['raw'], raw_h, stream=s)
    bufs.append(d)
def frame(s):
    b = bufs[s]
    ctx.frame(p, b['raw'], 1, b['flat'], b['sat'], b['m'], b['cnt'], idx=b['idx'], stream=s)
for s in range(NS):
    for _ in range(3): frame(s)
ctx.sync()
# 1. single-frame latency
N = 200
t0 = time.perf_counter()
for _ in range(N):
    frame(0); ctx.sync(0)
lat = (time.perf_counter() - t0) / N
print(f"single frame, 1 stream, host-synchronised: {lat*1e6:.1f} us/frame ({1/lat:.0f} frames/s)")
# 2. n streams in flight
for ns in (1, 2, 4, 8):
    t0 = time.perf_counter()
    for _ in range(N):
        for s in range(ns): frame(s)
    ctx.sync()
    dt = time.perf_counter() - t0
    print(f"{ns} streams in flight, batch 1 each: {N*ns/dt:.0f} frames/s ({dt/N*1e6:.1f} us per round)")
# 3. PCIe-inclusive, double buffered over 2 streams, pinned host memory
lib = capi.lib()
import ctypes as C
hp = C.c_void_p(); assert lib.vp_malloc_host(ctx.h, C.c_size_t(cfg.raw_bytes), C.byref(hp)) == 0
C.memmove(hp, raw_h.ctypes.data, cfg.raw_bytes)
ho = C.c_void_p(); assert lib.vp_malloc_host(ctx.h, C.c_size_t(22 * mm + 16), C.byref(ho)) == 0
t0 = time.perf_counter()
for i in range(N):
    s = i & 1
    b = bufs[s]
    lib.vp_copy_h2d(ctx.h, s, C.c_void_p(b['raw']), hp, C.c_size_t(cfg.raw_bytes))
    frame(s)
    lib.vp_copy_d2h(ctx.h, s, ho, C.c_void_p(b['m']), C.c_size_t(22 * mm))
    lib.vp_copy_d2h(ctx.h, s, C.c_void_p(ho.value + 22 * mm), C.c_void_p(b['cnt']), C.c_size_t(12))
ctx.sync()
dt = time.perf_counter() - t0
print(f"PCIe-inclusive (3.1 MB H2D + frame + 44 KB D2H per frame, 2 streams): {N/dt:.0f} frames/s")
# 4. PCIe-inclusive, batched: 32 frames per H2D copy + one vp_frame(32) + one D2H of the 32 records, double buffered over 2 streams
BB = 32
big = []
for s in range(2):
    d = dict(raw=ctx.malloc(BB * cfg.raw_bytes), flat=ctx.malloc(BB * cfg.n_flat * 4), sat=ctx.malloc(BB * cfg.n_flat * 4), m=ctx.malloc(BB * 22 * mm), idx=ctx.malloc(BB * 4 * mm), cnt=ctx.malloc(BB * 12 + 4))
    big.append(d)
hpb = C.c_void_p(); assert lib.vp_malloc_host(ctx.h, C.c_size_t(BB * cfg.raw_bytes), C.byref(hpb)) == 0
for i in range(BB):
    C.memmove(hpb.value + i * cfg.raw_bytes, raw_h.ctypes.data, cfg.raw_bytes)
hob = C.c_void_p(); assert lib.vp_malloc_host(ctx.h, C.c_size_t(BB * (22 * mm + 12)), C.byref(hob)) == 0
def round_(s):
    b = big[s]
    lib.vp_copy_h2d(ctx.h, s, C.c_void_p(b['raw']), hpb, C.c_size_t(BB * cfg.raw_bytes))
    ctx.frame(p, b['raw'], BB, b['flat'], b['sat'], b['m'], b['cnt'], idx=b['idx'], stream=s)
    lib.vp_copy_d2h(ctx.h, s, hob, C.c_void_p(b['m']), C.c_size_t(BB * 22 * mm))
    lib.vp_copy_d2h(ctx.h, s, C.c_void_p(hob.value + BB * 22 * mm), C.c_void_p(b['cnt']), C.c_size_t(BB * 12))
for s in range(2): round_(s)
ctx.sync()
R = 40
t0 = time.perf_counter()
for i in range(R): round_(i & 1)
ctx.sync()
dt = time.perf_counter() - t0
print(f"PCIe-inclusive, batched ({BB} frames per copy and launch group, 2 streams): {R*BB/dt:.0f} frames/s ({BB*cfg.raw_bytes*R/dt/1e9:.1f} GB/s host-to-device)")
