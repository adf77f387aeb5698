"""Host-thread scaling of the CPU hypothesis stage (vph_detect_batch) on one cfg2 blob list (tools/data/cfg2_blobs_seed1.npy:
135 CLMatch records of synthetic frame seed 1 from the CPU oracle).  No GPU needed."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _vp_pkg import load  # noqa: E402

vp = load()
H = vp.hostlib
cfg = vp.configs.ALL["cfg2_2048x1536_rggb"]
mm = 2000
mb = np.load(os.path.join(ROOT, "tools", "data", "cfg2_blobs_seed1.npy"))
n = len(mb) // 22
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dets = []
for i in range(B):
    d = H.Detector()
    d.set_geometry_direct(H.nadir_calib(cfg.model()), (cfg.qw, cfg.qh), cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh, field=(int(cfg.field_len), int(cfg.field_wid), 300, -1, 10))
    dets.append(d)
matches = np.zeros((B, 22 * mm), np.uint8)
matches[:, : len(mb)] = mb
counter = np.zeros((B, 3), np.int32)
counter[:, 0] = n
pk = np.zeros((B, 4092), np.uint8)
ln = np.zeros(B, np.int32)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
k = 0
for th in (1, 2, 4, 8, 16, 32, 64):
    best = 1e9
    for rep in range(3):
        k += 1
        t0 = time.perf_counter()
        H.detect_batch(dets, matches, counter, mm, k, k / 70, k / 70, k / 70 + 0.002, th, pk, ln)
        best = min(best, time.perf_counter() - t0)
    print(f"threads {th:3d}: {best * 1e3:8.2f} ms per {B} frames = {best * 1e3 / B:.3f} ms/frame, {B / best:9.0f} frames/s")
