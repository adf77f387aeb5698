"""Shared definitions of the parity cases: used by tests/golden/make_golden.py (generator, needs oracle/_ref) and by the tests.

Two families:
* chain cases  -- a synthetic raw sensor frame pushed through the whole chain (raw2quad .. blobList);
* flat cases   -- hand-made `flat` (and sometimes `blobcenter`) inputs exercising the edge cases of SURVEY 8(c):
                  blobs on every border (clamp-to-edge, H7), plateau peaks, uniform disc (inf / NaN score, H5),
                  blob-list overflow, o = 0, and a frame violating the 2^24 SAT exactness guard (H2).
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from _vp_pkg import load  # noqa: E402

vp = load()
cfgs = vp.configs
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

CHAIN_CASES = [(c, mode, 1) for c in (cfgs.T_RGGB, cfgs.T_GRBG, cfgs.T_BGR, cfgs.T_RAGGED) for mode in (0, 1)]


def chain_case_name(cfg, mode, seed):
    return f"chain_{cfg.name}_s{mode}_seed{seed}"


def _disc(img, cx, cy, r, color):
    h, w = img.shape[:2]
    yy, xx = np.mgrid[0:h, 0:w]
    m = (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r
    img[m, :3] = color


def _base(w, h, seed, noise=2):
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w, 4), np.uint8)
    img[..., :3] = np.clip(np.array([110, 140, 120]) + rng.integers(-noise, noise + 1, (h, w, 3)), 0, 255)
    img[..., 3] = 255
    return img


def flat_cases():
    """name -> dict(flat, o, r, R, thr, min_score, max_matches[, blobcenter])."""
    cases = {}

    # discs centred on / touching all four borders and corners
    f = _base(96, 72, 11)
    for (cx, cy) in [(0, 0), (95, 0), (0, 71), (95, 71), (48, 0), (48, 71), (0, 36), (95, 36), (3, 20), (92, 50), (30, 30), (60, 40)]:
        _disc(f, cx, cy, 4.5, (200, 60, 130))
    cases["flat_borders"] = dict(flat=f, o=2, r=5, R=4, thr=15.0, min_score=0.0, max_matches=2000)

    # noise-free symmetric discs on a perfectly uniform background: the R-disc around a peak is uniform => stddev 0 =>
    # score = inf; exact symmetry gives 2x2 / 4x4 plateaus of equal blobCenter (strict '>' keeps all, blobList.cl:47-52)
    # and the sub-pixel formula divides 0 by 0 on them => NaN x / y (blobList.cl:93-94)
    f = np.zeros((64, 80, 4), np.uint8)
    f[..., :3] = (100, 150, 120)
    f[..., 3] = 255
    _disc(f, 20, 20, 5.0, (220, 40, 140))
    _disc(f, 55, 40, 5.5, (40, 230, 100))
    _disc(f, 40.5, 20.5, 5.0, (10, 10, 240))
    f[50:54, 10:14, :3] = (250, 250, 250)
    cases["flat_uniform_discs"] = dict(flat=f, o=1, r=6, R=2, thr=1.0, min_score=0.0, max_matches=2000)
    cases["flat_uniform_discs_o2"] = dict(flat=f.copy(), o=2, r=6, R=3, thr=1.0, min_score=0.0, max_matches=2000)

    # threshold 0 on a constant image: gradDot == SAT == blobCenter == 0 everywhere, every pixel is a plateau "peak",
    # score = 0/0 = NaN (kept: NaN < minScore is false), x/y = NaN; count far above max_matches (overflow, main.cpp:301,315)
    f = np.zeros((24, 32, 4), np.uint8)
    f[..., :3] = (90, 90, 90)
    f[..., 3] = 255
    cases["flat_constant_overflow"] = dict(flat=f, o=2, r=5, R=4, thr=0.0, min_score=0.0, max_matches=50)

    # o = 0 (fieldScale > 12.5 mm/px): gradDot == 0, nothing above the default threshold
    f = _base(48, 40, 5, noise=20)
    _disc(f, 24, 20, 6, (230, 60, 60))
    cases["flat_offset0"] = dict(flat=f, o=0, r=2, R=1, thr=15.0, min_score=0.0, max_matches=2000)

    # many weak blobs + a tiny list: counter[0] > max_matches; also a positive min_score so counter[1] moves
    f = _base(128, 96, 21, noise=6)
    rng = np.random.default_rng(3)
    for _ in range(40):
        _disc(f, int(rng.integers(4, 124)), int(rng.integers(4, 92)), 4.3, tuple(int(v) for v in rng.integers(30, 250, 3)))
    cases["flat_overflow_minscore"] = dict(flat=f, o=1, r=4, R=4, thr=15.0, min_score=2.0, max_matches=7)

    # negative min_score: nothing is score-rejected except NaN-free finite negatives; exercises the compare direction
    cases["flat_negative_minscore"] = dict(flat=f.copy(), o=1, r=4, R=4, thr=-1e30, min_score=-1.0, max_matches=100000)

    # SAT exactness guard violated: gx = gy > 0 everywhere => SAT grows past 2^24 (H2).  Reference fp32 SAT loses integers.
    h, w = 160, 224
    yy, xx = np.mgrid[0:h, 0:w]
    f = np.zeros((h, w, 4), np.uint8)
    ramp = np.clip(((xx + yy) * 5) % 512, 0, 511)
    ramp = np.where(ramp > 255, 511 - ramp, ramp).astype(np.uint8)
    f[..., 0] = ramp
    f[..., 1] = ramp[::-1, ::-1]
    f[..., 2] = 128
    f[..., 3] = 255
    _disc(f, 100, 80, 7, (250, 20, 200))
    cases["flat_guard_violation"] = dict(flat=f, o=2, r=5, R=4, thr=15.0, min_score=0.0, max_matches=2000)
    return cases


def load_golden(name):
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}
