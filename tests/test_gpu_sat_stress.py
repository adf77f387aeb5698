"""Stress of the single-pass SAT's inter-workgroup carry hand-off (cdna guide: test every hand-off under UNEVEN load, with
every word checked): many launches over changing shapes / batch sizes / offsets, a second stream hammering HBM meanwhile,
every SAT element compared with an exact integer reference computed on the device by torch (cumsum in int64)."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def test_sat_lookback_under_uneven_load():
    import torch
    capi = cases.vp.load_capi()
    ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(1234)
    side = torch.cuda.Stream()
    junk_a = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    junk_b = torch.empty_like(junk_a)
    shapes = [(2048, 1536), (640, 384), (192, 128), (150, 98), (1024, 64), (64, 1024), (4096, 192), (320, 320), (2048, 1536)]
    launches = 0
    iters = int(os.environ.get("VP_SOAK_ITERS", "60"))  # soak runs: VP_SOAK_ITERS=2000 also draws the large batches of the benchmark
    big = [1, 2, 4, 9] if iters <= 60 else [1, 2, 4, 9, 24, 64, 128]
    try:
        for it in range(iters):
            w, h = shapes[it % len(shapes)]
            batch = int(rng.choice([1, 2, 3, 5, 8, 16])) if w * h < 1 << 20 else int(rng.choice(big))
            o = int(rng.integers(0, 5))
            flat = torch.randint(0, 256, (batch, h, w, 4), dtype=torch.uint8, device="cuda")
            flat[..., 3] = 255
            if it % 3 == 0:  # a background copy storm on another stream: uneven load while the carries are handed over
                with torch.cuda.stream(side):
                    for _ in range(4):
                        junk_b.copy_(junk_a, non_blocking=True)
            sat = torch.empty((batch, h, w), dtype=torch.int32, device="cuda")
            gd = torch.empty((batch, h, w), dtype=torch.float32, device="cuda")
            strips = None
            nb = ctx.halo_strip_bytes(w, h, o)
            if nb and it % 2 == 0:
                strips = torch.empty((batch, nb // 4), dtype=torch.int32, device="cuda")
                ctx.halo_strips(flat, w, h, o, batch, strips)
                ctx.graddot_sat_strips(flat, strips, w, h, o, batch, capi.SAT_TILED, sat, gd)
            else:
                ctx.graddot_sat(flat, w, h, o, batch, capi.SAT_TILED, sat, gd)
            # run the same launch again straight away: granules of the previous epoch are still in memory
            sat2 = torch.empty_like(sat)
            ctx.graddot_sat(flat, w, h, o, batch, capi.SAT_TILED, sat2, None)
            ctx.sync()  # raises VpError(VP_ERR_TIMEOUT) if a bounded wait expired
            want = gd.to(torch.int64).cumsum(1).cumsum(2).to(torch.int32)
            assert torch.equal(sat, want), f"iteration {it}: {w}x{h} batch {batch} o {o}"
            assert torch.equal(sat2, want), f"iteration {it} (relaunch): {w}x{h} batch {batch} o {o}"
            # gradDot itself against the two-pass variant
            sat3 = torch.empty_like(sat)
            ctx.graddot_sat(flat, w, h, o, batch, capi.SAT_TWOPASS, sat3, None)
            ctx.sync()
            assert torch.equal(sat3, want)
            launches += 3
        side.synchronize()
    finally:
        ctx.close()
    assert launches == 3 * iters
