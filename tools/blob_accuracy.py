"""Position accuracy of the blob list against the synthetic ground truth -- the counterpart of the reference's blob_benchmark
(src/blob_benchmark.cpp:44-84: offset of the detected peak from the true blob centre, in field mm).  GPU, tools only.
usage: blob_accuracy.py [FRAMES]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _vp_pkg import load
vp = load(); capi = vp.load_capi()
cfg = vp.configs.ALL[os.environ.get("VP_CFG", "cfg2_2048x1536_rggb")]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mm = cfg.max_matches


def accuracy(cfg, matches_per_frame, seeds):
    """matches_per_frame: list of structured blob arrays (x, y in flat px).  Returns the statistics dict."""
    fs, ox, oy = np.float32(cfg.field_scale), np.float32(cfg.off_x), np.float32(cfg.off_y)
    errs, found, total, extra = {"centre": [], "side": [], "ball": []}, 0, 0, 0
    offsets = []
    for m, seed in zip(matches_per_frame, seeds):
        truth = cfg.truth(seed)
        px, py = m["x"] * fs + ox, m["y"] * fs + oy  # Perspective::flat2field
        ok = np.isfinite(px) & np.isfinite(py)
        used = np.zeros(len(m), bool)
        for i, (tx, ty, tr, *_rgb) in enumerate(truth):
            total += 1
            d = np.where(ok, np.hypot(px - tx, py - ty), np.inf)
            j = int(np.argmin(d)) if len(d) else -1
            if j >= 0 and d[j] < tr:  # the peak lies inside the disc
                found += 1
                used[j] = True
                kind = "ball" if i >= 5 * cfg.n_robots else ("centre" if i % 5 == 0 else "side")
                errs[kind].append(float(d[j]))
                offsets.append((float(px[j] - tx), float(py[j] - ty)))
        extra += int((~used).sum())
    out = {"frames": len(seeds), "discs": total, "found": found, "blobs_not_on_a_disc_per_frame": extra / max(len(seeds), 1)}
    if offsets:
        # systematic offset (what blob_benchmark accumulates as offsetSum) and the scatter that is left once it is removed
        o = np.asarray(offsets)
        res = o - o.mean(0)
        out["mean_offset_px"] = [float(v) for v in o.mean(0) / fs]
        out["residual_rms_px"] = float(np.sqrt((res ** 2).sum(1).mean()) / fs)
        out["residual_max_px"] = float(np.sqrt((res ** 2).sum(1)).max() / fs)
    for k, v in errs.items():
        v = np.asarray(v)
        out[k] = {"n": len(v), "mean_mm": float(v.mean()) if len(v) else None, "rms_mm": float(np.sqrt((v ** 2).mean())) if len(v) else None, "max_mm": float(v.max()) if len(v) else None}
    return out


if __name__ == "__main__":
    ctx = capi.Context(0, stream_ptr=torch.cuda.current_stream().cuda_stream)
    params = capi.frame_params(cfg, sat_variant=capi.SAT_TILED, order=capi.ORDER_RASTER)
    seeds = list(range(500, 500 + B))
    d_raw = torch.from_numpy(np.stack([cfg.synth(s) for s in seeds])).cuda()
    flat = torch.empty((B, cfg.fh, cfg.fw, 4), dtype=torch.uint8, device="cuda")
    sat = torch.empty((B, cfg.fh, cfg.fw), dtype=torch.int32, device="cuda")
    matches = torch.zeros((B, 22 * mm), dtype=torch.uint8, device="cuda")
    idx = torch.zeros((B, mm), dtype=torch.int32, device="cuda")
    counter = torch.zeros((B, 3), dtype=torch.int32, device="cuda")
    ctx.frame(params, d_raw, B, flat, sat, matches, counter, idx=idx)
    ctx.sync()
    cnt = counter.cpu().numpy()
    mh = matches.cpu().numpy()
    per = [np.frombuffer(mh[i].tobytes(), capi.MATCH_DTYPE)[: min(mm, cnt[i, 0])] for i in range(B)]
    import json
    print(json.dumps({"workload": cfg.name, "mm_per_px": cfg.field_scale, **accuracy(cfg, per, seeds)}, indent=1))
