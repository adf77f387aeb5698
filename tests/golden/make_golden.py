#!/usr/bin/env python3
"""Generates tests/golden/*.npz from oracle/_ref -- the reference's own kernel/*.cl compiled unmodified for x86
(oracle/Makefile `ref`) and driven by the NDRange loops of oracle/ref_shim.c.

Run in the build container (needs /root/reference to build oracle/_ref):
    make -C oracle ref && python tests/golden/make_golden.py

The fixtures are DATA (inputs, parameters, expected outputs); no reference source text is stored.
Each .npz records the sampler mode used for integer-linear reads and max|SAT| for the 2^24 guard (H2).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import cases  # noqa: E402
import oracle as O  # noqa: E402


def exact_i32(a, what):
    assert np.all(a == np.rint(a)) and np.all(np.abs(a) < 2 ** 31), what
    return a.astype(np.int32)


def main():
    O.build(ref=True)
    ref = O.Ref()
    os.makedirs(cases.GOLDEN_DIR, exist_ok=True)
    total = 0

    for cfg, mode, seed in cases.CHAIN_CASES:
        raw = cfg.synth(seed)
        model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
        planes = ref.raw2quad(raw, cfg.layout, cfg.qw, cfg.qh)
        flat = ref.resample(planes, cfg.layout, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, mode, cfg.fw, cfg.fh)
        ch = ref.chain_from_flat(flat, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, cfg.max_matches)
        max_sat = float(np.abs(ch["sat"]).max())
        assert max_sat < 2 ** 24, "synthetic chain inputs must satisfy the SAT exactness guard"
        out = dict(
            raw=raw, model=model.to_array(), layout=np.int32(cfg.layout), qw=np.int32(cfg.qw), qh=np.int32(cfg.qh), fw=np.int32(cfg.fw), fh=np.int32(cfg.fh),
            scalars=np.array([cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.circ_threshold, cfg.min_score], np.float32),
            ints=np.array([cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.max_matches, mode], np.int32),
            planes=planes, flat=flat, graddot=exact_i32(ch["graddot"], "graddot"), sat=exact_i32(ch["sat"], "sat"), blobcenter=ch["blobcenter"],
            matches=np.frombuffer(ch["matches"].tobytes(), np.uint8), idx=ch["idx"], counter=ch["counter"], max_abs_sat=np.float64(max_sat),
        )
        # debug / stream colour kernels: quad2rgba, quad2nv12, rgb2nv12, f2nv12 (NV12 needs an even width)
        out["quad_rgba"] = ref.quad2rgba(planes, cfg.layout, mode)
        if cfg.qw % 2 == 0:
            out["quad_nv12"] = ref.quad2nv12(planes, cfg.layout, mode)
        if cfg.fw % 2 == 0:
            out["flat_nv12"] = ref.rgba2nv12(flat)
            out["blobcenter_nv12"] = ref.f2nv12(ch["blobcenter"])
        if cfg is cases.cfgs.T_BGR and mode == 1:
            # the two kernels the reference names but never launches: blobScore.cl, blobCenter.cl
            out["blobscore"] = ref.blobscore(flat, ch["blobcenter"], cfg.circ_threshold, cfg.list_radius)
            out["circularize"] = ref.circularize(ch["graddot"], cfg.list_radius, cfg.circle_radius)
        name = cases.chain_case_name(cfg, mode, seed)
        path = os.path.join(cases.GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        total += os.path.getsize(path)
        print(f"{name}: blobs={len(ch['matches'])} counter={ch['counter'].tolist()} max|SAT|={max_sat:.0f} ({os.path.getsize(path) // 1024} KiB)")

    for name, c in cases.flat_cases().items():
        flat = c["flat"]
        g = ref.graddot(flat, c["o"])
        sat_h = ref.sat_horizontal(g)
        sat = ref.sat_vertical(sat_h)
        bc = ref.blobcenter(sat, c["r"])
        m, idx, cnt = ref.bloblist(flat, bc, c["thr"], c["min_score"], c["R"], c["max_matches"])
        max_sat = float(np.abs(sat).max())
        out = dict(
            flat=flat, ints=np.array([c["o"], c["r"], c["R"], c["max_matches"]], np.int32), scalars=np.array([c["thr"], c["min_score"]], np.float32),
            graddot=exact_i32(g, "graddot"), blobcenter=bc, matches=np.frombuffer(m.tobytes(), np.uint8), idx=idx, counter=cnt, max_abs_sat=np.float64(max_sat),
        )
        if max_sat < 2 ** 24:
            out["sat"] = exact_i32(sat, "sat")
        else:
            out["sat_f32"] = sat  # the reference's own (inexact) binary32 table, kept to pin the port's fp32 path
        path = os.path.join(cases.GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        total += os.path.getsize(path)
        print(f"{name}: blobs={len(m)} counter={cnt.tolist()} max|SAT|={max_sat:.0f} ({os.path.getsize(path) // 1024} KiB)")
    # BASELINE config 0 at its own size: one 640x480 BGR frame (the OpenCV-driver layout) through the reference's kernels, both sampler
    # modes.  Compact form: the input frame, the blob list and counters, and SHA-256 digests of every intermediate image.
    import hashlib
    cfg, seed = cases.cfgs.CFG1, 7
    raw = cfg.synth(seed)
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    out = dict(raw=raw, model=model.to_array(), layout=np.int32(cfg.layout), qw=np.int32(cfg.qw), qh=np.int32(cfg.qh), fw=np.int32(cfg.fw), fh=np.int32(cfg.fh),
               scalars=np.array([cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, cfg.circ_threshold, cfg.min_score], np.float32),
               ints=np.array([cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.max_matches], np.int32))
    digest = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
    planes = ref.raw2quad(raw, cfg.layout, cfg.qw, cfg.qh)
    out["sha_planes"] = digest(planes[:3])
    for mode in (0, 1):
        flat = ref.resample(planes, cfg.layout, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, mode, cfg.fw, cfg.fh)
        ch = ref.chain_from_flat(flat, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, cfg.max_matches)
        assert float(np.abs(ch["sat"]).max()) < 2 ** 24
        out[f"m{mode}_matches"] = np.frombuffer(ch["matches"].tobytes(), np.uint8)
        out[f"m{mode}_idx"], out[f"m{mode}_counter"] = ch["idx"], ch["counter"]
        out[f"m{mode}_sha_flat"], out[f"m{mode}_sha_graddot"] = digest(flat), digest(exact_i32(ch["graddot"], "graddot"))
        out[f"m{mode}_sha_sat"], out[f"m{mode}_sha_blobcenter"] = digest(exact_i32(ch["sat"], "sat")), digest(ch["blobcenter"])
        print(f"config0 {cfg.name} mode {mode}: blobs={len(ch['matches'])} counter={ch['counter'].tolist()}")
    path = os.path.join(cases.GOLDEN_DIR, "config0_640x480_bgr.npz")
    np.savez_compressed(path, **out)
    total += os.path.getsize(path)
    print(f"total {total // 1024} KiB")


if __name__ == "__main__":
    main()
