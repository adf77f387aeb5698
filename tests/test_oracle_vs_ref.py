"""The CPU restatement against the reference's own compiled kernels (oracle/_ref), live, on inputs beyond the fixtures.
Skipped where oracle/_ref was never built (it needs /root/reference at build time; the built .so travels to the GPU box)."""
import numpy as np
import pytest

import cases
import oracle as O

cfgs = cases.cfgs


@pytest.mark.parametrize("cfg", [cfgs.T_MID, cfgs.T_RAGGED.with_(layout=cfgs.GRBG), cfgs.T_BGR.with_(distortion=-0.12)], ids=lambda c: c.name)
@pytest.mark.parametrize("mode", [0, 1])
def test_chain_port_equals_reference(port, ref, cfg, mode):
    raw = cfg.synth(7)
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    res = {}
    for impl in (port, ref):
        planes = impl.raw2quad(raw, cfg.layout, cfg.qw, cfg.qh)
        flat = impl.resample(planes, cfg.layout, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, mode, cfg.fw, cfg.fh)
        ch = impl.chain_from_flat(flat, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, cfg.max_matches)
        ch.update(planes=planes[: 3 if cfg.layout == cfgs.BGR else 4], flat=flat)
        res[impl.name] = ch
    a, b = res["port"], res["reference"]
    for k in ("planes", "flat", "graddot", "sat", "blobcenter", "idx", "counter"):
        assert np.array_equal(a[k], b[k]), k
    assert a["matches"].tobytes() == b["matches"].tobytes()
    assert len(a["matches"]) > 0


@pytest.mark.parametrize("seed", range(4))
def test_random_flat_port_equals_reference(port, ref, seed):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(20, 90)), int(rng.integers(20, 120))
    flat = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    flat[..., 3] = 255
    o, r, R = int(rng.integers(0, 4)), int(rng.integers(2, 8)), int(rng.integers(1, 6))
    thr = float(rng.choice([-1e9, 0.0, 15.0, 500.0]))
    a = port.chain_from_flat(flat, o, r, R, thr, 0.0, 300)
    b = ref.chain_from_flat(flat, o, r, R, thr, 0.0, 300)
    for k in ("graddot", "sat", "blobcenter", "idx", "counter"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    assert a["matches"].tobytes() == b["matches"].tobytes()
    assert np.array_equal(port.blobscore(flat, a["blobcenter"], thr, R), ref.blobscore(flat, b["blobcenter"], thr, R), equal_nan=True)
    assert np.array_equal(port.circularize(a["graddot"], R, r), ref.circularize(b["graddot"], R, r), equal_nan=True)


def test_threads_do_not_change_results(port):
    cfg = cfgs.T_MID
    raw = cfg.synth(3)
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    one = O.Port(threads=1)
    args = (raw, cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh,
            cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold)
    m1, i1, c1 = one.frame(*args)
    m4, i4, c4 = port.frame(*args)
    assert m1.tobytes() == m4.tobytes() and np.array_equal(i1, i4) and np.array_equal(c1, c4)
    m5, i5, c5 = port.frame(*args, sat_mode=1)  # int32 SAT path gives the same list under the guard
    assert m1.tobytes() == m5.tobytes() and np.array_equal(c1, c5)


@pytest.mark.parametrize("shape", [(16, 12), (18, 7), (6, 5), (64, 33)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_colour_kernels_port_equals_reference(port, ref, shape):
    """quad2rgba / quad2nv12 / rgb2nv12 / f2nv12: all layouts, both sampler modes, odd heights, NaN / inf / saturating floats."""
    qw, qh = shape
    rng = np.random.default_rng(qw * 100 + qh)
    planes = rng.integers(0, 256, (4, qh, qw), dtype=np.uint8)
    for layout in (cfgs.RGGB, cfgs.GRBG, cfgs.BGR):
        for mode in (0, 1):
            assert np.array_equal(port.quad2rgba(planes, layout, mode), ref.quad2rgba(planes, layout, mode)), (layout, mode)
            assert np.array_equal(port.quad2nv12(planes, layout, mode), ref.quad2nv12(planes, layout, mode)), (layout, mode)
    rgba = rng.integers(0, 256, (qh, qw, 4), dtype=np.uint8)
    rgba[0, :4] = [[255, 0, 0, 255], [0, 255, 0, 255], [0, 0, 255, 255], [255, 255, 255, 255]]  # chroma extremes
    assert np.array_equal(port.rgba2nv12(rgba), ref.rgba2nv12(rgba))
    f = (rng.standard_normal((qh, qw)) * 200).astype(np.float32)
    f[0, :6] = [np.nan, np.inf, -np.inf, 127.999, -127.0, 128.0]
    assert np.array_equal(port.f2nv12(f), ref.f2nv12(f))


@pytest.mark.parametrize("cfg", [cases.cfgs.T_MID, cases.cfgs.T_BGR, cases.cfgs.T_RAGGED], ids=lambda c: c.name)
def test_threaded_reference_frame_equals_the_sequential_launches(ref, cfg):
    """ref_frame_mt (bench.py's cpu_baseline kind "reference": every NDRange in row bands over host threads) produces the same
    blob records and counters as the per-kernel sequential launches; only the append order may differ (blobList.cl:87)."""
    import oracle as O
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    raw = cfg.synth(5)
    planes = ref.raw2quad(raw, cfg.layout, cfg.qw, cfg.qh)
    flat = ref.resample(planes, cfg.layout, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh)
    bc = ref.blobcenter(ref.sat_vertical(ref.sat_horizontal(ref.graddot(flat, cfg.grad_offset))), cfg.circle_radius)
    m1, _, c1 = ref.bloblist(flat, bc, cfg.circ_threshold, cfg.min_score, cfg.list_radius, cfg.max_matches)
    assert len(m1) > 3
    for threads in (1, 3, 8):
        m2, c2 = ref.frame_mt(raw, cfg.layout, cfg.qw, cfg.qh, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh, cfg.grad_offset,
                              cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, cfg.max_matches, threads)
        assert np.array_equal(c1, c2)
        recs = lambda m: sorted(m.tobytes()[k * 22:(k + 1) * 22] for k in range(len(m)))
        assert recs(m1) == recs(m2)
        if threads == 1:
            assert m1.tobytes() == m2.tobytes()  # one thread = raster order
