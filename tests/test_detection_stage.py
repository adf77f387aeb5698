"""Blob list -> SSL_DetectionFrame: the C++ stage in libvp_host.so against oracle/detect_oracle.py.

PARITY UNPINNED (reference host code needs Eigen / protobuf, absent here; the reference has no test for this stage): these
tests hold two independent restatements of the cited reference lines to each other bit for bit, and check domain
properties (rendered robots come back with their ids, poses and team colours).
"""
import math

import numpy as np
import pytest

import cases
import detect_oracle as O

H = cases.vp.hostlib
F = np.float32

NADIR = dict(calib=(1200.0, 512.0, 384.0, 0.08, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 5000.0), size=(1024, 768))
# tilted camera: rotation about x by ~170 deg plus a small twist (unit quaternion up to rounding; the model normalises it)
TILTED = dict(calib=(1100.0, 500.0, 390.0, -0.05, -0.9952, 0.03, 0.02, 0.0900, 150.0, -80.0, 4200.0), size=(1024, 768))

FIELD = (9000, 6000, 300, -1, 10)
LINES = [(-4500, -3000, 4500, -3000), (-4500, 3000, 4500, 3000), (0, -3000, 0, 3000), (-4500, -3000, -4500, 3000), (4500, -3000, 4500, 3000)]
ARCS = [(0, 0, 500, 0, 2 * math.pi)]

COLORS = {"yellow": (230, 150, 20), "blue": (20, 110, 235), "green": (30, 230, 120), "pink": (240, 30, 140), "orange": (200, 120, 50), "field": (120, 125, 118),
          "white": (230, 230, 225)}


def scene(seed, n_robots=6, n_balls=3, n_clutter=6, jitter=1.5, extent=(1500.0, 1000.0), dropout=0.0):
    """Blob list in field coordinates of robots (5 discs each), balls and clutter, plus the ground truth."""
    rng = np.random.default_rng(seed)
    recs, truth = [], []

    def blob(x, y, col, score=40.0):
        c = np.clip(np.array(col) + rng.integers(-12, 13, 3), 0, 255)
        ctr = np.clip(np.array(col) + rng.integers(-12, 13, 3), 0, 255)
        recs.append((F(x), F(y), tuple(int(v) for v in c), tuple(int(v) for v in ctr), F(rng.uniform(20, 60)), F(score)))

    centres = []
    while len(centres) < n_robots:
        p = rng.uniform(-1, 1, 2) * np.array(extent)
        if all(np.hypot(*(p - q)) > 260 for q in centres):
            centres.append(p)
    for i, p in enumerate(centres):
        ident = int(rng.integers(0, 16))
        team = i % 2
        ang = rng.uniform(-math.pi, math.pi)
        truth.append(dict(id=ident + 16 * team, x=p[0], y=p[1], w=ang))
        for b in range(5):
            if b and rng.uniform() < dropout:
                continue
            px, py = [float(v) for v in O.PATTERN_POS[b]]
            x = p[0] + math.cos(ang) * px - math.sin(ang) * py + rng.normal(0, jitter)
            y = p[1] + math.sin(ang) * px + math.cos(ang) * py + rng.normal(0, jitter)
            col = COLORS["blue" if team else "yellow"] if b == 0 else COLORS["green" if (O.PATTERNS[ident] >> (4 - b)) & 1 else "pink"]
            blob(x, y, col)
    balls = []
    while len(balls) < n_balls:
        p = rng.uniform(-1, 1, 2) * np.array(extent)
        if all(np.hypot(*(p - q)) > 200 for q in centres + balls):
            balls.append(p)
            blob(p[0], p[1], COLORS["orange"], score=rng.uniform(8, 60))
    for _ in range(n_clutter):
        p = rng.uniform(-1, 1, 2) * np.array(extent)
        blob(p[0], p[1], COLORS[("field", "white", "orange")[int(rng.integers(0, 3))]], score=rng.uniform(0, 12))
    order = rng.permutation(len(recs))  # kd-tree shape depends on the insertion order
    recs = [recs[i] for i in order]
    arr = np.zeros(len(recs), H.VPH_MATCH)
    for i, (x, y, c, ctr, circ, score) in enumerate(recs):
        arr[i] = (x, y, c, ctr, circ, score)
    return arr, truth, [tuple(b) for b in balls]


def to_oracle_matches(arr):
    return [dict(pos=(F(m["x"]), F(m["y"])), color=tuple(int(v) for v in m["color"]), center=tuple(int(v) for v in m["center"]), circ=F(m["circ"]), score=F(m["score"])) for m in arr]


def make_pair(cam=NADIR, **cfg_over):
    det_cfg = H.config_defaults()
    for k, v in cfg_over.items():
        setattr(det_cfg, k, v)
    d = H.Detector(det_cfg)
    d.set_geometry(cam["calib"], cam["size"], cam["size"], FIELD, lines=LINES, arcs=ARCS)
    ocam = O.Camera(cam["calib"], cam["size"], cam["size"])
    r = O.Res(ocam, field_length=FIELD[0], field_width=FIELD[1], boundary_width=FIELD[2], boundary_goal_line=FIELD[3], line_thickness=FIELD[4], lines=LINES, arcs=ARCS)
    r.cam_id = det_cfg.cam_id
    for k, v in cfg_over.items():
        if k in ("min_confidence", "clipping_tolerance", "geometry_tolerance", "reference_force", "history_force"):
            setattr(r, k, F(v))
        elif k != "cam_id":
            setattr(r, k, v)
    return d, r


def bits(v):
    return np.asarray(v, np.float32).view(np.uint32)


def assert_frames_equal(got, want):
    assert got["frame_number"] == want["frame_number"] and got["camera_id"] == want["camera_id"]
    assert got["t_capture"] == want["t_capture"] and got["t_capture_camera"] == want["t_capture_camera"]
    for key, fields in (("balls", ("confidence", "x", "y", "pixel_x", "pixel_y")), ("robots_yellow", ("confidence", "x", "y", "orientation", "pixel_x", "pixel_y", "height")),
                        ("robots_blue", ("confidence", "x", "y", "orientation", "pixel_x", "pixel_y", "height"))):
        assert len(got[key]) == len(want[key]), (key, len(got[key]), len(want[key]))
        for i, w in enumerate(want[key]):
            g = got[key][i]
            if "robot_id" in w:
                assert int(g["robot_id"]) == int(w["robot_id"]), (key, i)
            for f in fields:
                assert bits(g[f]) == bits(w[f]), (key, i, f, float(g[f]), float(w[f]))  # bit-exact binary32
    assert np.array_equal(bits(got["t_offsets"]), bits([F(o) for o in want["t_offsets"]]))


def assert_colors_equal(d, r):
    c = d.colors()
    for i, k in enumerate(("orange", "field", "yellow", "blue", "green", "pink", "fieldLine")):
        assert tuple(int(v) for v in c[i]) == tuple(int(v) for v in r.colors[k]), k


def assert_tracked_equal(d, r, cam):
    got = d.get_tracked(cam)
    want = r.tracked.get(cam, [])
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert int(g["id"]) == w["id"] and int(g["age"]) == w["age"] and float(g["timestamp"]) == w["timestamp"]
        for f in ("x", "y", "z", "w", "vx", "vy", "vz", "vw", "confidence"):
            a, b = bits(g[f]), bits(w[f])
            assert a == b or (np.isnan(g[f]) and np.isnan(w[f])), (f, float(g[f]), float(w[f]))


# ---- building blocks ----------------------------------------------------------------------------------------------------
def test_atan2_fast_matches_the_oracle_and_libm():
    rng = np.random.default_rng(1)
    pts = list(rng.normal(0, 100, (4000, 2)).astype(np.float32)) + [(0, 0), (0, 1), (1, 0), (-1, 0), (0, -1), (-0.0, -1), (1e-30, -1e30), (3, 3), (-3, 3)]
    worst = 0.0
    for y, x in pts:
        g, w = H.atan2_fast(y, x), O.atan2_fast(y, x)
        assert bits(g) == bits(w), (y, x)
        if x != 0 or y != 0:
            worst = max(worst, abs(float(g) - math.atan2(float(y), float(x))))
    assert worst < 2e-5  # the polynomial's published error bound is ~1e-5 rad


def test_kdtree_range_search_order_and_content():
    rng = np.random.default_rng(2)
    pts = rng.uniform(-500, 500, (300, 2)).astype(np.float32)
    pts[17] = pts[3]  # duplicates go right
    tree = O.kd_build([(F(x), F(y)) for x, y in pts])
    for q in range(40):
        c = pts[q] + rng.normal(0, 5, 2).astype(np.float32)
        rad = F(rng.uniform(10, 200))
        got = H.kdtree_range(pts, c, rad)
        want = O.kd_range(tree, [(F(x), F(y)) for x, y in pts], (F(c[0]), F(c[1])), rad, [])
        assert got == want  # same members in the same (pre-order) sequence
        d = np.sqrt(((pts - c) ** 2).sum(1, dtype=np.float32))
        assert sorted(got) == sorted(np.nonzero(d <= rad)[0].tolist())
    # NaN positions are inserted (to the right) and never returned
    pts[5] = (np.nan, np.nan)
    assert 5 not in H.kdtree_range(pts, (0, 0), 1e9)
    assert H.kdtree_range(pts, (np.nan, 0), 1e9) == []


def test_kmeans_against_the_oracle():
    rng = np.random.default_rng(3)
    n_true = 0
    for _ in range(300):
        n = int(rng.integers(0, 9))
        a, b = rng.integers(0, 256, 3), rng.integers(0, 256, 3)
        vals = [tuple(int(v) for v in np.clip((a if rng.uniform() < 0.5 else b) + rng.integers(-15, 16, 3), 0, 255)) for _ in range(n)]
        contrast = tuple(int(v) for v in rng.integers(0, 256, 3))
        c1, c2 = tuple(int(v) for v in rng.integers(0, 256, 3)), tuple(int(v) for v in rng.integers(0, 256, 3))
        got = H.kmeans(contrast, vals, c1, c2)
        want = O.kmeans(contrast, vals, c1, c2)
        assert got == (want[0], tuple(want[1]), tuple(want[2]))
        n_true += got[0]
    assert 20 < n_true < 280  # both outcomes exercised
    # fewer than two values, identical seeds: centres untouched
    assert H.kmeans((0, 0, 0), [(5, 5, 5)], (1, 2, 3), (4, 5, 6)) == (False, (1, 2, 3), (4, 5, 6))
    assert H.kmeans((0, 0, 0), [(5, 5, 5), (5, 5, 5)], (1, 2, 3), (4, 5, 6)) == (False, (1, 2, 3), (4, 5, 6))


# ---- whole stage ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cam", [NADIR, TILTED], ids=["nadir", "tilted"])
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_single_frame_matches_the_oracle(cam, seed):
    d, r = make_pair(cam)
    arr, truth, balls = scene(seed)
    got = d.detect(arr, 7, 12.5, 3.25)
    want = O.detect_frame(r, to_oracle_matches(arr), 7, 12.5, 3.25)
    assert_frames_equal(got, want)
    assert_colors_equal(d, r)
    assert len(got["robots_yellow"]) + len(got["robots_blue"]) >= len(truth) - 1


def test_rendered_robots_come_back_with_id_pose_and_team():
    d, _ = make_pair(NADIR)
    arr, truth, balls = scene(21, n_robots=8, n_balls=4, n_clutter=10, jitter=1.0)
    fr = d.detect(arr, 1, 0.5)
    found = [(int(b["robot_id"]) + 16 * t, float(b["x"]), float(b["y"]), float(b["orientation"]), float(b["confidence"])) for t, key in enumerate(("robots_yellow", "robots_blue"))
             for b in fr[key]]
    assert len(found) == len(truth)
    cam = O.Camera(NADIR["calib"], NADIR["size"], NADIR["size"])
    for t in truth:
        # blobs live on the plane z = maxBotHeight (180); the detection is re-projected to the team's robot height (143)
        img = cam.field2image((t["x"], t["y"], 180.0))
        ex, ey, _ = cam.image2field(img, 143.0)
        m = min(found, key=lambda f: math.hypot(f[1] - ex, f[2] - ey))
        assert m[0] == t["id"]
        assert math.hypot(m[1] - ex, m[2] - ey) < 3.0
        assert abs(math.remainder(m[3] - t["w"], 2 * math.pi)) < 0.05
        assert m[4] > 0.5
    assert len(fr["balls"]) >= len(balls) - 1
    for b in fr["balls"]:
        assert 0.2 < b["confidence"] <= 1.0


def test_tracked_sequence_matches_the_oracle():
    """Three frames of moving robots: frame n's detections feed the tracker (own packets loop back through the multicast
    group, udpsocket.cpp:146-154), frame n+1 builds tracked hypotheses from them, including ones with missing side blobs."""
    d, r = make_pair(NADIR, cam_id=2)
    rng = np.random.default_rng(5)
    arr0, truth, _ = scene(31, n_robots=5, n_balls=2, n_clutter=4)
    vel = rng.normal(0, 600, (len(arr0), 2))
    n_tracked_bots = 0
    for f in range(3):
        t = 1.0 + f / 60.0
        arr = arr0.copy()
        arr["x"] = (arr0["x"] + vel[:, 0] * 0).astype(np.float32) + F(f * 4.0)  # rigid drift keeps the patterns intact
        arr["y"] = arr0["y"] - F(f * 2.5)
        if f == 2:
            arr = np.delete(arr, [0, 3])  # lose two blobs: tracked hypotheses with nullptr members
        d.set_offsets([0.001 * f, -0.002])
        r.offsets = [0.001 * f, -0.002]
        got = d.detect(arr, f + 1, t, 0.0)
        want = O.detect_frame(r, to_oracle_matches(arr), f + 1, t, 0.0)
        assert_frames_equal(got, want)
        assert got["t_capture_camera"] is None
        assert_colors_equal(d, r)
        d.set_t_sent(t + 0.004)
        want["t_sent"] = t + 0.004
        d.track()
        O.detection_tracking(r, want)
        assert_tracked_equal(d, r, 2)
        n_tracked_bots += sum(1 for o in r.tracked[2] if o["id"] >= 0 and o["age"] > 1)
    assert n_tracked_bots >= 6  # the tracker re-identified robots across frames, so the tracked-hypothesis path ran


def test_tracked_objects_from_another_camera_and_stale_clock():
    d, r = make_pair(TILTED)
    arr, truth, _ = scene(41, n_robots=4, n_balls=1, n_clutter=3, dropout=0.3)
    objs = np.zeros(len(truth) + 1, H.VPH_TRACKED)
    for i, t in enumerate(truth):
        objs[i] = (t["id"], 9.0 if i % 2 else 9.98, t["x"] + 5, t["y"] - 4, 143.0, t["w"] + 0.03, 40.0, -25.0, 0.0, 0.1, 0.9, 5)
    objs[-1] = (-1, 9.9, 100.0, 100.0, 21.5, 0.0, 0, 0, 0, 0, 0.8, 3)  # a ball: skipped by the bot generator
    d.set_tracked(1, objs)
    r.tracked[1] = [dict(id=int(o["id"]), timestamp=float(o["timestamp"]), x=F(o["x"]), y=F(o["y"]), z=F(o["z"]), w=F(o["w"]), vx=F(o["vx"]), vy=F(o["vy"]), vz=F(o["vz"]),
                         vw=F(o["vw"]), confidence=F(o["confidence"]), age=int(o["age"])) for o in objs]
    got = d.detect(arr, 3, 10.0, 9.5)
    want = O.detect_frame(r, to_oracle_matches(arr), 3, 10.0, 9.5)
    assert_frames_equal(got, want)
    assert got["t_capture_camera"] == 9.5


def test_empty_and_degenerate_inputs():
    d, r = make_pair(NADIR)
    before = d.colors().copy()
    fr = d.detect(np.zeros(0, H.VPH_MATCH), 1, 0.0)
    assert len(fr["balls"]) == len(fr["robots_yellow"]) == len(fr["robots_blue"]) == 0
    assert np.array_equal(d.colors(), before)  # updateColors runs on empty lists and changes nothing
    # three blobs: below the four-neighbour minimum, only ball hypotheses
    arr, _, _ = scene(51, n_robots=0, n_balls=3, n_clutter=0)
    assert_frames_equal(d.detect(arr, 2, 0.1), O.detect_frame(r, to_oracle_matches(arr), 2, 0.1, 0.0))
    # NaN sub-pixel positions (plateau peaks) are data: they enter the tree and match nothing
    arr, _, _ = scene(52, n_robots=3, n_balls=2, n_clutter=2)
    arr["x"][1] = np.nan
    arr["y"][1] = np.nan
    got = d.detect(arr, 3, 0.2)
    want = O.detect_frame(r, to_oracle_matches(arr), 3, 0.2, 0.0)
    assert_frames_equal(got, want)
    # without a geometry the stage refuses to run (main.cpp:277)
    assert H.Detector().l.vph_detect(H.Detector().h, None, 0, 1, C_double(0), C_double(0)) != 0


def test_clipping_filter_keeps_the_better_of_two_overlapping_robots():
    d, r = make_pair(NADIR)
    arr, truth, _ = scene(61, n_robots=1, n_balls=0, n_clutter=0, jitter=0.2)
    ghost = arr.copy()
    ghost["x"] += F(60.0)  # a second, noisier pattern 60 mm away: physically impossible next to the first
    ghost["y"] += (np.arange(len(ghost)) * 1.7).astype(np.float32)
    both = np.concatenate([arr, ghost])
    got = d.detect(both, 1, 0.0)
    want = O.detect_frame(r, to_oracle_matches(both), 1, 0.0, 0.0)
    assert_frames_equal(got, want)
    assert len(got["robots_yellow"]) + len(got["robots_blue"]) == 1


def C_double(v):
    import ctypes
    return ctypes.c_double(v)


def test_rendered_frames_through_the_hot_path_oracle_yield_the_scene(port):
    """Synthetic sensor frames (host/synth.cpp: robot i carries id i % 16, odd i are blue) -> CPU oracle of the GPU hot path ->
    this stage: every rendered robot comes back with its id and team, every ball is found, on each of three consecutive
    frames, and the C++ stage agrees with the checker while colours and tracks evolve."""
    import oracle as HO
    cfg = cases.cfgs.T_MID
    model = HO.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    mm = cfg.model()
    calib = (mm.f, mm.p[0], mm.p[1], mm.d, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, mm.c[2])
    d = H.Detector()
    d.set_geometry(calib, (cfg.qw, cfg.qh), (cfg.qw, cfg.qh), (9000, 6000, 300, -1, 0))
    r = O.Res(O.Camera(calib, (cfg.qw, cfg.qh), (cfg.qw, cfg.qh)))
    fs = F(cfg.field_scale)
    for k in range(3):
        m, _, _ = port.frame(cfg.synth(80 + k), cfg.layout, cfg.qw, cfg.qh, model, 180.0, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh, cfg.grad_offset,
                             cfg.circle_radius, cfg.list_radius, 15.0, 0.0, 2000)
        arr = np.zeros(len(m), H.VPH_MATCH)
        arr["x"], arr["y"] = m["x"] * fs + F(cfg.off_x), m["y"] * fs + F(cfg.off_y)  # Perspective::flat2field
        for f in ("color", "center", "circ", "score"):
            arr[f] = m[f]
        t = (k + 1) / 30.0
        got = d.detect(arr, k + 1, t, t)
        want = O.detect_frame(r, to_oracle_matches(arr), k + 1, t, t)
        assert_frames_equal(got, want)
        assert_colors_equal(d, r)
        d.track()
        O.detection_tracking(r, want)
        assert_tracked_equal(d, r, 0)
        assert sorted(int(b["robot_id"]) for b in got["robots_yellow"]) == [i for i in range(cfg.n_robots) if i % 2 == 0]
        assert sorted(int(b["robot_id"]) for b in got["robots_blue"]) == [i for i in range(cfg.n_robots) if i % 2 == 1]
        assert len(got["balls"]) == cfg.n_balls


def test_blob_centres_hit_the_rendered_discs(port):
    """Position accuracy of the hot path's blob list against the synthetic ground truth (the measure of the reference's
    blob_benchmark, src/blob_benchmark.cpp:44-84), here on the CPU oracle's lists: every rendered disc carries a blob inside it.
    The peaks sit half a flat pixel up-left of the true centres -- a property of the reference's quadrant bounds (taps at +1 / +r
    and -1 / -r of an inclusive table, satBlobCenter.cl:37-40: the positive boxes start two pixels out, the negative ones one),
    reproduced bit for bit by this build; around that offset the sub-pixel estimate scatters by a fraction of a pixel."""
    import importlib.util, os
    import oracle as HO
    spec = importlib.util.spec_from_file_location("blob_accuracy", os.path.join(cases.ROOT, "tools", "blob_accuracy.py"))
    acc = importlib.util.module_from_spec(spec)
    old_argv, sys_mod = list(__import__("sys").argv), __import__("sys")
    sys_mod.argv = ["blob_accuracy.py"]
    try:
        spec.loader.exec_module(acc)
    finally:
        sys_mod.argv = old_argv
    cfg = cases.cfgs.T_MID
    model = HO.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    seeds = [90, 91, 92, 93]
    lists = []
    for s in seeds:
        m, _, _ = port.frame(cfg.synth(s), cfg.layout, cfg.qw, cfg.qh, model, 180.0, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh, cfg.grad_offset, cfg.circle_radius,
                             cfg.list_radius, 15.0, 0.0, 2000)
        lists.append(m)
    st = acc.accuracy(cfg, lists, seeds)
    assert st["found"] == st["discs"] == len(seeds) * (5 * cfg.n_robots + cfg.n_balls)
    assert abs(st["mean_offset_px"][0] + 0.5) < 0.1 and abs(st["mean_offset_px"][1] + 0.5) < 0.1, st["mean_offset_px"]
    assert st["residual_rms_px"] < 0.35 and st["residual_max_px"] < 1.0, st
    for kind in ("centre", "side", "ball"):
        assert st[kind]["max_mm"] < 1.5 * cfg.field_scale, (kind, st[kind])


def test_batch_step_from_compact_slots_equals_the_batch_step_from_full_arrays(port):
    """vph_detect_batch_slots (compact record slots in, packet slots out: what vp_feeds reads back and gathers) produces, feed by feed and step by
    step, the very packets vph_detect_batch produces from the fixed-capacity arrays; a slot too small for its frame is refused, not truncated."""
    import oracle as HO
    import torch
    feeds_mod = cases.vp.feeds
    cfg = cases.cfgs.T_MID
    model = HO.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    n, mm, cap, pslot = 3, 400, 96, 2048
    counters = np.zeros((n, 3), np.int32)
    matches = np.zeros((n, 22 * mm), np.uint8)
    idx = np.zeros((n, mm), np.int32)
    for i in range(n):
        m, ix, c = port.frame(cfg.synth(120 + i), cfg.layout, cfg.qw, cfg.qh, model, 180.0, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh, cfg.grad_offset,
                              cfg.circle_radius, cfg.list_radius, 15.0, 0.0, mm)
        counters[i] = c
        matches[i, :22 * len(m)] = np.frombuffer(m.tobytes(), np.uint8)
        idx[i, :len(ix)] = ix.view(np.int32)
    assert 0 < int(counters[:, 0].max()) <= cap

    def detectors():
        out = []
        for i in range(n):
            dc = H.config_defaults()
            dc.cam_id = 7 + i
            d = H.Detector(dc)
            d.set_geometry_direct(H.nadir_calib(cfg.model()), (cfg.qw, cfg.qh), cfg.field_scale, cfg.off_x, cfg.off_y, cfg.fw, cfg.fh)
            out.append(d)
        return out
    da, db = detectors(), detectors()
    block = feeds_mod.pack_compact_host(torch.from_numpy(counters), torch.from_numpy(matches), torch.from_numpy(idx), cap).numpy()
    slot = feeds_mod.compact_slot_bytes(cap)
    for step in range(1, 4):  # colour model and tracker carried along in both sets of detectors
        t = step / 70.0
        pk = np.zeros((n, pslot - 4), np.uint8)
        ln = np.zeros(n, np.int32)
        H.detect_batch(da, matches, counters, mm, step, t, t, t + 0.002, 2, pk, ln)
        pb = np.zeros(n * pslot, np.uint8)
        assert H.detect_batch_slots(db, block, slot, mm, step, t, t, t + 0.002, 3, pb, pslot) == 0
        for i in range(n):
            length = int(pb[i * pslot:i * pslot + 4].view(np.int32)[0])
            assert length == int(ln[i]) > 0 and pb[i * pslot + 4:i * pslot + 4 + length].tobytes() == pk[i, :length].tobytes(), (step, i)
    # a slot that holds fewer records than the host loop would read (main.cpp:301) is an error, nothing runs
    small = feeds_mod.pack_compact_host(torch.from_numpy(counters), torch.from_numpy(matches), torch.from_numpy(idx), 8).numpy()
    assert H.detect_batch_slots(db, small, feeds_mod.compact_slot_bytes(8), mm, 9, 0.1, 0.1, 0.1, 1, np.zeros(n * pslot, np.uint8), pslot) != 0
    # a packet slot too small for the packet: length = -needed, payload untouched
    tiny = np.full(n * 64, 0xEE, np.uint8)
    assert H.detect_batch_slots(db, block, slot, mm, 10, 0.2, 0.2, 0.2, 1, tiny, 64) == 0
    assert all(int(tiny[i * 64:i * 64 + 4].view(np.int32)[0]) < -60 and int(tiny[i * 64 + 4]) == 0xEE for i in range(n))
