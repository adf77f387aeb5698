"""The C++ host facade (vision-processor_amd/host: opencl.h / Resources / frame loop mirroring main.cpp:251-317) run as its own
process on the GPU, its blob lists checked against the CPU oracle frame by frame."""
import os
import struct
import subprocess

import numpy as np
import pytest

import cases
import oracle as O

pytestmark = pytest.mark.gpu
cfgs = cases.cfgs
BIN = os.path.join(cases.ROOT, "vision-processor_amd", "host", "vp_frame_loop")


def run_loop(cfg, frames, seed, mode, tmp_path, debug_dump=None, debug_images=False):
    dump = str(tmp_path / f"{cfg.name}_{mode}.bin")
    args = [BIN, str(cfg.layout), str(cfg.qw), str(cfg.qh), str(cfg.cam_height), str(cfg.visible_len), str(cfg.distortion), str(cfg.field_len), str(cfg.field_wid),
            str(frames), str(0x5EED0000 + seed), str(cfg.n_robots), str(cfg.n_balls), mode]
    if mode == "direct":
        args += [repr(cfg.field_scale), repr(cfg.off_x), repr(cfg.off_y), str(cfg.fw), str(cfg.fh)]
    args.append(dump)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    if debug_dump:
        env["VP_DEBUG_DUMP"] = debug_dump
    if debug_images:
        env["VP_DEBUG_IMAGES"] = "1"
        os.makedirs(tmp_path / "img", exist_ok=True)  # CLImage::save writes "img/" + name + suffix relative to the working directory (opencl.cpp:171)
    out = subprocess.run(args, capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout + out.stderr
    recs = []
    with open(dump, "rb") as f:
        while True:
            hdr = f.read(32)
            if len(hdr) < 32:
                break
            frame_id, n, c0, c1, c2, fw, fh, wire_len = struct.unpack("8i", hdr)
            fs, ox, oy, _ = struct.unpack("4f", f.read(16))
            m = np.frombuffer(f.read(22 * n), O.MATCH_DTYPE)
            pos = np.frombuffer(f.read(8 * n), np.float32).reshape(n, 2)
            packet = f.read(wire_len)  # the SSL_WrapperPacket the reference would hand to socket->send (main.cpp:370)
            recs.append(dict(frame=frame_id, counter=np.array([c0, c1, c2], np.int32), fw=fw, fh=fh, fs=fs, ox=ox, oy=oy, matches=m, pos=pos, packet=packet))
    return recs, out.stdout


def check_against_oracle(port, cfg, recs, seed):
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    for k, r in enumerate(recs):
        fs = np.float32(r["fs"])
        o = int(np.ceil(np.float32(25.0) / fs)) // 3
        rad = int(np.ceil(np.float32(20.0) / fs))
        R = int(np.floor(np.float32(20.0) / fs))
        raw = cfg.synth(seed + k)
        m, idx, cnt = port.frame(raw, cfg.layout, cfg.qw, cfg.qh, model, 180.0, float(fs), r["ox"], r["oy"], 1, r["fw"], r["fh"], o, rad, R, 15.0, 0.0, 2000)
        assert np.array_equal(cnt, r["counter"]), (k, cnt, r["counter"])
        assert m.tobytes() == r["matches"].tobytes(), f"frame {k}: blob records differ from the oracle"
        want = np.stack([m["x"] * fs + np.float32(r["ox"]), m["y"] * fs + np.float32(r["oy"])], 1).astype(np.float32)  # Perspective.cpp:127-129
        assert np.array_equal(want, r["pos"], equal_nan=True)
        assert len(m) > 3


@pytest.mark.parametrize("cfg", [cfgs.T_MID, cfgs.CFG1], ids=lambda c: c.name)
def test_frame_loop_direct_scalars(port, cfg, tmp_path):
    recs, stdout = run_loop(cfg, 3, 50, "direct", tmp_path)
    assert len(recs) == 3 and "ms" in stdout  # printRuntimes output
    check_against_oracle(port, cfg, recs, 50)


def test_frame_loop_geometry_check(port, tmp_path):
    """Kernel scalars produced by the Eigen-free Perspective::geometryCheck; the oracle is fed the same scalars."""
    cfg = cfgs.T_MID
    recs, _ = run_loop(cfg, 2, 60, "calib", tmp_path)
    assert len(recs) == 2
    r = recs[0]
    # 3000 mm of the plane over 320 quad pixels, distortion shrinks the footprint a little: ~9-10 mm per quad pixel
    assert 8.0 < r["fs"] < 11.0 and r["fw"] % 2 == 0 and r["fh"] % 2 == 0
    check_against_oracle(port, cfg, recs, 60)


@pytest.mark.parametrize("cfg", [cfgs.T_MID, cfgs.CFG1], ids=lambda c: c.name)
def test_debug_and_stream_views(port, cfg, tmp_path):
    """Resources::quad2rgba / streamQuad / streamImage (Resources.cpp:145-186) through the facade: the RGBA view and the three NV12
    frames of the first frame (main.cpp:291,377-403) against the CPU oracle."""
    dbg = str(tmp_path / "debug.bin")
    recs, _ = run_loop(cfg, 1, 70, "direct", tmp_path, debug_dump=dbg)
    bufs = []
    with open(dbg, "rb") as f:
        while True:
            n = f.read(8)
            if len(n) < 8:
                break
            bufs.append(np.frombuffer(f.read(struct.unpack("Q", n)[0]), np.uint8))
    assert len(bufs) == 4
    raw = cfg.synth(70)
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    planes = port.raw2quad(raw, cfg.layout, cfg.qw, cfg.qh)
    flat = port.resample(planes, cfg.layout, model, 180.0, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh)
    bc = port.blobcenter(port.sat(port.graddot(flat, cfg.grad_offset)), cfg.circle_radius)
    assert np.array_equal(bufs[0].reshape(cfg.qh, cfg.qw, 4), port.quad2rgba(planes, cfg.layout, 1))
    assert np.array_equal(bufs[1], port.quad2nv12(planes, cfg.layout, 1))
    assert np.array_equal(bufs[2], port.rgba2nv12(flat))
    assert np.array_equal(bufs[3], port.f2nv12(bc))


@pytest.mark.parametrize("mode", ["direct", "calib"])
def test_frame_loop_emits_the_oracles_detection_packets(mode, tmp_path):
    """main.cpp:319-370 behind the facade: every frame's SSL_WrapperPacket, parsed by google.protobuf, equals what the CPU
    checker (oracle/detect_oracle.py: hypotheses, colour model and tracker carried from frame to frame) derives from the same
    blob list -- binary32 fields bit for bit.  (Parity of this CPU stage against the reference itself is unpinned.)"""
    import detect_oracle as D
    import test_detection_stage as S
    import test_wire_format as W
    cfg = cfgs.T_MID
    frames = 4
    recs, stdout = run_loop(cfg, frames, 80, mode, tmp_path)
    assert len(recs) == frames
    model = cfg.model()
    calib = (model.f, model.p[0], model.p[1], model.d, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, model.c[2])  # nadir: q = (x -1, y 0, z 0, w 0), t = R (-pos)
    cam = D.Camera(calib, (cfg.qw, cfg.qh), (cfg.qw, cfg.qh))
    field = dict(field_length=int(cfg.field_len), field_width=int(cfg.field_wid), boundary_width=300) if mode == "calib" else {}
    r = D.Res(cam, **field)
    total_bots = 0
    for k, rec in enumerate(recs):
        m = rec["matches"]
        matches = [dict(pos=(np.float32(px), np.float32(py)), color=tuple(int(v) for v in mm["color"]), center=tuple(int(v) for v in mm["center"]), circ=np.float32(mm["circ"]),
                        score=np.float32(mm["score"])) for mm, (px, py) in zip(m, rec["pos"])]
        t = (k + 1) / 30.0  # SyntheticCamera::getTime after readImage
        want = D.detect_frame(r, matches, k + 1, t, t)
        want["t_sent"] = t
        pkt = W.Wrapper()
        pkt.ParseFromString(rec["packet"])
        assert pkt.IsInitialized() and pkt.source == 3
        d = pkt.detection
        got = dict(frame_number=d.frame_number, camera_id=d.camera_id, t_capture=d.t_capture, t_capture_camera=d.t_capture_camera if d.HasField("t_capture_camera") else None,
                   t_offsets=np.asarray(d.t_offsets, np.float32),
                   balls=[dict(confidence=b.confidence, x=b.x, y=b.y, pixel_x=b.pixel_x, pixel_y=b.pixel_y) for b in d.balls],
                   robots_yellow=[dict(confidence=b.confidence, robot_id=b.robot_id, x=b.x, y=b.y, orientation=b.orientation, pixel_x=b.pixel_x, pixel_y=b.pixel_y, height=b.height)
                                  for b in d.robots_yellow],
                   robots_blue=[dict(confidence=b.confidence, robot_id=b.robot_id, x=b.x, y=b.y, orientation=b.orientation, pixel_x=b.pixel_x, pixel_y=b.pixel_y, height=b.height)
                                for b in d.robots_blue])
        S.assert_frames_equal(got, want)
        assert d.t_sent == t
        D.detection_tracking(r, want)
        total_bots += len(d.robots_yellow) + len(d.robots_blue)
        assert f"frame {k + 1}: {len(m)} blobs {len(d.balls)} balls {len(d.robots_yellow) + len(d.robots_blue)} bots" in stdout
    assert total_bots >= frames  # the rendered robots are found in every frame


def test_native_multi_feed_pipeline():
    """host/vp_feeds: several camera feeds through vp_frame + read-back + the CPU stage on host threads, C++ on the C ABI alone (no
    PyTorch in the process: the library runs on the system HIP runtime).  Every rendered robot and ball of every feed is reported."""
    import json
    exe = os.path.join(cases.ROOT, "vision-processor_amd", "host", "vp_feeds")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe, "6", "4", "1", "4", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["feeds"] == 6 and r["steps"] == 4 and r["detect"] == 1
    assert r["robots_per_frame"] == 16.0 and 8.0 <= r["balls_per_frame"] <= 9.0 and 100 < r["blobs_per_frame"] < 200
    assert 500 < r["packet_bytes_per_frame"] < 1500 and r["frames_per_s"] > 100


def test_debug_images_are_saved_as_the_reference_saves_them(port, tmp_path):
    """main.cpp:291-295 with debug.images on: flat->save(".flat.1.png"), gradDot->save(".gradDot.1.png", 0.25f, 128.f), blobCenter->save(".blob.1.png") --
    CLImage::save of the facade (opencl.cpp:161-179 without OpenCV: PNGs written by vph_write_png).  Decoded by an independent reader and held to the oracle's
    images: flat's three colour channels, gradDot and blobCenter through cv::saturate_cast<uint8_t>(factor * v + offset)."""
    PIL = pytest.importorskip("PIL.Image")
    cfg, seed = cfgs.T_MID, 70
    recs, _ = run_loop(cfg, 1, seed, "direct", tmp_path, debug_images=True)
    assert len(recs) == 1
    model = O.CameraModel.from_array(np.frombuffer(bytes(cfg.model()), np.uint8))
    raw = cfg.synth(seed)
    planes = port.raw2quad(raw, cfg.layout, cfg.qw, cfg.qh)
    flat = port.resample(planes, cfg.layout, model, cfg.max_robot_height, cfg.field_scale, cfg.off_x, cfg.off_y, 1, cfg.fw, cfg.fh)
    ch = port.chain_from_flat(flat, cfg.grad_offset, cfg.circle_radius, cfg.list_radius, cfg.circ_threshold, cfg.min_score, cfg.max_matches)

    def grey(v, factor, offset):  # cv::saturate_cast<uint8_t>(float): round half to even, saturate
        t = np.float32(factor) * v.astype(np.float32) + np.float32(offset)
        return np.clip(np.rint(t), 0, 255).astype(np.uint8)

    def load(suffix, mode):
        im = PIL.open(tmp_path / "img" / suffix)   # RawImage::name is empty for the synthetic camera: "img/" + "" + suffix
        im.load()
        assert im.mode == mode and im.size == (cfg.fw, cfg.fh)
        return np.asarray(im)

    assert np.array_equal(load(".flat.1.png", "RGB"), flat[..., :3])
    assert np.array_equal(load(".gradDot.1.png", "L"), grey(ch["graddot"], 0.25, 128.0))
    assert np.array_equal(load(".blob.1.png", "L"), grey(ch["blobcenter"], 1.0, 0.0))
    assert grey(ch["graddot"], 0.25, 128.0).std() > 1.0  # not a blank image
