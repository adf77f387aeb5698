"""SSL_WrapperPacket wire bytes: vp::encodeWrapperPacket / decodeWrapperPacket (host/blobs/wire.cpp)
against google.protobuf's own serializer.

The schema below is the data definition of proto/ssl_vision_detection.proto:3-35 and proto/ssl_vision_wrapper.proto:5-19
(field names, numbers, types, labels) entered as a FileDescriptorProto, because protoc is not in the image; google.protobuf
(python, upb backend) then builds real message classes from it.  This IS a third-party pin: the encoder is byte-compared
with protobuf's output and the decoder parses protobuf's output.
"""
import struct

import numpy as np
import pytest

import cases

H = cases.vp.hostlib
F = np.float32

pb = pytest.importorskip("google.protobuf")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

T = descriptor_pb2.FieldDescriptorProto
REQ, OPT, REP = T.LABEL_REQUIRED, T.LABEL_OPTIONAL, T.LABEL_REPEATED


def _schema():
    f = descriptor_pb2.FileDescriptorProto(name="vp_test_ssl_vision.proto", syntax="proto2")

    def msg(name, fields):
        m = f.message_type.add(name=name)
        for fname, num, label, typ, tname in fields:
            fd = m.field.add(name=fname, number=num, label=label, type=typ)
            if tname:
                fd.type_name = tname

    msg("SSL_DetectionBall", [("confidence", 1, REQ, T.TYPE_FLOAT, None), ("area", 2, OPT, T.TYPE_UINT32, None), ("x", 3, REQ, T.TYPE_FLOAT, None), ("y", 4, REQ, T.TYPE_FLOAT, None),
                               ("z", 5, OPT, T.TYPE_FLOAT, None), ("pixel_x", 6, REQ, T.TYPE_FLOAT, None), ("pixel_y", 7, REQ, T.TYPE_FLOAT, None)])
    msg("SSL_DetectionRobot", [("confidence", 1, REQ, T.TYPE_FLOAT, None), ("robot_id", 2, OPT, T.TYPE_UINT32, None), ("x", 3, REQ, T.TYPE_FLOAT, None), ("y", 4, REQ, T.TYPE_FLOAT, None),
                                ("orientation", 5, OPT, T.TYPE_FLOAT, None), ("pixel_x", 6, REQ, T.TYPE_FLOAT, None), ("pixel_y", 7, REQ, T.TYPE_FLOAT, None),
                                ("height", 8, OPT, T.TYPE_FLOAT, None)])
    msg("SSL_DetectionFrame", [("frame_number", 1, REQ, T.TYPE_UINT32, None), ("t_capture", 2, REQ, T.TYPE_DOUBLE, None), ("t_sent", 3, REQ, T.TYPE_DOUBLE, None),
                                ("t_capture_camera", 8, OPT, T.TYPE_DOUBLE, None), ("t_offsets", 9, REP, T.TYPE_FLOAT, None), ("camera_id", 4, REQ, T.TYPE_UINT32, None),
                                ("balls", 5, REP, T.TYPE_MESSAGE, ".SSL_DetectionBall"), ("robots_yellow", 6, REP, T.TYPE_MESSAGE, ".SSL_DetectionRobot"),
                                ("robots_blue", 7, REP, T.TYPE_MESSAGE, ".SSL_DetectionRobot")])
    e = f.enum_type.add(name="SSL_Source")
    for i, n in enumerate(("SSL_SOURCE_UNKNOWN", "SSL_SOURCE_OTHER", "SSL_SOURCE_SSL_VISION", "SSL_SOURCE_VISION_PROCESSOR", "SSL_SOURCE_GRSIM", "SSL_SOURCE_ERFORCE_SIM")):
        e.value.add(name=n, number=i)
    # geometry (field 2) is an opaque sub-message for this path: declared as bytes-compatible message with no known fields
    msg("SSL_GeometryData", [])
    msg("SSL_WrapperPacket", [("detection", 1, OPT, T.TYPE_MESSAGE, ".SSL_DetectionFrame"), ("geometry", 2, OPT, T.TYPE_MESSAGE, ".SSL_GeometryData"),
                               ("source", 3, OPT, T.TYPE_ENUM, ".SSL_Source")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("SSL_WrapperPacket"))


Wrapper = _schema()


def _detector_with_frame(frame_number, camera_id, t_capture, t_sent, t_cam, balls, yellow, blue, offsets):
    """Puts an arbitrary frame into a detector by decoding protobuf's bytes, returns (detector, protobuf bytes)."""
    w = Wrapper()
    w.source = 3
    d = w.detection
    d.frame_number, d.camera_id, d.t_capture, d.t_sent = frame_number, camera_id, t_capture, t_sent
    if t_cam is not None:
        d.t_capture_camera = t_cam
    d.t_offsets.extend(float(F(o)) for o in offsets)
    for b in balls:
        m = d.balls.add()
        m.confidence, m.x, m.y, m.pixel_x, m.pixel_y = (float(F(v)) for v in b)
    for lst, src in ((d.robots_yellow, yellow), (d.robots_blue, blue)):
        for r in src:
            m = lst.add()
            m.confidence, m.robot_id, m.x, m.y, m.orientation, m.pixel_x, m.pixel_y, m.height = float(F(r[0])), int(r[1]), *(float(F(v)) for v in r[2:])
    data = w.SerializeToString(deterministic=True)
    det = H.Detector()
    has, src = det.decode(data)
    assert has and src == 3
    return det, data


CASES = {
    "empty": dict(frame_number=1, camera_id=0, t_capture=0.0, t_sent=0.0, t_cam=None, balls=[], yellow=[], blue=[], offsets=[]),
    "typical": dict(frame_number=123456, camera_id=3, t_capture=1712345678.123456, t_sent=1712345678.130001, t_cam=98765.4321,
                    balls=[(0.91, -1234.5, 678.9, 512.25, 300.75), (0.3, 0.0, -0.0, 1e-3, 1023.0)],
                    yellow=[(0.99, 0, 100.0, -200.0, 3.14159, 10.0, 20.0, 143.0), (0.5, 15, -4500.0, 3000.0, -3.14159, 1000.5, 700.25, 150.0)],
                    blue=[(0.75, 7, 1.5, 2.5, 0.0, 3.5, 4.5, 148.725)], offsets=[0.001, -0.25, 3.0]),
    # varint boundaries: 127/128, 16383/16384, 2^32-1; long sub-messages; many repeats (length prefixes > 127 bytes)
    "varints": dict(frame_number=0xFFFFFFFF, camera_id=128, t_capture=-1.5, t_sent=float("inf"), t_cam=0.0, balls=[(float("nan"), 1, 2, 3, 4)] * 40,
                    yellow=[(0.1, 127, 0, 0, 0, 0, 0, 0), (0.1, 128, 0, 0, 0, 0, 0, 0), (0.1, 16384, 0, 0, 0, 0, 0, 0)], blue=[(1.0, 0xFFFFFFFF, 1, 1, 1, 1, 1, 1)] * 16,
                    offsets=[float(i) for i in range(50)]),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_encoder_bytes_equal_protobufs(name):
    det, data = _detector_with_frame(**CASES[name])
    mine = det.encode()
    assert mine == data  # byte-identical to google.protobuf's deterministic serialisation
    back = Wrapper()
    back.ParseFromString(mine)
    assert back.IsInitialized() and back.source == 3


@pytest.mark.parametrize("name", sorted(CASES))
def test_decoder_reads_protobufs_bytes(name):
    c = CASES[name]
    det, _ = _detector_with_frame(**c)
    fr = det.frame()
    assert fr["frame_number"] == c["frame_number"] and fr["camera_id"] == c["camera_id"]
    assert fr["t_capture"] == c["t_capture"] and fr["t_sent"] == c["t_sent"] and fr["t_capture_camera"] == c["t_cam"]
    assert np.array_equal(fr["t_offsets"], np.asarray(c["offsets"], np.float32))
    assert len(fr["balls"]) == len(c["balls"]) and len(fr["robots_yellow"]) == len(c["yellow"]) and len(fr["robots_blue"]) == len(c["blue"])
    for g, w in zip(fr["robots_yellow"].tolist() + fr["robots_blue"].tolist(), c["yellow"] + c["blue"]):
        assert g[1] == w[1] and np.array_equal(np.asarray(g[2:], np.float32), np.asarray(w[2:], np.float32))
    for g, w in zip(fr["balls"].tolist(), c["balls"]):
        assert np.array_equal(np.asarray(g, np.float32), np.asarray(w, np.float32), equal_nan=True)


def test_a_detected_frame_round_trips_through_protobuf():
    import test_detection_stage as S
    d, _ = S.make_pair(S.NADIR, cam_id=5)
    arr, truth, _ = S.scene(71)
    d.set_offsets([0.5, -0.5])
    fr = d.detect(arr, 42, 100.25, 99.0)
    d.set_t_sent(100.26)
    w = Wrapper()
    w.ParseFromString(d.encode())
    assert w.IsInitialized()
    assert w.source == 3 and w.detection.frame_number == 42 and w.detection.camera_id == 5 and w.detection.t_capture == 100.25 and w.detection.t_sent == 100.26
    assert w.detection.t_capture_camera == 99.0 and list(w.detection.t_offsets) == [0.5, -0.5]
    assert len(w.detection.robots_yellow) == len(fr["robots_yellow"]) and len(w.detection.robots_blue) == len(fr["robots_blue"]) and len(w.detection.balls) == len(fr["balls"])
    for m, g in zip(w.detection.robots_yellow, fr["robots_yellow"]):
        assert m.robot_id == g["robot_id"] and F(m.x) == g["x"] and F(m.orientation) == g["orientation"] and F(m.height) == g["height"] and m.HasField("height")
    for m, g in zip(w.detection.balls, fr["balls"]):
        assert F(m.confidence) == g["confidence"] and F(m.pixel_x) == g["pixel_x"] and not m.HasField("z") and not m.HasField("area")
    assert w.SerializeToString(deterministic=True) == d.encode()


def test_decoder_skips_unknown_fields_accepts_packed_and_rejects_garbage():
    det = H.Detector()
    frame = b"\x08\x01" + b"\x11" + struct.pack("<d", 1.0) + b"\x19" + struct.pack("<d", 2.0) + b"\x20\x09"
    frame += b"\x4a\x08" + struct.pack("<ff", 1.5, 2.5)                   # t_offsets packed (LEN) -- accepted like protobuf parsers do
    frame += b"\xa0\x06\x2a" + b"\xaa\x06\x03abc"                          # unknown fields 100 (varint) and 101 (bytes)
    geometry = b"\x0a\x02\x08\x01"
    pkt = b"\x12" + bytes([len(geometry)]) + geometry + b"\x0a" + bytes([len(frame)]) + frame + b"\x18\x02"
    has, src = det.decode(pkt)
    assert has and src == 2
    fr = det.frame()
    assert fr["frame_number"] == 1 and fr["camera_id"] == 9 and fr["t_offsets"].tolist() == [1.5, 2.5]
    ref = Wrapper()
    ref.ParseFromString(pkt)
    assert list(ref.detection.t_offsets) == [1.5, 2.5] and ref.detection.camera_id == 9
    # geometry-only packet: no detection, previous frame stays
    has, src = det.decode(b"\x12\x00")
    assert not has and det.frame()["frame_number"] == 1
    for bad in (b"\x0a\x05\x08", b"\x0a\x02\x11\x00", b"\x80", b"\x0b\x00", b"\x0a\x04\x2a\x05\x0d\x00"):
        with pytest.raises(ValueError):
            det.decode(bad)
    assert det.decode(b"") == (False, 0)
