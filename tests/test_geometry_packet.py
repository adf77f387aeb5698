"""Geometry over the wire: an SSL_WrapperPacket{geometry} built by google.protobuf from the reference's schema
(proto/ssl_vision_geometry.proto, entered as a descriptor: protoc is absent) is parsed by host/blobs/wire.cpp and drives
Perspective::geometryCheck exactly like the same values handed over as plain arguments."""
import math

import numpy as np
import pytest

import cases
import test_detection_stage as S

H = cases.vp.hostlib
pytest.importorskip("google.protobuf")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

T = descriptor_pb2.FieldDescriptorProto
REQ, OPT, REP = T.LABEL_REQUIRED, T.LABEL_OPTIONAL, T.LABEL_REPEATED


def _schema():
    f = descriptor_pb2.FileDescriptorProto(name="vp_test_ssl_geometry.proto", syntax="proto2")

    def msg(name, fields):
        m = f.message_type.add(name=name)
        for fname, num, label, typ, tname in fields:
            fd = m.field.add(name=fname, number=num, label=label, type=typ)
            if tname:
                fd.type_name = tname

    e = f.enum_type.add(name="SSL_FieldShapeType")
    for i, n in enumerate(("Undefined", "CenterCircle", "TopTouchLine", "BottomTouchLine", "LeftGoalLine", "RightGoalLine", "HalfwayLine", "CenterLine")):
        e.value.add(name=n, number=i)
    msg("Vector2f", [("x", 1, REQ, T.TYPE_FLOAT, None), ("y", 2, REQ, T.TYPE_FLOAT, None)])
    msg("SSL_FieldLineSegment", [("name", 1, REQ, T.TYPE_STRING, None), ("p1", 2, REQ, T.TYPE_MESSAGE, ".Vector2f"), ("p2", 3, REQ, T.TYPE_MESSAGE, ".Vector2f"),
                                  ("thickness", 4, REQ, T.TYPE_FLOAT, None), ("type", 5, OPT, T.TYPE_ENUM, ".SSL_FieldShapeType")])
    msg("SSL_FieldCircularArc", [("name", 1, REQ, T.TYPE_STRING, None), ("center", 2, REQ, T.TYPE_MESSAGE, ".Vector2f"), ("radius", 3, REQ, T.TYPE_FLOAT, None),
                                  ("a1", 4, REQ, T.TYPE_FLOAT, None), ("a2", 5, REQ, T.TYPE_FLOAT, None), ("thickness", 6, REQ, T.TYPE_FLOAT, None),
                                  ("type", 7, OPT, T.TYPE_ENUM, ".SSL_FieldShapeType")])
    i32 = T.TYPE_INT32
    msg("SSL_GeometryFieldSize", [("field_length", 1, REQ, i32, None), ("field_width", 2, REQ, i32, None), ("goal_width", 3, REQ, i32, None), ("goal_depth", 4, REQ, i32, None),
                                   ("boundary_width", 5, REQ, i32, None), ("boundary_width_goal_line", 16, OPT, i32, None),
                                   ("field_lines", 6, REP, T.TYPE_MESSAGE, ".SSL_FieldLineSegment"), ("field_arcs", 7, REP, T.TYPE_MESSAGE, ".SSL_FieldCircularArc"),
                                   ("penalty_area_depth", 8, OPT, i32, None), ("penalty_area_width", 9, OPT, i32, None), ("center_circle_radius", 10, OPT, i32, None),
                                   ("line_thickness", 11, OPT, i32, None), ("goal_center_to_penalty_mark", 12, OPT, i32, None), ("goal_height", 13, OPT, i32, None),
                                   ("ball_radius", 14, OPT, T.TYPE_FLOAT, None), ("max_robot_radius", 15, OPT, T.TYPE_FLOAT, None),
                                   ("goal_substitution_area_width", 17, OPT, i32, None)])
    fl = T.TYPE_FLOAT
    msg("SSL_GeometryCameraCalibration", [("camera_id", 1, REQ, T.TYPE_UINT32, None), ("focal_length", 2, REQ, fl, None), ("principal_point_x", 3, REQ, fl, None),
                                           ("principal_point_y", 4, REQ, fl, None), ("distortion", 5, REQ, fl, None), ("q0", 6, REQ, fl, None), ("q1", 7, REQ, fl, None),
                                           ("q2", 8, REQ, fl, None), ("q3", 9, REQ, fl, None), ("tx", 10, REQ, fl, None), ("ty", 11, REQ, fl, None), ("tz", 12, REQ, fl, None),
                                           ("derived_camera_world_tx", 13, OPT, fl, None), ("derived_camera_world_ty", 14, OPT, fl, None),
                                           ("derived_camera_world_tz", 15, OPT, fl, None), ("pixel_image_width", 16, OPT, T.TYPE_UINT32, None),
                                           ("pixel_image_height", 17, OPT, T.TYPE_UINT32, None)])
    msg("SSL_BallModelStraightTwoPhase", [("acc_slide", 1, REQ, T.TYPE_DOUBLE, None), ("acc_roll", 2, REQ, T.TYPE_DOUBLE, None), ("k_switch", 3, REQ, T.TYPE_DOUBLE, None)])
    msg("SSL_GeometryModels", [("straight_two_phase", 1, OPT, T.TYPE_MESSAGE, ".SSL_BallModelStraightTwoPhase")])
    msg("SSL_GeometryData", [("field", 1, REQ, T.TYPE_MESSAGE, ".SSL_GeometryFieldSize"), ("calib", 2, REP, T.TYPE_MESSAGE, ".SSL_GeometryCameraCalibration"),
                              ("models", 3, OPT, T.TYPE_MESSAGE, ".SSL_GeometryModels")])
    msg("SSL_DetectionFrame", [("frame_number", 1, REQ, T.TYPE_UINT32, None), ("t_capture", 2, REQ, T.TYPE_DOUBLE, None), ("t_sent", 3, REQ, T.TYPE_DOUBLE, None),
                                ("camera_id", 4, REQ, T.TYPE_UINT32, None)])
    msg("SSL_WrapperPacket", [("detection", 1, OPT, T.TYPE_MESSAGE, ".SSL_DetectionFrame"), ("geometry", 2, OPT, T.TYPE_MESSAGE, ".SSL_GeometryData"),
                               ("source", 3, OPT, T.TYPE_INT32, None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("SSL_WrapperPacket"))


Wrapper = _schema()
OTHER = S.NADIR["calib"]
SMALL = (70.0, 30.0, 26.0) + tuple(S.TILTED["calib"][3:])  # the tilted camera with the intrinsics of a 64 x 48 sensor


def _packet(calib0, cam_ids=(0, 1), goal_boundary=250, with_detection=False, ball_radius=21.5, calib_size=(64, 48)):
    w = Wrapper()
    w.source = 2  # SSL_SOURCE_SSL_VISION
    if with_detection:
        w.detection.frame_number, w.detection.t_capture, w.detection.t_sent, w.detection.camera_id = 5, 1.0, 1.1, 9
    g = w.geometry
    f = g.field
    f.field_length, f.field_width, f.goal_width, f.goal_depth, f.boundary_width = 9000, 6000, 1000, 180, 300
    if goal_boundary is not None:
        f.boundary_width_goal_line = goal_boundary
    f.penalty_area_depth, f.penalty_area_width, f.center_circle_radius, f.line_thickness, f.goal_height = 1000, 2000, 500, 10, 160
    if ball_radius is not None:
        f.ball_radius = ball_radius
    f.max_robot_radius = 90.0
    for i, (x1, y1, x2, y2) in enumerate(S.LINES):
        ln = f.field_lines.add()
        ln.name, ln.thickness, ln.type = f"line{i}", 10.0, 2
        ln.p1.x, ln.p1.y, ln.p2.x, ln.p2.y = x1, y1, x2, y2
    for (cx, cy, rad, a1, a2) in S.ARCS:
        a = f.field_arcs.add()
        a.name, a.radius, a.a1, a.a2, a.thickness, a.type = "CenterCircle", rad, a1, a2, 10.0, 1
        a.center.x, a.center.y = cx, cy
    for cid in cam_ids:
        c = g.calib.add()
        vals = calib0 if cid == 0 else OTHER
        c.camera_id = cid
        (c.focal_length, c.principal_point_x, c.principal_point_y, c.distortion, c.q0, c.q1, c.q2, c.q3, c.tx, c.ty, c.tz) = [float(np.float32(v)) for v in vals]
        c.pixel_image_width, c.pixel_image_height = calib_size
        if cid == 1:
            c.derived_camera_world_tx, c.derived_camera_world_ty, c.derived_camera_world_tz = 0.0, 0.0, 5000.0
    g.models.straight_two_phase.acc_slide, g.models.straight_two_phase.acc_roll, g.models.straight_two_phase.k_switch = -3.0, -0.26, 0.64
    return w.SerializeToString()


def bits(v):
    return np.asarray(v, np.float32).view(np.uint32)


def test_geometry_packet_drives_the_kernel_scalars():
    d = H.Detector()
    got = d.apply_geometry_packet(_packet(SMALL), (64, 48))
    assert got is not None and got["geometry_version"] == 1
    want = H.geometry_check(SMALL, 64, 48, 64, 48, (9000, 6000, 300, 250), ball_radius=21.5, geometry_tolerance=10.0, max_bot_height=180.0, resampling_factor=1.0)
    assert bits(got["field_scale"]) == bits(want["field_scale"]) and np.array_equal(bits(got["extent"]), bits(want["extent"]))
    assert (got["fw"], got["fh"], got["grad_offset"], got["circle_radius"], got["list_radius"]) == (want["fw"], want["fh"], want["grad_offset"], want["circle_radius"], want["list_radius"])
    assert bytes(got["model"]) == bytes(want["model"])
    # a packet without a calibration for this camera changes nothing
    assert d.apply_geometry_packet(_packet(SMALL, cam_ids=(1,)), (64, 48)) is None
    # no geometry at all
    w = Wrapper()
    w.source = 2
    assert d.apply_geometry_packet(w.SerializeToString(), (64, 48)) is None
    # goal-line boundary absent -> boundary_width is used (CameraModel.cpp:20-22); a detection in the same packet does not disturb
    d2 = H.Detector()
    got2 = d2.apply_geometry_packet(_packet(SMALL, goal_boundary=None, with_detection=True), (64, 48))
    want2 = H.geometry_check(SMALL, 64, 48, 64, 48, (9000, 6000, 300, -1), resampling_factor=1.0)
    assert bits(got2["field_scale"]) == bits(want2["field_scale"]) and np.array_equal(bits(got2["extent"]), bits(want2["extent"]))
    # calibration recorded at twice the sensor size: CameraModel::ensureSize rescales it (CameraModel.cpp:125-136)
    big = (140.0, 60.0, 52.0) + tuple(SMALL[3:])
    d3 = H.Detector()
    got3 = d3.apply_geometry_packet(_packet(big, calib_size=(128, 96)), (64, 48), resampling_factor=0.5)
    want3 = H.geometry_check(big, 64, 48, 128, 96, (9000, 6000, 300, 250), resampling_factor=0.5)
    assert bits(got3["field_scale"]) == bits(want3["field_scale"]) and (got3["fw"], got3["fh"]) == (want3["fw"], want3["fh"])
    with pytest.raises(ValueError):
        d.apply_geometry_packet(b"\x12\x05\x0a\x03\x08", (64, 48))


def test_lines_and_arcs_from_the_packet_reach_the_colour_model():
    """The parsed field markings feed ballAtLine (colorupdate.cpp:32-55): a detector configured from the packet behaves exactly like
    one configured with the same markings as arguments."""
    a = H.Detector()
    assert a.apply_geometry_packet(_packet(S.NADIR["calib"], cam_ids=(0,), goal_boundary=None, calib_size=S.NADIR["size"]), S.NADIR["size"]) is not None
    b, _ = S.make_pair(S.NADIR)
    # balls sitting on the halfway line and the centre circle: the field-line colour is learned from them
    arr, _, _ = S.scene(81, n_robots=3, n_balls=0, n_clutter=0)
    extra = np.zeros(5, H.VPH_MATCH)
    for i, (x, y) in enumerate([(0.0, -800.0), (0.0, 300.0), (2.0, 1200.0), (500.0 * math.cos(0.7), 500.0 * math.sin(0.7)), (-353.0, 354.0)]):
        extra[i] = (x, y, (228, 231, 224), (230, 230, 226), 30.0, 20.0)
    arr = np.concatenate([arr, extra])
    fa, fb = a.detect(arr, 1, 0.5), b.detect(arr, 1, 0.5)
    S.assert_frames_equal(fa, {k: (list(v) if isinstance(v, np.ndarray) and v.dtype.names else v) for k, v in fb.items()})
    assert np.array_equal(a.colors(), b.colors())
    assert tuple(a.colors()[6]) != (128, 128, 128)  # fieldLineColor moved away from its reference: the markings were seen
