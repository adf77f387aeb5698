"""VisionSocket (src/udpsocket.cpp:26-113,146-163,204-256) on the loopback interface: the packet of wire.cpp travels through a
real datagram socket, comes back to its sender (as the multicast group delivers it) and feeds the tracker; a plain Python
socket -- what python/visionsocket.py opens -- receives the same bytes."""
import socket
import struct

import numpy as np
import pytest

import cases
import test_detection_stage as S

H = cases.vp.hostlib


def _free_udp_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tracked_equal(a, b):
    return len(a) == len(b) and all(x.tobytes() == y.tobytes() for x, y in zip(a, b))


def test_unicast_loopback_feeds_the_tracker_like_the_direct_path():
    port = _free_udp_port()
    a, _ = S.make_pair(S.NADIR, cam_id=4)  # packets over UDP
    b, _ = S.make_pair(S.NADIR, cam_id=4)  # the same frames, tracker fed directly (vph_track)
    a.udp_open("127.0.0.1", port)
    arr0, _, _ = S.scene(91, n_robots=5, n_balls=2, n_clutter=4)
    try:
        for f in range(3):
            arr = arr0.copy()
            arr["x"] += np.float32(3.0 * f)
            t = 2.0 + f / 60.0
            fa, fb = a.detect(arr, f + 1, t, t), b.detect(arr, f + 1, t, t)
            S.assert_frames_equal(fa, {k: (list(v) if isinstance(v, np.ndarray) and v.dtype.names else v) for k, v in fb.items()})
            for d in (a, b):
                d.set_t_sent(t + 0.003)
            assert a.encode() == b.encode()
            a.udp_send()
            assert a.udp_sync(f + 1, 3000) == f + 1  # own packet received, parsed and tracked by the receiver thread
            b.track()
            assert _tracked_equal(a.get_tracked(4), b.get_tracked(4)) and len(a.get_tracked(4)) >= 6
    finally:
        a.udp_close()
    assert a.udp_sync(1, 10) < 0  # closed socket: invalid


def test_a_plain_python_socket_receives_the_protobuf_packet():
    """Consumer side as in python/visionsocket.py: a UDP socket bound to the port, bytes parsed with google.protobuf."""
    W = pytest.importorskip("test_wire_format")
    port = _free_udp_port()
    rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    rx.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    d, _ = S.make_pair(S.NADIR, cam_id=1)
    group = "224.5.23.2"  # network.vision_ip default (Resources.cpp:113)
    multicast = True
    try:
        rx.bind(("", port))
        rx.setsockopt(socket.IPPROTO_IP, socket.IP_ADD_MEMBERSHIP, struct.pack("4s4s", socket.inet_aton(group), socket.inet_aton("0.0.0.0")))
    except OSError:
        multicast = False
    if multicast:
        try:
            d.udp_open(group, port)
        except OSError:
            multicast = False
    if not multicast:
        # no multicast route in this sandbox: the consumer takes the port, the producer sends unicast to it (its own bind fails
        # with a warning, exactly like a second process on the reference's port without the group)
        rx.close()
        rx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        rx.bind(("127.0.0.1", port))
        d.udp_open("127.0.0.1", port)
    rx.settimeout(3.0)
    arr, _, _ = S.scene(92)
    fr = d.detect(arr, 9, 5.0, 4.5)
    d.set_t_sent(5.01)
    d.udp_send()
    try:
        data = None
        for _ in range(4):
            try:
                data, _addr = rx.recvfrom(65535)
            except socket.timeout:
                break
            if data:
                break
        if not data and multicast:
            pytest.skip("multicast group joined but nothing delivered on this host")
        assert data == d.encode()
        w = W.Wrapper()
        w.ParseFromString(data)
        assert w.source == 3 and w.detection.frame_number == 9 and w.detection.camera_id == 1
        assert len(w.detection.robots_yellow) == len(fr["robots_yellow"]) and len(w.detection.robots_blue) == len(fr["robots_blue"])
    finally:
        d.udp_close()
        rx.close()


def test_foreign_and_malformed_datagrams_are_ignored():
    port = _free_udp_port()
    d, _ = S.make_pair(S.NADIR, cam_id=0)
    d.udp_open("127.0.0.1", port)
    tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    try:
        tx.sendto(b"\x0a\x05\x08", ("127.0.0.1", port))        # truncated
        tx.sendto(b"\x12\x02\x08\x01", ("127.0.0.1", port))    # geometry only
        tx.sendto(b"garbage-not-protobuf\xff\xff", ("127.0.0.1", port))
        other = H.Detector()
        arr, _, _ = S.scene(93)
        o, _ = S.make_pair(S.NADIR, cam_id=7)
        o.detect(arr, 1, 1.0)
        o.set_t_sent(1.001)
        tx.sendto(o.encode(), ("127.0.0.1", port))             # another camera's detections
        assert d.udp_sync(1, 3000) == 1                        # only the well-formed detection packet counted
        assert len(d.get_tracked(7)) >= 6 and len(d.get_tracked(0)) == 0
        del other
    finally:
        tx.close()
        d.udp_close()


def test_geometry_arrives_over_the_socket():
    """A geometry publisher (ssl-vision, or python/geom_publisher.py of the reference) multicasts SSL_WrapperPacket{geometry}; the
    receiver thread keeps the latest one and bumps a version when it differs (udpsocket.cpp:113-130,156-162); the frame loop then
    derives the kernel scalars (Perspective::geometryCheck)."""
    G = pytest.importorskip("test_geometry_packet")
    port = _free_udp_port()
    d = H.Detector()
    d.udp_open("127.0.0.1", port)
    tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    try:
        assert d.udp_apply_geometry((64, 48), timeout_ms=50) is None  # nothing heard yet
        pkt = G._packet(G.SMALL)
        tx.sendto(pkt, ("127.0.0.1", port))
        got = d.udp_apply_geometry((64, 48))
        want = H.geometry_check(G.SMALL, 64, 48, 64, 48, (9000, 6000, 300, 250), resampling_factor=1.0)
        assert got is not None and got["geometry_version"] == 1
        assert G.bits(got["field_scale"]) == G.bits(want["field_scale"]) and np.array_equal(G.bits(got["extent"]), G.bits(want["extent"])) and bytes(got["model"]) == bytes(want["model"])
        tx.sendto(pkt, ("127.0.0.1", port))  # the publisher repeats itself every few seconds: same content, no new version
        assert d.udp_apply_geometry((64, 48), timeout_ms=200) is None
        tx.sendto(G._packet(G.SMALL, goal_boundary=None), ("127.0.0.1", port))  # the field changed
        got2 = d.udp_apply_geometry((64, 48))
        want2 = H.geometry_check(G.SMALL, 64, 48, 64, 48, (9000, 6000, 300, -1), resampling_factor=1.0)
        assert got2 is not None and got2["geometry_version"] == 2 and np.array_equal(G.bits(got2["extent"]), G.bits(want2["extent"]))
    finally:
        tx.close()
        d.udp_close()


def test_clock_offsets_between_cameras():
    """timeSynchronization / updateTime (udpsocket.cpp:259-302): camera 0 hears camera 2, whose clock runs 50 ms ahead and which
    reports back the offset it measured against camera 0; the received offset lands in the next frame's t_offsets and the local clock
    is steered by half the asymmetry; a large backward step is refused."""
    import time
    port = _free_udp_port()
    d, _ = S.make_pair(S.NADIR, cam_id=0)
    other, _ = S.make_pair(S.NADIR, cam_id=2)
    d.udp_open("127.0.0.1", port)
    tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    arr, _, _ = S.scene(95)
    try:
        def hear(t_sent_skew, reported):
            other.set_offsets(reported)
            other.detect(arr, 1, 1.0)
            other.set_t_sent(time.time() + t_sent_skew)
            n = d.udp_sync(0, 10)
            tx.sendto(other.encode(), ("127.0.0.1", port))
            assert d.udp_sync(max(n, 0) + 1, 3000) == max(n, 0) + 1

        t0 = d.udp_update_time()
        assert abs(t0 - time.time()) < 0.05  # nothing heard yet: plain system time
        hear(0.050, [0.010])                  # other.t_sent - local time = +50 ms; it saw us 10 ms ahead of itself
        t1 = d.udp_update_time()
        fr = d.detect(arr, 2, 2.0)
        offs = fr["t_offsets"]
        assert len(offs) == 3 and offs[0] == 0 and offs[1] == 0 and abs(offs[2] - 0.050) < 0.02  # indexed by sender id
        # offset = (received - sent) / (2 * cams) = (0.050 - 0.010) / 6, applied once
        assert abs((t1 - time.time()) - (0.040 / 6)) < 0.01
        hear(-0.500, [0.0])                   # a 500 ms backward step: refused (offset < -10 ms)
        t2 = d.udp_update_time()
        assert abs((t2 - time.time()) - (0.040 / 6)) < 0.01
    finally:
        tx.close()
        d.udp_close()


def test_game_controller_referee_sets_the_robot_heights():
    """GCSocket (udpsocket.cpp:304-330): the height table gives maxBotHeight (its maximum) and the default (its mean); a Referee packet
    naming listed teams switches the per-colour heights, which the detection frame then reports (hypothesis.cpp:141-149)."""
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    T = descriptor_pb2.FieldDescriptorProto
    f = descriptor_pb2.FileDescriptorProto(name="vp_test_referee.proto", syntax="proto2")
    ti = f.message_type.add(name="TeamInfo")  # proto/ssl_gc_referee_message.proto:119-150 (the fields needed to build a valid packet)
    for name, num, typ in (("name", 1, T.TYPE_STRING), ("score", 2, T.TYPE_UINT32), ("red_cards", 3, T.TYPE_UINT32), ("yellow_cards", 5, T.TYPE_UINT32),
                           ("timeouts", 6, T.TYPE_UINT32), ("timeout_time", 7, T.TYPE_UINT32), ("goalkeeper", 8, T.TYPE_UINT32)):
        ti.field.add(name=name, number=num, label=T.LABEL_REQUIRED, type=typ)
    ref = f.message_type.add(name="Referee")  # :152-157
    for name, num, typ, tn in (("packet_timestamp", 1, T.TYPE_UINT64, None), ("stage", 2, T.TYPE_INT32, None), ("command", 4, T.TYPE_INT32, None),
                               ("command_counter", 5, T.TYPE_UINT32, None), ("command_timestamp", 6, T.TYPE_UINT64, None), ("yellow", 7, T.TYPE_MESSAGE, ".TeamInfo"),
                               ("blue", 8, T.TYPE_MESSAGE, ".TeamInfo")):
        fd = ref.field.add(name=name, number=num, label=T.LABEL_REQUIRED, type=typ)
        if tn:
            fd.type_name = tn
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    Referee = message_factory.GetMessageClass(pool.FindMessageTypeByName("Referee"))

    def referee(yellow, blue):
        m = Referee()
        m.packet_timestamp, m.stage, m.command, m.command_counter, m.command_timestamp = 1, 0, 0, 7, 2
        for team, name in ((m.yellow, yellow), (m.blue, blue)):
            team.name, team.score, team.red_cards, team.yellow_cards, team.timeouts, team.timeout_time, team.goalkeeper = name, 0, 0, 0, 4, 300000000, 0
        return m.SerializeToString()

    heights = {"TIGERs Mannheim": 143.0, "ER-Force": 148.0, "Robocup Junior SSL Team": 180.0, "ibis": 90.0}
    port = _free_udp_port()
    d, _ = S.make_pair(S.NADIR)
    d.gc_open("127.0.0.1", port, heights)
    tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    try:
        n, h = d.gc_sync(0, 10)
        assert h == {"max": 180.0, "default": (143.0 + 148.0 + 180.0 + 90.0) / 4, "yellow": 140.25, "blue": 140.25}
        tx.sendto(referee("TIGERs Mannheim", "ibis"), ("127.0.0.1", port))
        n, h = d.gc_sync(1, 3000)
        assert n == 1 and h["yellow"] == 143.0 and h["blue"] == 90.0 and h["max"] == 180.0
        tx.sendto(referee("Unknown Team", "ER-Force"), ("127.0.0.1", port))  # an unlisted name leaves that colour alone
        n, h = d.gc_sync(2, 3000)
        assert n == 2 and h["yellow"] == 143.0 and h["blue"] == 148.0
        tx.sendto(b"\\x3a\\x05\\x0a\\x09", ("127.0.0.1", port))  # truncated TeamInfo: dropped
        assert d.gc_sync(3, 200)[0] == -1
        arr, _, _ = S.scene(96, n_robots=4, n_balls=1, n_clutter=0)
        fr = d.detect(arr, 1, 1.0)
        assert len(fr["robots_yellow"]) == 2 and len(fr["robots_blue"]) == 2
        assert all(b["height"] == np.float32(143.0) for b in fr["robots_yellow"]) and all(b["height"] == np.float32(148.0) for b in fr["robots_blue"])
    finally:
        tx.close()
        d.gc_close()
