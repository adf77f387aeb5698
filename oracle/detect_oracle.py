"""TEST INFRASTRUCTURE ONLY -- CPU checker for the blob -> detection stage.  Never imported by the product.

An independent restatement, in numpy binary32 scalars + glibc libm (ctypes), of the reference's
  src/blobs/kdtree.cpp:18-37, src/blobs/kmeans.cpp:19-90, src/blobs/hypothesis.cpp:21-282, src/blobs/colorupdate.cpp:21-141,
  src/main.cpp:43-241,319-362, src/udpsocket.cpp:165-256, src/CameraModel.cpp:86-172, src/pattern.h.
Structure differs from the product on purpose (recursive dict tree, tuples, python lists) so that the two only agree if both
follow the cited lines.

PARITY UNPINNED: the reference's host C++ needs Eigen / protobuf / yaml-cpp, none of which exist in this image, and the
reference holds no test or fixture for this stage.  This file pins the product's C++ against a second reading of the same
source, nothing more.  Every arithmetic step is one IEEE binary32 operation in the order of the cited source line; libm calls
(sinf, cosf, remainderf, atan2f, sqrtf, fmaf, copysignf) go to the same glibc the C++ links.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

F = np.float32
_m = C.CDLL("libm.so.6")
for _n, _k in (("sinf", 1), ("cosf", 1), ("sqrtf", 1), ("fabsf", 1), ("remainderf", 2), ("atan2f", 2), ("copysignf", 2), ("fmaf", 3)):
    getattr(_m, _n).restype = C.c_float
    getattr(_m, _n).argtypes = [C.c_float] * _k


def sinf(x): return F(_m.sinf(float(x)))
def cosf(x): return F(_m.cosf(float(x)))
def sqrtf(x): return F(_m.sqrtf(float(x)))
def fabsf(x): return F(_m.fabsf(float(x)))
def remainderf(x, y): return F(_m.remainderf(float(x), float(y)))
def atan2f(y, x): return F(_m.atan2f(float(y), float(x)))
def copysignf(x, y): return F(_m.copysignf(float(x), float(y)))
def fmaf(a, b, c): return F(_m.fmaf(float(a), float(b), float(c)))


PI = F(math.pi)
PI_2 = F(math.pi / 2)
TWO_PI = F(2.0 * math.pi)

# pattern.h:18-61
PATTERNS = [0b0100, 0b1100, 0b1101, 0b0101, 0b0010, 0b1010, 0b1011, 0b0011, 0b1111, 0b0000, 0b0110, 0b1001, 0b1110, 0b1000, 0b0111, 0b0001]
PATTERN_LUT = [9, 15, 4, 7, 0, 3, 10, 14, 13, 11, 5, 6, 1, 2, 12, 8]
ANGLES_B2B = [F(v) for v in (
    0., -2.13940875, -0.56861242, 0.56861242, 2.13940875,
    1.00218391, 0., 0.21678574, 0.78539816, 1.57079633,
    2.57298023, -2.92480691, 0., 1.57079633, 2.35619449,
    -2.57298023, -2.35619449, -1.57079633, 0., 2.92480691,
    -1.00218391, -1.57079633, -0.78539816, -0.21678574, 0.)]
PATTERN_POS = [(F(0), F(0)), (F(35), F(54.772)), (F(-54.772), F(35)), (F(-54.772), F(-35)), (F(35), F(-54.772))]
MIN_ROBOT_RADIUS = F(85.0)
MIN_ROBOT_FRONT_DISTANCE = F(55.0)
MIN_ROBOT_OPENING_ANGLE = F(0.86708)


def sqd(a, b):
    """(a - b).squaredNorm() of integer colour triples."""
    return sum((int(p) - int(q)) ** 2 for p, q in zip(a, b))


# ---- camera model (CameraModel.cpp:86-172) ---------------------------------------------------------------------------
class Camera:
    def __init__(self, calib, calib_size, img_size):
        focal, ppx, ppy, dist, q0, q1, q2, q3, tx, ty, tz = [F(v) for v in calib]
        self.f, self.pp, self.k2 = focal, (ppx, ppy), dist
        q = (q3, q0, q1, q2)  # Eigen::Quaternionf(w, x, y, z) = (q3, q0, q1, q2), :89
        self.pos = self._rotate(self._inverse(q), (-tx, -ty, -tz))
        n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])  # normalize(), :120
        q = tuple(c / n for c in q)
        self.i2f = self._matrix(self._inverse(q))
        self.f2i = self._matrix(q)
        self.f2i_t = self._mul(self.f2i, (-self.pos[0], -self.pos[1], -self.pos[2]))
        self.size = tuple(calib_size)
        if self.size != tuple(img_size):  # ensureSize, :125-136
            factor = F(img_size[0]) / F(self.size[0])
            self.size = tuple(img_size)
            self.f = self.f * factor
            self.pp = (self.pp[0] * factor, self.pp[1] * factor)

    @staticmethod
    def _inverse(q):
        n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]
        return (q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2)

    @staticmethod
    def _cross(a, b):
        return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])

    @classmethod
    def _rotate(cls, q, v):  # Eigen Quaternion::_transformVector: uv = 2 (qv x v); v + w uv + qv x uv
        qv = q[1:]
        uv = cls._cross(qv, v)
        uv = tuple(c + c for c in uv)
        c = cls._cross(qv, uv)
        return tuple(v[i] + q[0] * uv[i] + c[i] for i in range(3))

    @staticmethod
    def _matrix(q):  # Eigen Quaternion::toRotationMatrix
        w, x, y, z = q
        tx, ty, tz = F(2) * x, F(2) * y, F(2) * z
        twx, twy, twz = tx * w, ty * w, tz * w
        txx, txy, txz = tx * x, ty * x, tz * x
        tyy, tyz, tzz = ty * y, tz * y, tz * z
        return ((F(1) - (tyy + tzz), txy - twz, txz + twy), (txy + twz, F(1) - (txx + tzz), tyz - twx), (txz - twy, tyz + twx, F(1) - (txx + tyy)))

    @staticmethod
    def _mul(m, v):
        return tuple(m[r][0] * v[0] + m[r][1] * v[1] + m[r][2] * v[2] for r in range(3))

    def field2image(self, p):  # :148-159 (10 iterations on the host)
        r = self._mul(self.f2i, tuple(F(c) for c in p))
        ray = (r[0] + self.f2i_t[0], r[1] + self.f2i_t[1], r[2] + self.f2i_t[2])
        ox, oy = ray[0] / ray[2], ray[1] / ray[2]
        nx, ny = ox, oy
        for _ in range(10):
            d = F(1) + self.k2 * (nx * nx + ny * ny)
            nx, ny = ox / d, oy / d
        return (self.f * nx + self.pp[0], self.f * ny + self.pp[1])

    def image2field(self, p, height):  # :137-146,161-172
        height = F(height)
        nx, ny = (F(p[0]) - self.pp[0]) / self.f, (F(p[1]) - self.pp[1]) / self.f
        k = F(1) + self.k2 * (nx * nx + ny * ny)
        ray = self._mul(self.i2f, (nx * k, ny * k, F(1)))
        if ray[2] >= 0:
            return (F(np.nan),) * 3
        t = (-self.pos[2] + height) / ray[2]
        return (ray[0] * t + self.pos[0], ray[1] * t + self.pos[1], height)


# ---- kd-tree (kdtree.cpp:18-37) --------------------------------------------------------------------------------------
def kd_insert(node, idx, pts):
    dim = node["dim"]
    side = "l" if pts[idx][dim] < pts[node["i"]][dim] else "r"
    if node[side] is None:
        node[side] = {"i": idx, "dim": (dim + 1) % 2, "l": None, "r": None}
    else:
        kd_insert(node[side], idx, pts)


def kd_build(pts):
    root = {"i": 0, "dim": 0, "l": None, "r": None}
    for i in range(1, len(pts)):
        kd_insert(root, i, pts)
    return root


def kd_range(node, pts, point, radius, out):
    px, py = pts[node["i"]]
    dx, dy = px - point[0], py - point[1]
    if sqrtf(dx * dx + dy * dy) <= radius:
        out.append(node["i"])
    dim = node["dim"]
    if node["l"] is not None and point[dim] <= pts[node["i"]][dim] + radius:
        kd_range(node["l"], pts, point, radius, out)
    if node["r"] is not None and point[dim] >= pts[node["i"]][dim] - radius:
        kd_range(node["r"], pts, point, radius, out)
    return out


# ---- k-means (kmeans.cpp:19-90) --------------------------------------------------------------------------------------
def kmeans(contrast, values, c1, c2):
    """Returns (ok, c1, c2)."""
    if len(values) < 2:
        return False, c1, c2
    ing, outg = F(np.inf), F(np.inf)
    for i, v in enumerate(values):
        outg = min(outg, F(sqd(v, contrast)))
        for w in values[i + 1:]:
            ing = min(ing, F(sqd(w, v)))
    if ing > outg:
        return False, c1, c2
    b1, b2 = c1, c2
    c1 = min(values, key=lambda v: sqd(v, b1))  # python's min keeps the first minimum, like std::min_element
    c2 = min(values, key=lambda v: sqd(v, b2))
    if tuple(c1) == tuple(c2):
        return False, b1, b2
    o1, o2 = c2, c1
    while tuple(o1) != tuple(c1) and tuple(o2) != tuple(c2):
        g1 = [v for v in values if sqd(v, c1) < sqd(v, c2)]
        g2 = [v for v in values if not sqd(v, c1) < sqd(v, c2)]
        if not g1 or not g2:
            return False, b1, b2
        o1, o2 = c1, c2
        c1 = tuple(_idiv(sum(v[k] for v in g1), len(g1)) for k in range(3))
        c2 = tuple(_idiv(sum(v[k] for v in g2), len(g2)) for k in range(3))
    if F(int(math.sqrt(float(sqd(c1, c2))))) < sqrtf(outg) / F(2):  # Eigen norm() of an int vector is an int
        return False, b1, b2
    return True, tuple(c1), tuple(c2)


def _idiv(a, b):
    """C integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


# ---- hypotheses (hypothesis.cpp) -------------------------------------------------------------------------------------
def atan2_fast(y, x):  # :21-58
    y, x = F(y), F(x)
    swap = fabsf(x) < fabsf(y)
    divisor = y if swap else x
    if divisor == 0:
        return x if swap else y
    t = (x if swap else y) / divisor
    t2 = t * t
    p = fmaf(t2, F(-0.01172120), F(0.05265332))
    for a in (F(-0.11643287), F(0.19354346), F(-0.33262347), F(0.99997726)):
        p = fmaf(t2, p, a)
    res = t * p
    if swap:
        res = copysignf(PI_2, t) - res
    if x < 0:
        res = copysignf(PI, y) + res
    return res


def rot(angle, v):
    s, c = sinf(angle), cosf(angle)
    return (c * v[0] + (-s) * v[1], s * v[0] + c * v[1])


class Res:
    """The slice of `Resources` the stage uses; defaults of Resources.cpp:84-85,189-209."""

    def __init__(self, cam: Camera, **kw):
        self.cam = cam
        self.cam_id = 0
        self.min_tracking_radius, self.max_bot_acceleration, self.min_score, self.min_cam_edge_distance = 20.0, 6500.0, 5.0, 170.0
        self.min_confidence, self.clipping_tolerance, self.geometry_tolerance = F(0.2), F(10.0), F(10.0)
        self.reference_force, self.history_force = F(0.1), F(0.7)
        self.refs = {"orange": (192, 128, 64), "field": (128, 128, 128), "yellow": (255, 128, 0), "blue": (0, 128, 255), "green": (0, 255, 128), "pink": (255, 0, 128)}
        self.max_bot_height, self.default_bot_height, self.yellow_bot_height, self.blue_bot_height = 180.0, 143.0, 143.0, 143.0
        self.field_length, self.field_width, self.boundary_width, self.boundary_goal_line, self.line_thickness = 9000, 6000, 300, -1, 0
        self.ball_radius, self.max_robot_radius = F(21.5), F(90.0)
        self.lines, self.arcs = [], []
        self.tracked = {}  # cam -> list of dicts
        self.offsets = []
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)
        self.colors = dict(self.refs)
        self.colors["fieldLine"] = self.refs["field"]


def bot_new(blobs):
    """blobs: 5 entries, each a match dict or None (BotHypothesis ctor, :97-105)."""
    b = {"blobs": list(blobs), "pos": (F(0), F(0)), "orientation": F(0), "score": F(1), "offsetScore": F(1), "botId": -1, "tracked": None}
    b["n"] = sum(m is not None for m in blobs)
    # calcPos :155-191
    o_sin, o_cos = F(0), F(0)
    for a in range(5):
        if blobs[a] is None:
            continue
        for c in range(a + 1, 5):
            if blobs[c] is None:
                continue
            dx, dy = blobs[c]["pos"][0] - blobs[a]["pos"][0], blobs[c]["pos"][1] - blobs[a]["pos"][1]
            delta = atan2_fast(dy, dx) - ANGLES_B2B[c * 5 + a]
            o_sin = o_sin + sinf(delta)
            o_cos = o_cos + cosf(delta)
    if b["n"] >= 1:
        if b["n"] > 1:
            b["orientation"] = atan2_fast(o_sin, o_cos)
        px, py = F(0), F(0)
        for i in range(5):
            if blobs[i] is None:
                continue
            rp = rot(b["orientation"], PATTERN_POS[i])
            px = px + (blobs[i]["pos"][0] - rp[0])
            py = py + (blobs[i]["pos"][1] - rp[1])
        b["pos"] = (px / F(b["n"]), py / F(b["n"]))
    # calcOffsetScore :193-205
    for i in range(5):
        if blobs[i] is None:
            continue
        rp = rot(b["orientation"], PATTERN_POS[i])
        ox = (blobs[i]["pos"][0] - (b["pos"][0] + rp[0])) / F(10)
        oy = (blobs[i]["pos"][1] - (b["pos"][1] + rp[1])) / F(10)
        b["offsetScore"] = min(b["offsetScore"], F(1) / (F(1) + (ox * ox + oy * oy)))
    b["score"] = b["offsetScore"]
    return b


def calc_bot_id(r: Res, b):  # :216-228
    cols = [m["color"] for m in b["blobs"]]
    _, green, pink = kmeans(cols[0], cols[1:], r.colors["green"], r.colors["pink"])
    bits = [1 if sqd(c, green) < sqd(c, pink) else 0 for c in cols[1:]]
    team = 16 if sqd(cols[0], r.colors["blue"]) < sqd(cols[0], r.colors["yellow"]) else 0
    b["botId"] = team + PATTERN_LUT[(bits[0] << 3) + (bits[1] << 2) + (bits[2] << 1) + bits[3]]


def calc_tracking_score(r: Res, b):  # :252-282
    b["score"] = b["offsetScore"]
    if b["n"] < 2:
        b["score"] = F(0)
        return
    for i, m in enumerate(b["blobs"]):
        if m is None:
            continue
        if i == 0:
            own, other = ("blue", "yellow") if b["botId"] >= 16 else ("yellow", "blue")
        else:
            own, other = ("green", "pink") if (PATTERNS[b["botId"] % 16] >> (4 - i)) & 1 else ("pink", "green")
        if sqd(m["color"], r.colors[other]) - sqd(m["color"], r.colors[own]) <= 0:
            b["score"] = F(0)
            return


def bot_tracked(r: Res, tracked, tpos, blobs):  # :241-250
    b = bot_new(blobs)
    b["tracked"] = tracked
    b["botId"] = tracked["id"]
    rot_off = remainderf(b["orientation"] - tpos[2], TWO_PI) / PI
    dx, dy = (b["pos"][0] - tpos[0]) / F(10), (b["pos"][1] - tpos[1]) / F(10)
    b["offsetScore"] = b["offsetScore"] * (F(1) / (F(1) + (dx * dx + dy * dy) + rot_off * rot_off))
    b["offsetScore"] = b["offsetScore"] * (F(b["n"]) / F(5))
    calc_tracking_score(r, b)
    return b


def ball_color_score(r: Res, ball):  # :82-94
    c = ball["blob"]["color"]
    false_orange, orange, line = sqd(c, r.colors["field"]), sqd(c, r.colors["orange"]), sqd(c, r.colors["fieldLine"])
    if false_orange <= orange or line <= orange:
        ball["score"] = F(0)
    else:
        ball["score"] = ball["score"] * (F(1) - F(orange) / F(false_orange))


def clip_bot_bot(r: Res, a, b):  # :108-122
    dx, dy = b["pos"][0] - a["pos"][0], b["pos"][1] - a["pos"][1]
    d2 = dx * dx + dy * dy
    if d2 >= (F(2) * MIN_ROBOT_RADIUS) * (F(2) * MIN_ROBOT_RADIUS):
        return False
    ang = atan2_fast(dy, dx)
    sa, oa = remainderf(ang - a["orientation"], TWO_PI), remainderf(ang - b["orientation"], TWO_PI)
    m = ((MIN_ROBOT_FRONT_DISTANCE / cosf(sa) if fabsf(sa) < MIN_ROBOT_OPENING_ANGLE else MIN_ROBOT_RADIUS)
         + (MIN_ROBOT_FRONT_DISTANCE / cosf(oa) if fabsf(oa) < MIN_ROBOT_OPENING_ANGLE else MIN_ROBOT_RADIUS)) - r.clipping_tolerance
    return bool(d2 < m * m)


def clip_bot_ball(r: Res, bot, ball):  # :124-139
    cr = F(0.48837) * r.ball_radius
    dx, dy = ball["pos"][0] - bot["pos"][0], ball["pos"][1] - bot["pos"][1]
    d2 = dx * dx + dy * dy
    m = MIN_ROBOT_RADIUS + cr
    if d2 >= m * m:
        return False
    ang = remainderf(atan2_fast(dy, dx) - bot["orientation"], TWO_PI)
    if fabsf(ang) >= MIN_ROBOT_OPENING_ANGLE:
        return True
    m = (MIN_ROBOT_FRONT_DISTANCE + cr) / cosf(ang) - r.clipping_tolerance
    return bool(d2 < m * m)


# ---- colour update (colorupdate.cpp) ---------------------------------------------------------------------------------
def _mix(r: Res, ref, old, cur):  # :73-76
    uf = F(1) - r.reference_force - r.history_force
    return tuple(int(r.reference_force * F(a) + r.history_force * F(b) + uf * F(c)) for a, b, c in zip(ref, old, cur))


def ball_at_line(r: Res, ball):  # :32-55
    img = r.cam.field2image((ball["pos"][0], ball["pos"][1], F(r.max_bot_height)))
    p = r.cam.image2field(img, r.ball_radius)[:2]
    max_d = F(r.line_thickness) / F(2) + r.geometry_tolerance
    for (x1, y1, x2, y2) in r.lines:
        vx, vy = F(x2) - F(x1), F(y2) - F(y1)
        wx, wy = p[0] - F(x1), p[1] - F(y1)
        t = max(F(0), min(F(1), (wx * vx + wy * vy) / (vx * vx + vy * vy)))
        ex, ey = wx - t * vx, wy - t * vy
        if ex * ex + ey * ey <= max_d * max_d:
            return True
    for (cx, cy, rad, a1, a2) in r.arcs:
        qx, qy = p[0] - F(cx), p[1] - F(cy)
        ang = atan2f(qy, qx)
        if ang < 0:
            ang = F(float(ang) + 2 * math.pi)
        if fabsf(sqrtf(qx * qx + qy * qy) - F(rad)) <= max_d and ang >= F(a1) and ang <= F(a2):
            return True
    return False


def update_colors(r: Res, bots, balls):  # :78-141
    old = dict(r.colors)
    centers, sums, counts = [], {"green": [0, 0, 0], "pink": [0, 0, 0]}, {"green": 0, "pink": 0}
    for b in bots:
        if b["blobs"][0] is not None:
            centers.append(b["blobs"][0]["color"])
        bid = _cmod(b["botId"], 16)
        for i in range(1, 5):
            m = b["blobs"][i]
            if m is None:
                continue
            key = "green" if (PATTERNS[bid] >> (4 - i)) & 1 else "pink"
            for k in range(3):
                sums[key][k] += m["color"][k]
            counts[key] += 1
    for key in ("pink", "green"):
        if counts[key] > 0:
            r.colors[key] = _mix(r, r.refs[key], old[key], tuple(_idiv(s, counts[key]) for s in sums[key]))
    ok, c1, c2 = kmeans(r.colors["pink"], centers, r.colors["yellow"], r.colors["blue"])
    r.colors["yellow"], r.colors["blue"] = c1, c2
    if ok:
        r.colors["yellow"] = _mix(r, r.refs["yellow"], old["yellow"], r.colors["yellow"])
        r.colors["blue"] = _mix(r, r.refs["blue"], old["blue"], r.colors["blue"])
    ok, c1, c2 = kmeans(r.colors["blue"], [b["blob"]["center"] for b in balls], r.colors["orange"], r.colors["field"])
    r.colors["orange"], r.colors["field"] = c1, c2
    if ok:
        r.colors["orange"] = _mix(r, r.refs["orange"], old["orange"], r.colors["orange"])
        r.colors["field"] = _mix(r, r.refs["field"], old["field"], r.colors["field"])
    at_line = [b["blob"]["color"] for b in balls if ball_at_line(r, b)]
    if len(at_line) > 2:
        r.colors["fieldLine"] = tuple(_idiv(sum(c[k] for c in at_line), len(at_line)) for k in range(3))


def _cmod(a, b):
    """C remainder (sign of the dividend)."""
    return a - b * _idiv(a, b)


# ---- per-frame sequence (main.cpp:43-241, 319-362) -------------------------------------------------------------------
def _goal_boundary(r: Res):  # CameraModel.cpp:20-22
    return F(r.boundary_goal_line if r.boundary_goal_line >= 0 else r.boundary_width)


def detect_frame(r: Res, matches, frame_id, start_time, img_timestamp):
    """matches: list of dicts {pos:(F,F), color:(i,i,i), center:(i,i,i), circ:F, score:F}.  Returns the detection frame as a dict."""
    bots, balls = [], []
    if matches:
        pts = [m["pos"] for m in matches]
        tree = kd_build(pts)

        # tracked hypotheses, main.cpp:85-139
        for cam in sorted(r.tracked):
            for tr in r.tracked[cam]:
                if tr["id"] == -1:
                    continue
                dt = F(start_time - tr["timestamp"])
                rp = r.cam.image2field(r.cam.field2image((F(tr["x"]), F(tr["y"]), F(tr["z"]))), F(r.max_bot_height))
                tpos = (rp[0] + F(tr["vx"]) * dt, rp[1] + F(tr["vy"]) * dt, F(tr["w"]) + F(tr["vw"]) * dt)
                dt = max(min(dt, F(0.05)), F(0))
                radius = F(r.max_bot_acceleration) * dt * dt + F(r.min_tracking_radius)
                cand = []
                for i in range(5):
                    off = rot(tpos[2], PATTERN_POS[i])
                    cand.append([None] + [matches[k] for k in kd_range(tree, pts, (tpos[0] + off[0], tpos[1] + off[1]), radius, [])])
                best, best_score = None, F(0)
                for a in cand[0]:
                    for b in cand[1]:
                        if b is not None and a is b:
                            continue
                        for c in cand[2]:
                            if c is not None and (a is c or b is c):
                                continue
                            for d in cand[3]:
                                if d is not None and (a is d or b is d or c is d):
                                    continue
                                for e in cand[4]:
                                    if e is not None and (a is e or b is e or c is e or d is e):
                                        continue
                                    h = bot_tracked(r, tr, tpos, (a, b, c, d, e))
                                    if h["score"] > best_score:
                                        best, best_score = h, h["score"]
                if best is not None:
                    bots.append(best)

        # angle-sorted detection hypotheses, main.cpp:43-83
        for i, blob in enumerate(matches):
            near = kd_range(tree, pts, blob["pos"], r.max_robot_radius, [])
            if len(near) < 4:
                continue
            keys = {k: atan2_fast(pts[k][1] - blob["pos"][1], pts[k][0] - blob["pos"][0]) for k in near}
            near = sorted(near, key=lambda k: keys[k])  # the tests use inputs without equal angles (std::sort is not stable)
            n = len(near)
            best, best_score = None, F(0)
            for a in range(n):
                for b in range(a + 1, a + n - 2):
                    for c in range(b + 1, a + n - 1):
                        for d in range(c + 1, a + n):
                            h = bot_new((blob, matches[near[a]], matches[near[b % n]], matches[near[c % n]], matches[near[d % n]]))
                            calc_bot_id(r, h)
                            if h["score"] > best_score:
                                best, best_score = h, h["score"]
            if best is not None:
                bots.append(best)

        bots = [b for b in bots if not b["score"] <= r.min_confidence]  # :141-150
        # :189-219 (the first inner loop of the reference compares a hypothesis with itself and never fires)
        gone = [False] * len(bots)
        for i in range(len(bots)):
            if gone[i]:
                continue
            for j in range(len(bots)):
                if j != i and not gone[j] and bots[j]["score"] <= bots[i]["score"] and clip_bot_bot(r, bots[i], bots[j]):
                    gone[j] = True
        bots = [b for b, g in zip(bots, gone) if not g]
        for m in matches:  # :221-239
            ball = {"blob": m, "pos": m["pos"], "score": F(1)}
            ball_color_score(r, ball)
            if not any(clip_bot_ball(r, b, ball) for b in bots):
                balls.append(ball)

    update_colors(r, bots, balls)
    for b in bots:
        if b["tracked"] is not None:
            calc_tracking_score(r, b)
        else:
            calc_bot_id(r, b)
    for ball in balls:
        ball["score"] = F(1)
        ball_color_score(r, ball)

    balls = [b for b in balls if not b["score"] <= r.min_confidence]
    balls = [b for b in balls if not _ball_at_cam_edge(r, b)]
    balls = [b for b in balls if not b["blob"]["score"] <= F(r.min_score)]

    frame = {"frame_number": frame_id, "t_capture": start_time, "t_sent": 0.0, "t_capture_camera": img_timestamp if img_timestamp != 0 else None, "camera_id": r.cam_id,
             "balls": [], "robots_yellow": [], "robots_blue": [], "t_offsets": [F(o) for o in r.offsets]}
    for b in bots:  # hypothesis.cpp:140-153
        yellow = b["botId"] < 16
        img = r.cam.field2image((b["pos"][0], b["pos"][1], F(r.max_bot_height)))
        p = r.cam.image2field(img, F(r.yellow_bot_height if yellow else r.blue_bot_height))
        frame["robots_yellow" if yellow else "robots_blue"].append(
            {"confidence": b["score"], "robot_id": _cmod(b["botId"], 16), "x": p[0], "y": p[1], "orientation": b["orientation"], "pixel_x": img[0], "pixel_y": img[1], "height": p[2]})
    for b in balls:  # hypothesis.cpp:69-80
        img = r.cam.field2image((b["pos"][0], b["pos"][1], F(r.max_bot_height)))
        p = r.cam.image2field(img, r.ball_radius)
        frame["balls"].append({"confidence": b["score"], "x": p[0], "y": p[1], "pixel_x": img[0], "pixel_y": img[1]})
    return frame


def _ball_at_cam_edge(r: Res, ball):  # main.cpp:158-187
    half_len = F(_idiv(r.field_length, 2)) + _goal_boundary(r)
    half_wid = F(_idiv(r.field_width, 2) + r.boundary_width)
    img = r.cam.field2image((ball["pos"][0], ball["pos"][1], F(r.max_bot_height)))
    for border in ((F(0), img[1]), (F(r.cam.size[0] - 1), img[1]), (img[0], F(0)), (img[0], F(r.cam.size[1] - 1))):
        bp = r.cam.image2field(border, F(r.max_bot_height))
        inside = bp[0] >= -half_len and bp[0] <= half_len and bp[1] >= -half_wid and bp[1] <= half_wid
        dx, dy = bp[0] - ball["pos"][0], bp[1] - ball["pos"][1]
        if inside and float(dx * dx + dy * dy) < r.min_cam_edge_distance * r.min_cam_edge_distance:
            return True
    return False


# ---- tracker (udpsocket.cpp:165-256) ---------------------------------------------------------------------------------
def detection_tracking(r: Res, frame):
    ts = frame["t_capture"]
    prev = r.tracked.get(frame["camera_id"], [])
    objs = []

    def follow(x, y, z, w, conf, ident, use_z):
        best, best_d = None, math.inf
        for old in prev:
            if old["id"] != ident:
                continue
            dx, dy = F(x) - F(old["x"]), F(y) - F(old["y"])
            d = dx * dx + dy * dy
            if use_z:
                dz = F(z) - F(old["z"])
                d = d + dz * dz
            d = float(d)
            if d > best_d:
                continue
            best, best_d = old, d
        if best is None:
            return {"id": ident, "timestamp": ts, "x": F(x), "y": F(y), "z": F(z), "w": F(w), "vx": F(0), "vy": F(0), "vz": F(0), "vw": F(0), "confidence": F(conf), "age": 1}
        dt = F(ts - best["timestamp"])
        return {"id": ident, "timestamp": ts, "x": F(x), "y": F(y), "z": F(z), "w": F(w), "vx": (F(x) - F(best["x"])) / dt, "vy": (F(y) - F(best["y"])) / dt,
                "vz": (F(z) - F(best["z"])) / dt if use_z else F(0), "vw": F(0) if use_z else (F(w) - F(best["w"])) / dt, "confidence": F(conf), "age": best["age"] + 1}

    for b in frame["balls"]:
        objs.append(follow(b["x"], b["y"], r.ball_radius, F(0), b["confidence"], -1, True))
    for key, off in (("robots_yellow", 0), ("robots_blue", 16)):
        for b in frame[key]:
            objs.append(follow(b["x"], b["y"], b["height"], b["orientation"], b["confidence"], int(b["robot_id"]) + off, False))
    r.tracked[frame["camera_id"]] = objs


# ---- Perspective::geometryCheck (Perspective.cpp:35-125) ---------------------------------------------------------------
def geometry_check(cam: Camera, field_length, field_width, boundary_width, boundary_goal_line, ball_radius, geometry_tolerance, max_bot_height, resampling_factor):
    """Kernel scalars from a camera model (already at the image size): fieldScale, visibleFieldExtent, reprojectedFieldSize,
    min/max blob radius and the three launch radii (Resources.cpp:160,163; main.cpp:289).  Pure-Python loops: small images only."""
    w, h = cam.size
    hgt = F(max_bot_height)
    goal_b = F(boundary_goal_line if boundary_goal_line >= 0 else boundary_width)
    lim_x = F(field_length) / F(2) + goal_b
    lim_y = F(field_width) / F(2) + F(boundary_width)
    total, n = F(0), 0
    for y in range(h - 1):
        for x in range(w - 1):
            p = cam.image2field((F(x), F(y)), hgt)
            if fabsf(p[0]) < lim_x and fabsf(p[1]) < lim_y:
                q = cam.image2field((F(x + 1), F(y)), hgt)
                s = cam.image2field((F(x), F(y + 1)), hgt)
                dx = sqrtf((q[0] - p[0]) * (q[0] - p[0]) + (q[1] - p[1]) * (q[1] - p[1]))
                dy = sqrtf((s[0] - p[0]) * (s[0] - p[0]) + (s[1] - p[1]) * (s[1] - p[1]))
                total = total + (dx + dy)
                n += 2
    with np.errstate(all="ignore"):
        field_scale = total / F(n) * F(resampling_factor)
    c = cam.image2field((F(0), F(0)), hgt)
    ext = [c[0], c[0], c[1], c[1]]

    def upd(p):  # NaN never updates (all comparisons false)
        if p[0] < ext[0]: ext[0] = p[0]
        if p[0] > ext[1]: ext[1] = p[0]
        if p[1] < ext[2]: ext[2] = p[1]
        if p[1] > ext[3]: ext[3] = p[1]

    for x in range(w):
        upd(cam.image2field((F(x), F(0)), hgt))
        upd(cam.image2field((F(x), F(h) - F(1)), hgt))
    for y in range(h):
        upd(cam.image2field((F(0), F(y)), hgt))
        upd(cam.image2field((F(w) - F(1), F(y)), hgt))
    half_len = F(field_length) / F(2) + goal_b + F(geometry_tolerance)
    half_wid = F(field_width) / F(2) + F(boundary_width) + F(geometry_tolerance)
    ext[0], ext[1] = max(ext[0], -half_len), min(ext[1], half_len)
    ext[2], ext[3] = max(ext[2], -half_wid), min(ext[3], half_wid)
    size = [int(np.rint((ext[1] - ext[0]) / field_scale)), int(np.rint((ext[3] - ext[2]) / field_scale))]
    size = [s + 1 if _cmod(s, 2) else s for s in size]
    min_r = min(F(25.0), F(20.0), F(ball_radius))  # pattern.h:55-56 and the ball
    max_r = max(F(25.0), F(20.0), F(ball_radius))
    return {"field_scale": field_scale, "extent": ext, "fw": size[0], "fh": size[1], "min_blob_radius": min_r, "max_blob_radius": max_r,
            "grad_offset": _idiv(int(math.ceil(max_r / field_scale)), 3), "circle_radius": int(math.ceil(min_r / field_scale)), "list_radius": int(math.floor(min_r / field_scale))}
