"""Standard parameter sets of the hot path (SURVEY.md section 8 header and 8(d)) plus small variants for parity tests.

The kernel scalars are set directly (not via Perspective::geometryCheck) exactly as the three reference call sites
derive them: grad offset = (int)ceilf(maxBlobRadius/fieldScale)/3 (Resources.cpp:160), circle radius =
(int)ceilf(minBlobRadius/fieldScale) (Resources.cpp:163), list radius = (int)floorf(minBlobRadius/fieldScale) (main.cpp:289).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, replace

import numpy as np

from . import hostlib

RGGB, GRBG, BGR = 0, 1, 2
SAMPLER_NEAREST, SAMPLER_BILINEAR_RTZ = 0, 1
MIN_BLOB_RADIUS, MAX_BLOB_RADIUS = 20.0, 25.0  # pattern.h:55-56 with ball_radius 21.5


@dataclass(frozen=True)
class FrameConfig:
    name: str
    layout: int
    qw: int            # RawImage.width  (Bayer: 2x2 cells)
    qh: int
    fw: int            # reprojectedFieldSize
    fh: int
    field_scale: float
    off_x: float       # visibleFieldExtent[0]
    off_y: float       # visibleFieldExtent[2]
    visible_len: float  # mm of the plane mapped onto the quad-plane width
    field_len: float   # painted field (lines) for the synthetic scene
    field_wid: float
    cam_height: float = 5000.0
    max_robot_height: float = 180.0
    distortion: float = 0.08
    n_robots: int = 16
    n_balls: int = 8
    noise_sigma: float = 3.0
    circ_threshold: float = 15.0
    min_score: float = 0.0
    max_matches: int = 2000
    sampler_mode: int = SAMPLER_BILINEAR_RTZ

    @property
    def grad_offset(self) -> int:
        return int(math.ceil(np.float32(MAX_BLOB_RADIUS) / np.float32(self.field_scale))) // 3

    @property
    def circle_radius(self) -> int:
        return int(math.ceil(np.float32(MIN_BLOB_RADIUS) / np.float32(self.field_scale)))

    @property
    def list_radius(self) -> int:
        return int(math.floor(np.float32(MIN_BLOB_RADIUS) / np.float32(self.field_scale)))

    @property
    def sensor_shape(self):
        return (self.qh, self.qw, 3) if self.layout == BGR else (2 * self.qh, 2 * self.qw)

    @property
    def raw_bytes(self) -> int:
        return int(np.prod(self.sensor_shape))

    @property
    def n_flat(self) -> int:
        return self.fw * self.fh

    def model(self) -> hostlib.VpCameraModel:
        return hostlib.synth_camera(self.qw, self.qh, self.cam_height, self.max_robot_height, self.visible_len, self.distortion)

    def synth(self, seed: int) -> np.ndarray:
        return hostlib.synth_frame(self.model(), self.layout, self.qw, self.qh, self.max_robot_height, self.field_len, self.field_wid,
                                   0x5EED0000 + seed, self.n_robots, self.n_balls, self.noise_sigma)

    def truth(self, seed: int) -> np.ndarray:
        """Ground truth of synth(seed): (n, 6) {x, y, radius, r, g, b} discs in field mm on the plane z = max_robot_height."""
        return hostlib.synth_truth(0x5EED0000 + seed, self.n_robots, self.n_balls, self.field_len, self.field_wid)

    def field_markings(self):
        """The painted markings of the synthetic scene (host/synth.cpp shade(): touch lines, goal lines, halfway line, centre
        circle) as SSL_FieldLineSegment / SSL_FieldCircularArc data: (lines n x {p1x, p1y, p2x, p2y}, arcs n x {cx, cy, radius, a1, a2})."""
        hl, hw = self.field_len / 2.0, self.field_wid / 2.0
        lines = [(-hl, -hw, hl, -hw), (-hl, hw, hl, hw), (-hl, -hw, -hl, hw), (hl, -hw, hl, hw), (0.0, -hw, 0.0, hw)]
        arcs = [(0.0, 0.0, min(self.field_len, self.field_wid) * 0.07, 0.0, 2.0 * np.pi)]
        return lines, arcs

    def with_(self, **kw) -> "FrameConfig":
        return replace(self, **kw)


# BASELINE.json configs[1]: sensor 2048x1536 RGGB -> planes 1024x768 -> flat 2048x1536 (o=2, r=5, R=4)
CFG2 = FrameConfig("cfg2_2048x1536_rggb", RGGB, 1024, 768, 2048, 1536, 4.6875, -4800.0, -3600.0, 9600.0, 9000.0, 6000.0)
# configs[4]: everything x2 per side (o=3, r=9, R=8)
CFG5 = FrameConfig("cfg5_4096x3072_rggb", RGGB, 2048, 1536, 4096, 3072, 2.34375, -4800.0, -3600.0, 9600.0, 9000.0, 6000.0)
# configs[0]: BGR8 640x480 (OpenCV-driver layout), flat 640x480, fieldScale 5 (o=1, r=4, R=4)
CFG1 = FrameConfig("cfg1_640x480_bgr", BGR, 640, 480, 640, 480, 5.0, -1600.0, -1200.0, 3200.0, 2800.0, 2000.0, n_robots=4, n_balls=3)

# small parity-test variants (seconds on the CPU oracle); low camera so the 8-iteration distortion loop matters
T_RGGB = FrameConfig("t_rggb_192x128", RGGB, 96, 64, 192, 128, 4.6875, -450.0, -300.0, 900.0, 760.0, 460.0, cam_height=630.0, n_robots=2, n_balls=2)
T_GRBG = T_RGGB.with_(name="t_grbg_192x128", layout=GRBG)
T_BGR = FrameConfig("t_bgr_160x120", BGR, 160, 120, 160, 120, 5.0, -400.0, -300.0, 800.0, 660.0, 460.0, cam_height=580.0, n_robots=2, n_balls=1)
# odd sizes / non-multiples of any tile, flat larger than the sensor footprint (clamp-to-edge sampling, H7)
T_RAGGED = FrameConfig("t_ragged_150x98", RGGB, 67, 45, 150, 98, 6.1, -470.0, -310.0, 800.0, 700.0, 420.0, cam_height=600.0, n_robots=2, n_balls=2)
T_MID = FrameConfig("t_mid_640x384", RGGB, 320, 192, 640, 384, 4.6875, -1500.0, -900.0, 3000.0, 2700.0, 1500.0, cam_height=1700.0, n_robots=5, n_balls=3)

ALL = {c.name: c for c in (CFG1, CFG2, CFG5, T_RGGB, T_GRBG, T_BGR, T_RAGGED, T_MID)}
