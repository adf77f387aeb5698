"""Perspective::geometryCheck / CameraModel (Perspective.cpp:35-150, CameraModel.cpp:86-172): the Eigen-free C++ producer in
libvp_host.so against the independent numpy restatement in oracle/detect_oracle.py, bit for bit.  (Parity against the reference
itself is unpinned: its host code cannot be built here.)"""
import numpy as np
import pytest

import cases
import detect_oracle as O

H = cases.vp.hostlib
F = np.float32

CASES = {
    # calib (focal, ppx, ppy, dist, q0..q3, tx, ty, tz), calib size, image size, field (len, wid, boundary, goal boundary), factor
    "nadir": ((60.0, 32.0, 24.0, 0.08, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 5000.0), (64, 48), (64, 48), (4000, 3000, 300, -1), 1.0),
    "nadir_half_res": ((60.0, 32.0, 24.0, 0.08, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 5000.0), (64, 48), (64, 48), (4000, 3000, 300, -1), 0.5),
    "tilted_goal_boundary": ((70.0, 30.0, 26.0, -0.05, -0.9952, 0.03, 0.02, 0.09, 150.0, -80.0, 4200.0), (64, 48), (64, 48), (9000, 6000, 300, 250), 1.0),
    "rescaled_sensor": ((120.0, 64.0, 48.0, 0.1, -1.0, 0.0, 0.0, 0.0, 100.0, 50.0, 4000.0), (128, 96), (64, 48), (3000, 2000, 200, -1), 1.0),  # ensureSize
    "field_smaller_than_view": ((40.0, 32.0, 24.0, 0.0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 6000.0), (64, 48), (64, 48), (1500, 1000, 100, -1), 0.75),  # extent clamps
}


def bits(v):
    return np.asarray(v, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", sorted(CASES))
def test_geometry_check_matches_the_independent_restatement(name):
    calib, csize, isize, field, factor = CASES[name]
    got = H.geometry_check(calib, isize[0], isize[1], csize[0], csize[1], field, ball_radius=21.5, geometry_tolerance=10.0, max_bot_height=180.0, resampling_factor=factor)
    cam = O.Camera(calib, csize, isize)
    want = O.geometry_check(cam, field[0], field[1], field[2], field[3], 21.5, 10.0, 180.0, factor)
    assert got["geometry_version"] == 1
    assert bits(got["field_scale"]) == bits(want["field_scale"]), (got["field_scale"], want["field_scale"])
    assert np.array_equal(bits(got["extent"]), bits(want["extent"])), (got["extent"], want["extent"])
    assert (got["fw"], got["fh"]) == (want["fw"], want["fh"]) and got["fw"] % 2 == 0 and got["fh"] % 2 == 0
    assert bits(got["min_blob_radius"]) == bits(want["min_blob_radius"]) and bits(got["max_blob_radius"]) == bits(want["max_blob_radius"])
    assert (got["grad_offset"], got["circle_radius"], got["list_radius"]) == (want["grad_offset"], want["circle_radius"], want["list_radius"])
    # the packed kernel argument (Perspective.cpp:136-150): focal length, principal point, distortion, rotation rows, camera position
    m = got["model"]
    assert (m.shape[0], m.shape[1]) == isize
    assert bits(m.f) == bits(cam.f) and bits(m.p[0]) == bits(cam.pp[0]) and bits(m.p[1]) == bits(cam.pp[1]) and bits(m.d) == bits(cam.k2)
    assert np.array_equal(bits(list(m.r)), bits([v for row in cam.f2i for v in row]))
    assert np.array_equal(bits(list(m.c)), bits(list(cam.pos)))
    # sanity of the scale: the mean footprint of a pixel on the plane at robot height
    assert 0.2 * factor < float(got["field_scale"]) / (float(cam.pos[2] - 180.0) / float(cam.f)) < 2.0 * factor


def test_flat2field_roundtrip_and_detector_geometry_agree():
    calib, csize, isize, field, factor = CASES["tilted_goal_boundary"]
    g = H.geometry_check(calib, isize[0], isize[1], csize[0], csize[1], field, resampling_factor=factor)
    d = H.Detector()
    d.set_geometry(calib, csize, isize, (field[0], field[1], field[2], field[3], 10), resampling_factor=factor)
    capi = cases.vp.load_capi()
    cl = np.zeros(3, capi.MATCH_DTYPE)
    cl["x"], cl["y"] = [0.0, 10.5, 63.25], [0.0, 7.75, 40.0]
    fm = d.matches_from_cl(cl)
    fs, ox, oy = F(g["field_scale"]), F(g["extent"][0]), F(g["extent"][2])
    assert np.array_equal(bits(fm["x"]), bits(cl["x"] * fs + ox)) and np.array_equal(bits(fm["y"]), bits(cl["y"] * fs + oy))  # Perspective.cpp:127-129
