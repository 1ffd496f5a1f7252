"""The C++ class shims (openvslam_amd/cpp: feature::orb_extractor and match::{robust, area, projection, bow_tree, stereo} with
upstream's signatures) produce the oracle's results when driven the way tracking / initialisation code drives them."""
import os
import subprocess

import numpy as np
import pytest

from openvslam_amd.synth import synth_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "openvslam_amd", "cpp", "test_shim")


def test_shim_builds():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "openvslam_amd", "cpp")])
    assert os.path.exists(SHIM)


@pytest.mark.gpu
def test_shim_matches_oracle(oracle, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "openvslam_amd", "cpp")])
    rows, cols, nfeat = 480, 752, 1000
    a = synth_frame(rows, cols, seed=21)
    b = synth_frame(rows, cols, seed=21, shift=(4, 3), noise_seed=5)
    a.tofile(tmp_path / "a.raw")
    b.tofile(tmp_path / "b.raw")
    out = tmp_path / "out.bin"
    subprocess.check_call([SHIM, str(rows), str(cols), str(nfeat), str(tmp_path / "a.raw"), str(tmp_path / "b.raw"), str(out)])
    raw = out.read_bytes()
    na, nb, nm, r7, c7 = (int(v) for v in np.frombuffer(raw[:20], np.int32))
    off = 20
    ka = np.frombuffer(raw[off:off + 28 * na], np.uint8); off += 28 * na
    da = np.frombuffer(raw[off:off + 32 * na], np.uint8).reshape(na, 32); off += 32 * na
    kb = np.frombuffer(raw[off:off + 28 * nb], np.uint8); off += 28 * nb
    db = np.frombuffer(raw[off:off + 32 * nb], np.uint8).reshape(nb, 32); off += 32 * nb
    pairs = np.frombuffer(raw[off:off + 8 * nm], np.int32).reshape(nm, 2); off += 8 * nm
    n_area, n_proj, n_bow, n_st, n_cl, n_tri = (int(v) for v in np.frombuffer(raw[off:off + 24], np.int32)); off += 24
    area_m = np.frombuffer(raw[off:off + 4 * na], np.int32); off += 4 * na
    area_prev = np.frombuffer(raw[off:off + 8 * na], np.float32).reshape(na, 2); off += 8 * na
    proj_assigned = np.frombuffer(raw[off:off + 4 * na], np.int32); off += 4 * na
    bow_m = np.frombuffer(raw[off:off + 4 * na], np.int32); off += 4 * na
    st_x = np.frombuffer(raw[off:off + 4 * n_st], np.uint32); off += 4 * n_st
    st_d = np.frombuffer(raw[off:off + 4 * n_st], np.uint32); off += 4 * n_st
    cl_assigned = np.frombuffer(raw[off:off + 4 * na], np.int32); off += 4 * na
    tri_m = np.frombuffer(raw[off:off + 4 * na], np.int32); off += 4 * na
    assert off == len(raw)
    ox = oracle.OrbExtractor(oracle.make_params(nfeat))
    wa, wda = ox.extract(a)
    wb, wdb = ox.extract(b)
    assert (r7, c7) == ox.level_image(7).shape
    assert np.array_equal(ka, wa.view(np.uint8)) and np.array_equal(da, wda)
    assert np.array_equal(kb, wb.view(np.uint8)) and np.array_equal(db, wdb)
    valid = np.array([(i % 10 != 3) and (i % 10 != 7) for i in range(nb)], np.uint8)
    assert np.array_equal(pairs, oracle.robust_brute_force_match(wda, wdb, valid, 0.9)) and nm > 100

    # ---- windowed matchers
    gp = oracle.grid_params(cols, rows)
    prev = np.ascontiguousarray(np.stack([wa["x"], wa["y"]], 1), np.float32)
    wn, want = oracle.area_match_in_consistent_area(gp, wa, wda, wb, wdb, prev, 100, 0.9, True)
    assert n_area == wn and np.array_equal(area_m, want) and np.array_equal(area_prev, prev) and wn > 50
    sf = oracle.orb_tables(oracle.make_params(nfeat))["scale_factors"]
    reproj = np.stack([wa["x"].astype(np.float64) - 4.0, wa["y"].astype(np.float64) - 3.0], 1)
    lm_valid = np.array([i % 7 != 0 for i in range(na)], np.uint8)
    want, wn = oracle.projection_match_frame_and_landmarks(gp, wb, wdb, sf, reproj, wa["octave"], wda, 5.0, 0.8, lm_valid=lm_valid)
    assert n_proj == wn and np.array_equal(proj_assigned, want) and wn > 50
    fv_b = {}
    for i in range(nb):
        fv_b.setdefault(int(wdb[i, 0]) & 127, []).append(i)
    fv_a = {}
    for i in range(na):
        fv_a.setdefault(int(wda[i, 0]) & 127, []).append(i)
    wn, want = oracle.bow_match_frame_and_keyframe(wb, wdb, fv_b, wa, wda, fv_a, 0.75, True, valid)
    assert n_bow == wn and np.array_equal(bow_m, want) and wn > 20
    oxa, oxb = oracle.OrbExtractor(oracle.make_params(nfeat)), oracle.OrbExtractor(oracle.make_params(nfeat))
    oxa.extract(a)
    oxb.extract(b)
    wx, wd, _ = oracle.stereo_compute(oxa, oxb, wa, wda, wb, wdb, 386.1448, 0.5372)
    assert n_st == na and np.array_equal(st_x, wx.view(np.uint32)) and np.array_equal(st_d, wd.view(np.uint32))

    # ---- projection::match_current_and_last_frames (identity poses, landmarks back-projected from the shifted keypoints)
    fx = fy = 500.0
    cx, cy = cols / 2.0, rows / 2.0
    ocam = oracle.Camera(0, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    idx = np.arange(na)
    z = 2.0 + (idx % 7).astype(np.float64)
    pos = np.stack([((wa["x"].astype(np.float64) - 4.0) - cx) / fx * z, ((wa["y"].astype(np.float64) - 3.0) - cy) / fy * z, z], 1)
    last_valid = ((idx % 13 != 5) & (idx % 11 != 0)).astype(np.uint8)
    T = np.eye(4)[:3]
    want, wn = oracle.projection_match_current_and_last_frames(ocam, gp, wb, wdb, T, wa, pos, wda, T, sf, 15.0, True, last_valid=last_valid)
    assert n_cl == wn and np.array_equal(cl_assigned, want) and wn > 100

    # ---- robust::match_for_triangulation
    def bearings(k):
        vx, vy = (k["x"].astype(np.float64) - cx) / fx, (k["y"].astype(np.float64) - cy) / fy
        nrm = np.sqrt((vx * vx + vy * vy) + 1.0)
        return np.stack([vx / nrm, vy / nrm, 1.0 / nrm], 1)

    t12 = np.array([0.2, 0.15, 0.0])
    E12 = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    ep = np.array([-0.2, -0.15, 0.0])
    ep = ep / np.sqrt((ep[0] * ep[0] + ep[1] * ep[1]) + ep[2] * ep[2])
    h1 = (np.arange(na) % 3 == 0).astype(np.uint8)
    h2 = (np.arange(nb) % 4 == 0).astype(np.uint8)
    wn, want = oracle.robust_match_for_triangulation(wa, wda, fv_a, bearings(wa), wb, wdb, fv_b, bearings(wb), E12, ep, sf, True, has_lm_1=h1,
                                                     has_lm_2=h2)
    assert n_tri == wn and np.array_equal(tri_m, want) and wn > 20
