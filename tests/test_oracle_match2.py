"""Known-answer tests that pin the ORACLE's windowed matchers (oracle/ovo_match2.cc) to hand-worked cases: the grid rules of
data::common, match::angle_checker, and the accept / claim / steal rules of projection, area and bow_tree. (Upstream holds no
fixtures for these paths -- SURVEY.md 8(c) -- so the expected values below are derived by hand from the rules in
ORACLE_SPEC.md.)"""
import numpy as np
import pytest


def _kps(oracle, xs, ys, octaves=None, angles=None):
    k = np.zeros(len(xs), oracle.KP_DTYPE)
    k["x"], k["y"] = xs, ys
    k["octave"] = 0 if octaves is None else octaves
    k["angle"] = 0 if angles is None else angles
    return k


def _desc(bits_set):
    """32-byte descriptor with the given bit indices set."""
    b = np.zeros(256, np.uint8)
    b[list(bits_set)] = 1
    return np.packbits(b, bitorder="little")


def test_grid_cell_rule_is_cvround(oracle):
    gp = oracle.grid_params(640, 480, 64, 48)   # 10 x 10 px cells
    # x = 14.9 -> cvRound(1.49) = 1; x = 15.0 -> cvRound(1.5) = 2 (half to even); x = 25.0 -> cvRound(2.5) = 2; 635 -> 64 = outside
    k = _kps(oracle, [14.9, 15.0, 25.0, 635.0, 4.0], [0.0, 0.0, 0.0, 0.0, 475.1])
    start, items = oracle.assign_keypoints_to_grid(gp, k)
    cell = {int(i): c for c in range(64 * 48) for i in items[start[c]:start[c + 1]]}
    assert cell == {0: 1 * 48 + 0, 1: 2 * 48 + 0, 2: 2 * 48 + 0}   # keypoint 3: cx = 64, keypoint 4: cy = cvRound(47.51) = 48 -> no cell
    assert items[start[2 * 48]:start[2 * 48 + 1]].tolist() == [1, 2]   # ascending inside a cell


def test_get_keypoints_in_cell_order_and_filters(oracle):
    gp = oracle.grid_params(640, 480, 64, 48)
    xs = [100.0, 111.0, 100.0, 104.9, 105.0, 100.0]
    ys = [100.0, 100.0, 111.0, 100.0, 100.0, 100.0]
    oc = [0, 0, 0, 0, 0, 3]
    k = _kps(oracle, xs, ys, oc)
    # margin 5 around (100, 100): |dx| < 5 strictly -> 3 is in, 4 (dx = 5.0) is out; 1 and 2 are 11 px away
    assert oracle.get_keypoints_in_cell(gp, k, 100.0, 100.0, 5.0).tolist() == [0, 3, 5]   # cells x-major: (10,10) holds 0, 3, 5
    assert oracle.get_keypoints_in_cell(gp, k, 100.0, 100.0, 5.0, 0, 0).tolist() == [0, 3]          # level filter [0, 0]
    assert oracle.get_keypoints_in_cell(gp, k, 100.0, 100.0, 5.0, 1, -1).tolist() == [5]           # min level only
    # margin 12: cells x-major then y. x = 105.0 -> cvRound(10.5) = 10 (half to even), so (10,10) = {0,3,4,5}; (10,11) = {2}; (11,10) = {1}
    assert oracle.get_keypoints_in_cell(gp, k, 100.0, 100.0, 12.0).tolist() == [0, 3, 4, 5, 2, 1]


def test_angle_checker_keeps_three_fullest_bins(oracle):
    # bin = cvRound(delta / 30) after wrapping into [0, 360): 10 -> 0, 20 -> 1 (0.667), 44 -> 1, 46 -> 2, -10 -> 350 -> 12
    deltas = [10, 10, 10, 20, 44, 46, 46, 46, 46, -10, 100, 370]   # bins: 0,0,0,1,1,2,2,2,2,12,3,0
    inv = oracle.angle_checker_invalid(np.array(deltas, np.float32))
    # sizes: bin0 = 4 (incl. 370 -> 10), bin2 = 4, bin1 = 2, bin3 = 1, bin12 = 1 -> keep {0, 2, 1}
    assert inv.tolist() == [False, False, False, False, False, False, False, False, False, True, True, False]
    # tie for the third place: bins 3 and 12 both hold one entry -> the lower bin (3) is kept
    inv = oracle.angle_checker_invalid(np.array([10, 10, 46, 46, 100, -10], np.float32))
    assert inv.tolist() == [False, False, False, False, False, True]


def test_projection_rules(oracle):
    gp = oracle.grid_params(640, 480, 64, 48)
    sf = np.array([1.0, 1.2, 1.44], np.float32)
    # three frame keypoints close together, all level 1
    k = _kps(oracle, [200.0, 202.0, 204.0], [200.0, 200.0, 200.0], [1, 1, 1])
    d = np.stack([_desc(range(10)), _desc(range(30)), _desc(range(120))])
    lm_desc = np.stack([_desc([]), _desc([]), _desc([]), _desc([])])
    xy = np.array([[201, 200]] * 4, np.float32)
    lv = np.array([1, 1, 1, 2], np.int32)
    # landmark 0: best 10 (kp 0), second 30 (kp 1), same level: 10 > 0.6 * 30 = 18? no -> accept kp 0
    # landmark 1: kp 0 claimed; best 30 (kp 1), second 120 (kp 2): 30 > 72? no -> accept kp 1
    # landmark 2: best 120 > THR_HIGH -> none.  landmark 3: level 2 -> levels [1, 2], radius 5 * 1.44; only kp 2 left: 120 -> none
    a, n = oracle.projection_match_frame_and_landmarks(gp, k, d, sf, xy, lv, lm_desc, 5.0, 0.6)
    assert a.tolist() == [0, 1, -1, -1] and n == 2
    # ratio test only bites when best and second are on the SAME level
    d2 = np.stack([_desc(range(20)), _desc(range(22)), _desc(range(200))])
    a, n = oracle.projection_match_frame_and_landmarks(gp, k, d2, sf, xy[:1], lv[:1], lm_desc[:1], 5.0, 0.6)
    assert a.tolist() == [-1] and n == 0      # 20 > 0.6 * 22
    k2 = k.copy()
    k2["octave"][1] = 0                        # second best now on another level -> accepted
    a, n = oracle.projection_match_frame_and_landmarks(gp, k2, d2, sf, xy[:1], lv[:1], lm_desc[:1], 5.0, 0.6)
    assert a.tolist() == [0] and n == 1
    # occupied keypoints are skipped; invalid landmarks are skipped; stereo: |x_right difference| must be <= radius
    a, n = oracle.projection_match_frame_and_landmarks(gp, k, d, sf, xy[:1], lv[:1], lm_desc[:1], 5.0, 0.6, frm_occupied=[1, 0, 0])
    assert a.tolist() == [1]
    a, n = oracle.projection_match_frame_and_landmarks(gp, k, d, sf, xy[:1], lv[:1], lm_desc[:1], 5.0, 0.6, lm_valid=[0])
    assert a.tolist() == [-1]
    a, n = oracle.projection_match_frame_and_landmarks(gp, k, d, sf, xy[:1], lv[:1], lm_desc[:1], 5.0, 0.6,
                                                       frm_stereo_x_right=[150.0, 180.0, -1.0], lm_x_right=[187.0])
    assert a.tolist() == [-1]   # kp 0 is 37 px off (> 6), kp 1 is 7 px off (> 6), kp 2 (mono, 120) fails THR_HIGH
    a, n = oracle.projection_match_frame_and_landmarks(gp, k, d, sf, xy[:1], lv[:1], lm_desc[:1], 5.0, 0.6,
                                                       frm_stereo_x_right=[150.0, 182.0, -1.0], lm_x_right=[187.0])
    assert a.tolist() == [1]    # kp 1 is 5 px off (<= 6)


def test_area_steal_rule(oracle):
    gp = oracle.grid_params(640, 480, 64, 48)
    k1 = _kps(oracle, [100.0, 101.0, 102.0, 300.0], [100.0] * 4, [0, 0, 0, 1])
    k2 = _kps(oracle, [100.0, 140.0], [100.0, 100.0])
    d2 = np.stack([_desc([]), _desc(range(100, 140))])
    d1 = np.stack([_desc(range(8)), _desc(range(3)), _desc(range(5)), _desc([])])
    prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], 1), np.float32)
    # query 0 takes target 0 at distance 8; query 1 (distance 3 < 8) steals it; query 2 (distance 5 >= 3) sees target 0 as taken,
    # its only other candidate is 40 away (distance 40+5) -> best 45 <= 50 and 256*0.9 >= 45 -> matches target 1; query 3 is level 1
    n, m = oracle.area_match_in_consistent_area(gp, k1, d1, k2, d2, prev, 50, 0.9, False)
    assert m.tolist() == [-1, 0, 1, -1] and n == 2
    assert prev[1].tolist() == [100.0, 100.0] and prev[2].tolist() == [140.0, 100.0] and prev[0].tolist() == [100.0, 100.0]


def test_bow_rules(oracle):
    k = _kps(oracle, [0.0] * 4, [0.0] * 4)
    kd = np.stack([_desc(range(4)), _desc(range(6)), _desc(range(200)), _desc([])])
    fd = np.stack([_desc([]), _desc(range(60)), _desc([])])
    kf_fv = {5: [0, 1], 9: [2], 11: [3]}
    fr_fv = {5: [0, 1], 7: [2], 11: [2]}
    # node 5: kf 0 -> frame 0 (4, second 56+...): accept; kf 1: frame 0 claimed, frame 1 at |6 xor 60| = 54 > 50 -> none
    # node 9 / 7: no partner.  node 11: kf 3 -> frame 2 at 0
    n, m = oracle.bow_match_frame_and_keyframe(k, kd, kf_fv, k[:3], fd, fr_fv, 0.75, False)
    assert m.tolist() == [0, -1, 3] and n == 2
    n, m = oracle.bow_match_frame_and_keyframe(k, kd, kf_fv, k[:3], fd, fr_fv, 0.75, False, kf_has_landmark=[0, 1, 1, 1])
    assert m.tolist() == [1, -1, 3] and n == 2   # kf 0 has no landmark: kf 1 gets frame 0 (distance 6)
