"""N-version check of the oracle ("Next round" #4): tests/nversion_numpy.py restates cv::resize(INTER_LINEAR, u8),
cv::FAST(TYPE_9_16) + cornerScore + NMS and the 7x7 fixed-point GaussianBlur a second time, in whole-array numpy written from
oracle/ORACLE_SPEC.md rules 3, 5, 10 without looking at oracle/ovo_orb.cc; both must agree on every pixel / keypoint of the golden
inputs, random images, odd sizes and the extractor's own pyramid levels. CPU only."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nversion_numpy as nv   # noqa: E402

from openvslam_amd import synth   # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _images():
    rng = np.random.default_rng(11)
    out = {"white_noise": rng.integers(0, 256, (203, 331), dtype=np.uint8),
           "synth_752x480": synth.synth_frame(480, 752, seed=0),
           "synth_331x203": synth.synth_frame(203, 331, seed=5),
           "odd_131x97": synth.synth_frame(97, 131, seed=3),
           "low_contrast": (118 + rng.integers(0, 14, (150, 222))).astype(np.uint8),
           "saturated": np.clip(rng.normal(128, 90, (120, 160)), 0, 255).astype(np.uint8)}
    yy, xx = np.mgrid[0:140, 0:180]
    out["checkerboard"] = (((yy // 5 + xx // 5) & 1) * 200).astype(np.uint8)   # many equal scores: NMS ties
    out["tiny_7x7"] = rng.integers(0, 256, (7, 7), dtype=np.uint8)
    out["tiny_6x9"] = rng.integers(0, 256, (6, 9), dtype=np.uint8)
    return out


IMAGES = _images()


@pytest.mark.parametrize("name", sorted(IMAGES))
def test_resize_second_restatement(oracle, name):
    img = IMAGES[name]
    H, W = img.shape
    sizes = {(max(int(round(H / 1.2)), 1), max(int(round(W / 1.2)), 1)), (max(H - 1, 1), max(W - 1, 1)), (max(H // 2, 1), max(W // 3, 1)), (H, W),
             (max(int(round(H / 1.5)), 1), max(int(round(W / 1.1)), 1))}
    for dr, dc in sorted(sizes):
        assert np.array_equal(nv.resize_linear_u8(img, dr, dc), oracle.resize_linear(img, dr, dc)), (name, dr, dc)


def test_resize_chain_is_the_extractors_pyramid(oracle):
    """The eight levels the oracle's extractor builds (each from the previous one, sizes from the original) == the numpy chain."""
    img = IMAGES["synth_752x480"]
    ox = oracle.OrbExtractor(oracle.make_params(1000))
    ox.extract(img)
    lr, lc = oracle.pyramid_sizes(oracle.make_params(1000), *img.shape)
    cur = img
    for level in range(1, 8):
        r, c = int(lr[level]), int(lc[level])
        cur = nv.resize_linear_u8(cur, r, c)
        assert np.array_equal(cur, ox.level_image(level)), level


@pytest.mark.parametrize("name", sorted(IMAGES))
@pytest.mark.parametrize("thr", [7, 20, 40])
def test_fast_second_restatement(oracle, name, thr):
    img = IMAGES[name]
    for nonmax in (True, False):
        x, y, r = nv.fast9_16(img, thr, nonmax)
        ox, oy, orr = (np.asarray(v).astype(np.int32) for v in oracle.fast9_16(img, thr, nonmax)[:3])
        assert len(x) == len(ox) and np.array_equal(x, ox) and np.array_equal(y, oy) and np.array_equal(r, orr), (name, thr, nonmax)


@pytest.mark.parametrize("name", sorted(IMAGES))
@pytest.mark.parametrize("taps", [0, 1])
def test_blur_second_restatement(oracle, name, taps):
    img = IMAGES[name]
    if min(img.shape) < 4:
        pytest.skip("BORDER_REFLECT_101 needs at least 4 pixels for a 7-tap kernel")
    assert np.array_equal(nv.gaussian_blur_7x7(img, taps), oracle.gaussian_blur(img, taps))


def test_golden_keypoints_follow_from_the_second_restatement(oracle):
    """The committed golden vectors (oracle outputs, tools/make_golden.py) against the numpy chain, with the oracle out of the loop: on
    every pyramid level (built by the numpy resize from the regenerated input) every golden keypoint is a FAST corner of the numpy
    restatement at min_fast_thr or above, its response is the numpy score, and it survives the numpy non-maximum suppression; the golden
    candidate COUNT per level is bounded by the numpy detections at min and ini threshold (the cell loop picks one of the two per cell)."""
    from openvslam_amd.feature import KP_DTYPE
    g = np.load(os.path.join(GOLDEN, "orb_752x480_seed0.npz"))
    img = synth.synth_frame(480, 752, seed=0)
    kps = g["kps_a"].view(KP_DTYPE).reshape(-1) if g["kps_a"].dtype != KP_DTYPE else g["kps_a"]
    lr, lc = oracle.pyramid_sizes(oracle.make_params(1000), 480, 752)
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    cur = img
    n_checked = 0
    for level in range(8):
        if level:
            cur = nv.resize_linear_u8(cur, int(lr[level]), int(lc[level]))
        S = nv.fast_strength(cur)
        kl = kps[kps["octave"] == level]
        xs = np.rint(kl["x"] / sf[level]).astype(int)
        ys = np.rint(kl["y"] / sf[level]).astype(int)
        assert (S[ys, xs] > 7).all(), level
        assert np.array_equal(S[ys, xs] - 1, kl["response"].astype(int)), level
        x7, y7, _ = nv.fast9_16(cur, 7, True)
        surv = set(zip(x7.tolist(), y7.tolist()))
        x20, y20, _ = nv.fast9_16(cur, 20, True)
        # a golden keypoint survives NMS inside its cell at its cell's threshold; at min threshold over the whole image it may lose to a
        # neighbour across a cell border, so only the threshold-free facts are asserted per keypoint and the counts per level
        n_cand = int(g["n_cand"][level])
        assert n_cand >= len(kl)
        n_checked += len(kl)
        assert len(surv) > 0 and len(x20) <= len(x7)
    assert n_checked == len(kps)


def test_orientation_and_descriptor_second_restatement(oracle):
    """ic_angle + cv::fastAtan2 (rule 9) and the steered rBRIEF (rule 11) in float32 numpy against the oracle's single-keypoint functions,
    on random positions of a noisy image and a synthetic frame: angles by bit pattern, descriptors by byte."""
    rng = np.random.default_rng(21)
    pat = oracle.orb_pattern()
    for img in (IMAGES["white_noise"], IMAGES["synth_752x480"]):
        H, W = img.shape
        xs = rng.integers(22, W - 22, 300)
        ys = rng.integers(22, H - 22, 300)
        tab = oracle.orb_tables(oracle.make_params(1000))
        got = nv.ic_angle(img, xs, ys)
        want = np.array([oracle.ic_angle(img, int(x), int(y), tab["u_max"]) for x, y in zip(xs, ys)], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        blurred = oracle.gaussian_blur(img)
        gd = nv.orb_descriptors(blurred, xs, ys, want, pat)
        wd = np.stack([oracle.orb_descriptor(blurred, int(x), int(y), float(a)) for x, y, a in zip(xs, ys, want)])
        assert np.array_equal(gd, wd)
    # fastAtan2 on axes, diagonals, zero and tiny arguments
    yy = np.array([0, 0, 1, -1, 1, -1, 1e-30, 5, -3, 0, 7], np.float32)
    xx = np.array([0, 1, 0, 0, 1, -1, -1e-30, 5, 3, -2, -7], np.float32)
    import ctypes as C
    f = oracle.lib().ovo_fast_atan2
    f.restype = C.c_float
    f.argtypes = [C.c_float, C.c_float]
    want = np.array([f(float(a), float(b)) for a, b in zip(yy, xx)], np.float32)
    assert np.array_equal(nv.fast_atan2_deg(yy, xx).view(np.uint32), want.view(np.uint32))


def test_golden_angles_and_descriptors_follow_from_the_numpy_chain(oracle):
    """End to end with the oracle's C code out of the loop (only its keypoint POSITIONS, i.e. the cell loop and the quad-tree, are taken from
    the golden file): numpy pyramid -> numpy ic_angle / fastAtan2 -> numpy blur -> numpy rBRIEF == the committed golden angles and descriptors of
    all 1008 keypoints of tests/golden/orb_752x480_seed0.npz."""
    from openvslam_amd.feature import KP_DTYPE
    g = np.load(os.path.join(GOLDEN, "orb_752x480_seed0.npz"))
    kps = g["kps_a"].view(KP_DTYPE).reshape(-1) if g["kps_a"].dtype != KP_DTYPE else g["kps_a"]
    desc = g["desc_a"]
    img = synth.synth_frame(480, 752, seed=0)
    lr, lc = oracle.pyramid_sizes(oracle.make_params(1000), 480, 752)
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    pat = oracle.orb_pattern()
    cur = img
    n = 0
    for level in range(8):
        if level:
            cur = nv.resize_linear_u8(cur, int(lr[level]), int(lc[level]))
        sel = np.nonzero(kps["octave"] == level)[0]
        if not len(sel):
            continue
        xs = np.rint(kps["x"][sel] / sf[level]).astype(int)
        ys = np.rint(kps["y"][sel] / sf[level]).astype(int)
        ang = nv.ic_angle(cur, xs, ys)
        assert np.array_equal(ang.view(np.uint32), kps["angle"][sel].view(np.uint32)), level
        d = nv.orb_descriptors(nv.gaussian_blur_7x7(cur), xs, ys, ang, pat)
        assert np.array_equal(d, desc[sel]), level
        n += len(sel)
    assert n == len(kps) == 1008


def test_matcher_second_restatement(oracle):
    """Rule 14 (match::robust's brute-force matcher) and the best / second-best kernel contract restated over an integer matrix product:
    the golden pairs of the two golden frames come out of the numpy form alone, and on random descriptor sets with clusters of
    near-duplicates, landmark masks and frame-side masks it equals the C oracle pair for pair."""
    g = np.load(os.path.join(GOLDEN, "orb_752x480_seed0.npz"))
    got = nv.robust_brute_force_match(g["desc_a"], g["desc_b"], None, 0.9)
    assert np.array_equal(got, g["pairs_ab_ratio09"]) and len(got) > 300
    rng = np.random.default_rng(3)
    for n1, n2, ratio in ((300, 280, 0.8), (64, 500, 0.9), (1, 40, 0.75), (200, 1, 1.01), (150, 150, 0.6)):
        base = rng.integers(0, 256, (max(n1, n2), 32), dtype=np.uint8)
        d1 = base[:n1].copy()
        d2 = base[rng.permutation(max(n1, n2))[:n2]].copy()
        for d in (d1, d2):   # a few flipped bits: most keypoints have a near neighbour on the other side, some have several
            flips = rng.integers(0, 256, (len(d), 6))
            for k in range(6):
                d[np.arange(len(d)), flips[:, k] >> 3] ^= (1 << (flips[:, k] & 7)).astype(np.uint8)
        d2[: n2 // 5] = d2[rng.integers(0, n2, n2 // 5)]   # exact duplicates on the keyframe side: claim conflicts
        kf_valid = (rng.random(n2) < 0.8).astype(np.uint8)
        frm_valid = (rng.random(n1) < 0.9).astype(np.uint8)
        for kv, fv in ((None, None), (kf_valid, None), (kf_valid, frm_valid)):
            want = oracle.robust_brute_force_match(d1, d2, kv, ratio, frm_valid=fv)
            assert np.array_equal(nv.robust_brute_force_match(d1, d2, kv, ratio, fv), want), (n1, n2, ratio)
        wi, wb, ws = oracle.hamming_best2(d2, d1, frm_valid)
        gi, gb, gs = nv.hamming_best2(d2, d1, frm_valid)
        assert np.array_equal(gb, wb) and np.array_equal(gs, ws) and np.array_equal(gi, wi)


@pytest.mark.parametrize("n,stereo_frac,outlier_frac,pose_err,seed", [(1500, 0.4, 0.1, 1.0, 1), (2000, 0.0, 0.2, 2.0, 2), (300, 1.0, 0.05, 0.5, 3),
                                                                      (60, 0.5, 0.0, 1.0, 4), (7, 0.5, 0.0, 1.0, 5), (3, 0.0, 0.0, 1.0, 6)])
def test_pose_optimizer_second_restatement(oracle, n, stereo_frac, outlier_frac, pose_err, seed):
    """optimize::pose_optimizer::optimize a second time (tests/nversion_pose.py: numpy edges and LAPACK solve, written from rules 15 / 25)
    against the C oracle: the same inlier flags and number of valid observations, the pose to 2e-8 (the normal equations are summed in a
    different order: rule 25's stated spread)."""
    import nversion_pose as nvp
    from openvslam_amd.synth import synth_pose_frame
    T0, obs, cam, bf, _ = synth_pose_frame(oracle.POSE_OBS_DTYPE, n, seed, stereo_frac, outlier_frac, pose_err)
    wT, wout, wnv = oracle.pose_optimize(T0, obs, cam, bf)
    T, out, nv = nvp.pose_optimize(T0, obs, cam, bf)
    assert nv == wnv and np.array_equal(out, wout.astype(bool))
    assert np.allclose(T, wT, rtol=0, atol=2e-8), np.abs(T - wT).max()


@pytest.mark.parametrize("n,outlier_frac,pose_err,seam,pole,seed", [(1500, 0.1, 1.0, 0.0, 0.0, 1), (2000, 0.2, 2.0, 0.1, 0.05, 2), (300, 0.05, 0.5, 0.3, 0.3, 3),
                                                                     (7, 0.0, 1.0, 0.0, 0.0, 4)])
def test_equirect_pose_optimizer_second_restatement(oracle, n, outlier_frac, pose_err, seam, pole, seed):
    """The equirectangular pose-only edge (rule 26: atan2 / asin projection, no wrap-around at the seam) through the same numpy optimiser,
    with bearings on the +-180 degree seam and near the poles, against the C oracle."""
    import nversion_pose as nvp
    from openvslam_amd.synth import synth_pose_frame_equirect
    T0, obs, cols, rows, _ = synth_pose_frame_equirect(oracle.POSE_OBS_DTYPE, n, seed, outlier_frac=outlier_frac, pose_err=pose_err,
                                                       seam_frac=seam, pole_frac=pole)
    wT, wout, wnv = oracle.pose_optimize_equirect(T0, obs, cols, rows)
    T, out, nv = nvp.pose_optimize_equirect(T0, obs, cols, rows)
    flips = int((out != wout.astype(bool)).sum())
    assert flips <= max(1, n // 500) and abs(nv - wnv) <= flips     # (the tolerance of tests/test_gpu_pose.py for these frames)
    assert np.allclose(T, wT, rtol=0, atol=1e-7), np.abs(T - wT).max()


def _blocks_close(got, want, rtol=1e-11):
    for k in ("Hpp", "bp", "Hll", "bl", "Hpl", "chi2"):
        scale = max(float(np.abs(want[k]).max()), 1e-300)
        assert np.allclose(got[k], want[k], rtol=rtol, atol=rtol * scale), (k, float(np.abs(got[k] - want[k]).max() / scale))


def test_ba_linearisation_second_restatement(oracle):
    """The blocks of one local-BA linearisation (rule 15: residuals, 2x3 / 3x3 point and 2x6 / 3x6 pose Jacobians, Huber weights, Hpp, bp,
    Hll, bl, the per-edge Hpl, both chi2 sums) from whole-array numpy (tests/nversion_pose.py: chain rule through d pi / d p) against the C
    oracle's edge loops: monocular, stereo and equirectangular edges, with fixed keyframes and with / without the robust kernel."""
    import nversion_pose as nvp
    from openvslam_amd.synth import synth_local_ba
    from test_ba import _lba_scene
    d = synth_local_ba(n_pose=8, n_pt=900, obs_per_pose=300, seed=2, pose_noise=0.03, point_noise=0.03, n_fixed=2)
    for delta in (0.0, float(np.sqrt(np.float32(5.99146)))):
        want = oracle.ba_linearize(d["poses"], d["pose_fixed"], d["points"], d["edges"], d["cam"], delta)
        _blocks_close(nvp.ba_linearize(d["poses"], d["pose_fixed"], d["points"], d["edges"], d["cam"], delta), want)
    s, mono, st, bf, _, _ = _lba_scene(4, n_pose=7, n_pt=700, obs_per_pose=250, stereo_frac=1.0)
    want = oracle.ba_linearize_stereo(s["poses"], s["pose_fixed"], s["points"], st, s["cam"], bf, 2.7955)
    _blocks_close(nvp.ba_linearize(s["poses"], s["pose_fixed"], s["points"], st, s["cam"], 2.7955, bf=bf), want)
    # equirectangular: landmarks all around the keyframes (the seam and the poles included)
    rng = np.random.default_rng(8)
    pts = rng.normal(size=(600, 3)) * 6.0
    poses = np.zeros((5, 7))
    poses[:, :3] = rng.normal(size=(5, 3)) * 0.3
    q = rng.normal(size=(5, 4)) * 0.05 + np.array([0, 0, 0, 1.0])
    poses[:, 3:] = q / np.linalg.norm(q, axis=1)[:, None]
    e = np.zeros(1500, oracle.BA_EDGE_DTYPE)
    e["pose_idx"], e["point_idx"] = rng.integers(0, 5, 1500), rng.integers(0, 600, 1500)
    e["obs_x"], e["obs_y"] = rng.uniform(0, 3840, 1500), rng.uniform(0, 1920, 1500)
    e["inv_sigma_sq"] = rng.choice([1.0, 0.69, 0.48], 1500)
    fixed = np.array([1, 0, 0, 1, 0], np.uint8)
    want = oracle.ba_linearize_equirect(poses, fixed, pts, e, 3840, 1920, 2.4477)
    _blocks_close(nvp.ba_linearize(poses, fixed, pts, e, (3840.0, 1920.0, 0.0, 0.0), 2.4477, equirect=True), want, rtol=1e-10)


def test_grid_second_restatement(oracle):
    """Rule 16 (the 64 x 48 keypoint grid behind every windowed matcher): cell assignment by cvRound in float and the area query's cell range,
    order, level filter and strict distance test, whole-array, against the C oracle -- keypoints on cell borders and outside the image, queries
    at the borders, every level-filter combination."""
    rng = np.random.default_rng(12)
    cols, rows = 752, 480
    gp = oracle.grid_params(cols, rows)
    n = 3000
    kps = np.zeros(n, oracle.KP_DTYPE)
    kps["x"] = rng.uniform(-8, cols + 8, n).astype(np.float32)
    kps["y"] = rng.uniform(-8, rows + 8, n).astype(np.float32)
    kps["x"][:400] = (np.round(kps["x"][:400] / 11.75) * 11.75).astype(np.float32)    # on cell borders (752 / 64 = 11.75)
    kps["y"][:400] = (np.round(kps["y"][:400] / 10.0) * 10.0 + rng.choice([0.0, 5.0], 400)).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n)
    start, items = oracle.assign_keypoints_to_grid(gp, kps)
    cx, cy, inside = nv.grid_cells(kps["x"], kps["y"], 0.0, 0.0, cols, rows, 64, 48)
    cell = cx * 48 + cy   # (the oracle's CSR is x-major)
    counts = np.bincount(cell[inside], minlength=64 * 48)
    assert np.array_equal(np.diff(start), counts) and len(items) == int(inside.sum())
    for c in rng.integers(0, 64 * 48, 200):
        assert np.array_equal(items[start[c]:start[c + 1]], np.nonzero(inside & (cell == c))[0])
    for _ in range(300):
        rx, ry = float(rng.uniform(-20, cols + 20)), float(rng.uniform(-20, rows + 20))
        m = float(rng.choice([3.0, 7.5, 15.0, 40.0, 120.0]))
        lo, hi = [(-1, -1), (0, -1), (2, -1), (-1, 3), (1, 4), (3, 3), (0, 0)][int(rng.integers(0, 7))]
        want = oracle.get_keypoints_in_cell(gp, kps, rx, ry, m, lo, hi)
        got = nv.keypoints_in_cell(kps["x"], kps["y"], kps["octave"], rx, ry, m, 0.0, 0.0, cols, rows, 64, 48, lo, hi)
        assert np.array_equal(got, want), (rx, ry, m, lo, hi)


def test_angle_checker_second_restatement(oracle):
    rng = np.random.default_rng(2)
    for n in (0, 1, 5, 400, 3000):
        d = np.concatenate([rng.normal(12.0, 4.0, n // 2), rng.uniform(-359.9, 719.0, n - n // 2)]).astype(np.float32)
        d[: n // 10] = (np.round(d[: n // 10] / 15.0) * 15.0).astype(np.float32)   # on bin borders (half-to-even decides)
        assert np.array_equal(nv.angle_checker_invalid(d), oracle.angle_checker_invalid(d)), n
    ties = np.array([10.0] * 4 + [40.0] * 4 + [70.0] * 4 + [100.0] * 4 + [130.0] * 2, np.float32)   # four bins of equal size: the lower three stay
    assert np.array_equal(nv.angle_checker_invalid(ties), oracle.angle_checker_invalid(ties))


def test_projection_matcher_second_restatement(oracle):
    """Rule 18 (projection::match_frame_and_landmarks, the local-map matcher of every tracked frame) from its text over the numpy grid
    and distance matrix: sequential claims, the level window, the stereo gate, the level-aware ratio test -- equal to the C oracle landmark
    for landmark on a frame with clustered descriptors, occupied keypoints, stereo keypoints and invalid landmarks."""
    rng = np.random.default_rng(21)
    cols, rows, n, m = 752, 480, 1500, 900
    kps = np.zeros(n, oracle.KP_DTYPE)
    kps["x"] = rng.uniform(0, cols, n).astype(np.float32)
    kps["y"] = rng.uniform(0, rows, n).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    src = rng.integers(0, n, m)                      # every landmark was seen near some keypoint, with a few flipped bits
    lm_desc = desc[src].copy()
    flips = rng.integers(0, 256, (m, 10))
    for k in range(10):
        lm_desc[np.arange(m), flips[:, k] >> 3] ^= (1 << (flips[:, k] & 7)).astype(np.uint8)
    lm_xy = np.stack([kps["x"][src] + rng.normal(0, 2.0, m), kps["y"][src] + rng.normal(0, 2.0, m)], 1).astype(np.float32)
    lm_level = np.clip(kps["octave"][src] + rng.integers(-1, 2, m), 0, 7).astype(np.int32)
    x_right = np.where(rng.random(n) < 0.4, kps["x"] - rng.uniform(1, 30, n), -1.0).astype(np.float32)
    lm_x_right = (lm_xy[:, 0] - rng.uniform(1, 30, m)).astype(np.float32)
    occupied = (rng.random(n) < 0.1).astype(np.uint8)
    lm_valid = (rng.random(m) < 0.9).astype(np.uint8)
    gp = oracle.grid_params(cols, rows)
    for margin, ratio, xr in ((5.0, 0.6, None), (15.0, 0.9, x_right), (5.0, 0.75, x_right)):
        want, nm = oracle.projection_match_frame_and_landmarks(gp, kps, desc, sf, lm_xy, lm_level, lm_desc, margin, ratio, xr, occupied,
                                                               lm_x_right if xr is not None else None, lm_valid)
        got = nv.projection_match_frame_and_landmarks(kps["x"], kps["y"], kps["octave"], desc, sf, lm_xy, lm_level, lm_desc, cols, rows, margin, ratio,
                                                      xr, occupied, lm_x_right if xr is not None else None, lm_valid)
        assert nm > 200 and np.array_equal(got, want), (margin, ratio, int((got != want).sum()))


@pytest.mark.parametrize("setup,dz", [(0, 0.0), (1, 0.0), (1, 0.4), (1, -0.4)])
def test_current_and_last_frames_matcher_second_restatement(oracle, setup, dz):
    """Rule 21's matcher of every tracked frame (projection::match_current_and_last_frames) from its text: reprojection with the current pose,
    the motion-dependent level window (monocular; stereo rig standing, moving forward, moving backward by more than the baseline), claims,
    the stereo gate, best <= 100, the orientation histogram -- equal to the C oracle landmark for landmark."""
    rng = np.random.default_rng(31 + setup)
    cols, rows, n = 752, 480, 1200
    fx = fy = 420.0
    cx, cy, fxb, base = cols / 2.0, rows / 2.0, 42.0, 0.1
    cam = oracle.Camera(0, setup, fx, fy, cx, cy, fxb, base, cols, rows)
    gp = oracle.grid_params(cols, rows)
    pts = np.stack([rng.uniform(-4, 4, n), rng.uniform(-2.5, 2.5, n), rng.uniform(3, 12, n)], 1)
    T_last = np.concatenate([np.eye(3), np.zeros((3, 1))], 1)
    a = 0.01
    Rc = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    T_curr = np.concatenate([Rc, np.array([[0.02], [-0.01], [-dz]])], 1)   # the current camera centre sits at about z = dz in the last frame
    sf = (1.2 ** np.arange(8)).astype(np.float32)

    def observe(T, noise):
        p = pts @ T[:, :3].T + T[:, 3]
        k = np.zeros(n, oracle.KP_DTYPE)
        k["x"] = (fx * p[:, 0] / p[:, 2] + cx + rng.normal(0, noise, n)).astype(np.float32)
        k["y"] = (fy * p[:, 1] / p[:, 2] + cy + rng.normal(0, noise, n)).astype(np.float32)
        return k, p[:, 2]

    last, _ = observe(T_last, 0.0)
    curr, zc = observe(T_curr, 0.8)
    last["octave"] = rng.integers(0, 8, n)
    curr["octave"] = np.clip(last["octave"] + rng.integers(-2, 3, n), 0, 7)
    last["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    curr["angle"] = np.mod(last["angle"] + rng.normal(8, 3, n) + np.where(rng.random(n) < 0.15, 90, 0), 360).astype(np.float32)
    desc_l = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    desc_c = desc_l.copy()
    flips = rng.integers(0, 256, (n, 12))
    for k in range(12):
        desc_c[np.arange(n), flips[:, k] >> 3] ^= (1 << (flips[:, k] & 7)).astype(np.uint8)
    perm = rng.permutation(n)                      # the current frame lists its keypoints in another order
    curr, desc_c, zc = curr[perm], desc_c[perm], zc[perm]
    x_right = np.where(rng.random(n) < 0.5, curr["x"] - fxb / zc + rng.normal(0, 0.5, n), -1.0).astype(np.float32) if setup else None
    occupied = (rng.random(n) < 0.08).astype(np.uint8)
    last_valid = (rng.random(n) < 0.9).astype(np.uint8)
    for margin, orient in ((7.0, True), (15.0, False)):
        want, nm = oracle.projection_match_current_and_last_frames(cam, gp, curr, desc_c, T_curr, last, pts, desc_l, T_last, sf, margin, orient, x_right,
                                                                   occupied, last_valid)
        got = nv.projection_match_current_and_last_frames(0, setup, (fx, fy, cx, cy, fxb), base, cols, rows, curr["x"], curr["y"], curr["octave"],
                                                          curr["angle"], desc_c, T_curr, last["octave"], last["angle"], pts, desc_l, T_last, sf, margin,
                                                          orient, x_right, occupied, last_valid)
        assert nm > 150 and np.array_equal(got, want), (setup, dz, margin, int((got != want).sum()), nm)


def test_area_matcher_second_restatement(oracle):
    """Rule 19's match::area (the monocular initialiser's matcher): windows around the previously matched points, targets changing owner,
    the histogram that keeps stolen entries, the update of prev_matched_pts -- equal to the C oracle match for match, twice in a row (the second
    call starts from the first call's prev_matched_pts, as the initialiser does frame after frame)."""
    rng = np.random.default_rng(41)
    cols, rows, n = 752, 480, 1400
    gp = oracle.grid_params(cols, rows)
    k1 = np.zeros(n, oracle.KP_DTYPE)
    k1["x"], k1["y"] = rng.uniform(0, cols, n).astype(np.float32), rng.uniform(0, rows, n).astype(np.float32)
    k1["octave"] = np.where(rng.random(n) < 0.7, 0, rng.integers(1, 8, n))
    k1["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    perm = rng.permutation(n)
    k2 = k1[perm].copy()
    k2["x"] = (k2["x"] + rng.normal(3, 2.5, n)).astype(np.float32)
    k2["y"] = (k2["y"] + rng.normal(-2, 2.5, n)).astype(np.float32)
    k2["angle"] = np.mod(k2["angle"] + rng.normal(5, 3, n) + np.where(rng.random(n) < 0.1, 120, 0), 360).astype(np.float32)
    d1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    d1[: n // 6] = d1[rng.integers(0, n, n // 6)]          # look-alikes: several frame-1 keypoints compete for one target
    d2 = d1[perm].copy()
    flips = rng.integers(0, 256, (n, 8))
    for k in range(8):
        d2[np.arange(n), flips[:, k] >> 3] ^= (1 << (flips[:, k] & 7)).astype(np.uint8)
    prev_o = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], 1), np.float32)
    prev_n = prev_o.copy()
    for margin, ratio, orient in ((20, 0.9, True), (12, 0.8, True), (30, 0.95, False)):
        nm, want = oracle.area_match_in_consistent_area(gp, k1, d1, k2, d2, prev_o, margin, ratio, orient)   # (updates prev_o in place)
        gn, got, prev_n = nv.area_match_in_consistent_area(k1["octave"], k1["angle"], d1, k2["x"], k2["y"], k2["octave"], k2["angle"], d2, prev_n, cols,
                                                           rows, margin, ratio, orient)
        assert nm > 100 and gn == nm and np.array_equal(got, want), (margin, ratio, int((got != want).sum()))
        assert np.array_equal(prev_n, prev_o)


def _two_extracted_frames(oracle, shift):
    ox = oracle.OrbExtractor(oracle.make_params(1000))
    ka, da = ox.extract(synth.synth_frame(480, 752, seed=21))
    kb, db = ox.extract(synth.synth_frame(480, 752, seed=21, shift=shift, noise_seed=77))
    return ka, da, kb, db


@pytest.mark.parametrize("ratio,orient", [(0.75, True), (0.6, True), (0.9, False)])
def test_bow_tree_matchers_second_restatement(oracle, ratio, orient):
    """Rule 19's bow_tree walk (common nodes in ascending id, first-come claims inside a node, the ratio test, the orientation histogram) for
    both of its users, on two extracted frames with a synthetic vocabulary assignment and a node missing on one side."""
    ka, da, kb, db = _two_extracted_frames(oracle, (3, 2))
    for n_nodes in (120, 25):      # 25 nodes: long lists, many candidates per keypoint, more contested targets
        fa, fb = synth.synth_bow(da, seed=1, n_nodes=n_nodes), synth.synth_bow(db, seed=1, n_nodes=n_nodes)
        fb.pop(sorted(fb)[3])
        rng = np.random.default_rng(2)
        live_a, live_b = rng.random(len(ka)) < 0.85, rng.random(len(kb)) < 0.8
        wn, want = oracle.bow_match_frame_and_keyframe(ka, da, fa, kb, db, fb, ratio, orient, live_a.astype(np.uint8))
        gn, got = nv.bow_match_frame_and_keyframe(ka["angle"], da, fa, live_a, kb["angle"], db, fb, ratio, orient)
        assert wn > 30 and gn == wn and np.array_equal(got, want)
        wn, want = oracle.bow_match_keyframes(ka, da, fa, kb, db, fb, ratio, orient, live_a.astype(np.uint8), live_b.astype(np.uint8))
        gn, got = nv.bow_match_keyframes(ka["angle"], da, fa, live_a, kb["angle"], db, fb, live_b, ratio, orient)
        assert wn > 20 and gn == wn and np.array_equal(got, want)


def _rot(axis, deg):
    a = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    t = np.deg2rad(deg)
    return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * (K @ K)


@pytest.mark.parametrize("orient,stereo,forward", [(True, False, False), (False, True, False), (True, True, True), (True, False, True)])
def test_triangulation_matcher_second_restatement(oracle, orient, stereo, forward):
    """Rule 23: two keyframes seeing the same 1500 points (15 % moved off their epipolar line, a fifth with look-alike descriptors), keypoints
    with landmarks excluded on both sides, the 3-degree epipole gate (forward motion puts the epipole inside the image) and its stereo
    exemption, the one-sided epipolar residual, 'last of equal distances wins', the orientation histogram."""
    rows, cols, n = 720, 1280, 1500
    rng = np.random.default_rng(10 + forward)
    fx = fy = 0.7 * cols
    cx, cy = cols / 2.0, rows / 2.0
    R2 = _rot((0, 1, 0), 0.5 if forward else 4.0) @ _rot((1, 0, 0), 1.5)
    t2 = np.array([0.02, 0.01, 0.7]) if forward else np.array([-0.6, 0.05, 0.1])
    X = np.stack([rng.uniform(-4, 4, n), rng.uniform(-2.5, 2.5, n), rng.uniform(4, 15, n)], 1)
    P2 = X @ R2.T + t2
    u1 = np.stack([fx * X[:, 0] / X[:, 2] + cx, fy * X[:, 1] / X[:, 2] + cy], 1)
    u2 = np.stack([fx * P2[:, 0] / P2[:, 2] + cx, fy * P2[:, 1] / P2[:, 2] + cy], 1)
    k1, d1 = synth.synth_keypoints(n, rows, cols, seed=31)
    k2 = k1.copy()
    k1["x"], k1["y"] = u1[:, 0], u1[:, 1]
    k2["x"], k2["y"] = u2[:, 0] + rng.normal(0, 0.4, n), u2[:, 1] + rng.normal(0, 0.4, n)
    off = rng.random(n) < 0.15
    k2["y"][off] += rng.uniform(8, 40, int(off.sum()))
    k2["angle"] = (k1["angle"] + np.where(rng.random(n) < 0.8, rng.normal(0, 4, n), rng.uniform(0, 360, n))) % 360
    d2 = np.stack([synth.flip_bits(rng, d1[i], 45) for i in range(n)])
    dup = rng.integers(0, n, n // 5)
    d2[dup] = np.stack([synth.flip_bits(rng, d1[(i + 1) % n], 20) for i in dup])
    perm = rng.permutation(n)
    k2, d2 = k2[perm], d2[perm]

    def bearings(k):
        b = np.stack([(k["x"].astype(np.float64) - cx) / fx, (k["y"].astype(np.float64) - cy) / fy, np.ones(len(k))], 1)
        return b / np.linalg.norm(b, axis=1)[:, None]

    b1, b2 = bearings(k1), bearings(k2)
    R12, t12 = R2.T, -R2.T @ t2                      # x1 = R12 x2 + t12
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    E12 = tx @ R12
    epipole = t2 / np.linalg.norm(t2)                # keyframe 1's centre seen from keyframe 2
    fv1, fv2 = synth.synth_bow(d1, seed=2, n_nodes=60), synth.synth_bow(d2, seed=2, n_nodes=60)
    h1, h2 = (rng.random(n) < 0.3).astype(np.uint8), (rng.random(n) < 0.3).astype(np.uint8)
    x1 = x2 = None
    if stereo:
        x1 = np.where(rng.random(n) < 0.4, k1["x"] - 10, -1.0).astype(np.float32)
        x2 = np.where(rng.random(n) < 0.4, k2["x"] - 10, -1.0).astype(np.float32)
    sf = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)
    wn, want = oracle.robust_match_for_triangulation(k1, d1, fv1, b1, k2, d2, fv2, b2, E12, epipole, sf, orient, h1, h2, x1, x2)
    gn, got = nv.robust_match_for_triangulation(k1["angle"], k1["octave"], d1, fv1, b1, h1, x1, k2["angle"], d2, fv2, b2, h2, x2, E12, epipole, sf, orient)
    assert wn > 100 and gn == wn and np.array_equal(got, want), (wn, gn, int((got != want).sum()))
    if forward and not stereo:       # the epipole gate did remove candidates in this scene
        near = (b2 @ epipole) > 0.99862953475
        assert near.sum() > 5 and not near[want[want >= 0]].any()


def _keyframe_and_landmarks(model, rows, cols, n, seed):
    """A keyframe of n keypoints; landmarks = its keypoints back-projected at random depths (reprojecting onto them up to noise), a tenth
    replaced by points anywhere; per landmark a valid distance range consistent with some level near the keypoint's, and a mean normal."""
    rng = np.random.default_rng(seed)
    ck, cd = synth.synth_keypoints(n, rows, cols, seed=seed)
    R = _rot((0, 1, 0), 2.0) @ _rot((1, 0, 0), -1.0)
    t = np.array([0.05, -0.02, 0.3])
    T = np.concatenate([R, t[:, None]], 1)
    fx = fy = 0.6 * cols
    cx, cy = cols / 2.0, rows / 2.0
    depth = rng.uniform(2.0, 20.0, n)
    u, v = ck["x"].astype(float) + rng.normal(0, 1.5, n), ck["y"].astype(float) + rng.normal(0, 1.5, n)
    if model == 0:
        pc = np.stack([(u - cx) / fx * depth, (v - cy) / fy * depth, depth], 1)
    else:
        lon, lat = (u / cols - 0.5) * 2 * np.pi, -(v / rows - 0.5) * np.pi
        pc = np.stack([np.cos(lat) * np.sin(lon), -np.sin(lat), np.cos(lat) * np.cos(lon)], 1) * depth[:, None]
    pw = (pc - t) @ R
    m = int(1.2 * n)
    src = np.concatenate([rng.permutation(n), rng.integers(0, n, m - n)])
    lk = ck[src].copy()
    lk["angle"] = (lk["angle"] + np.where(rng.random(m) < 0.8, rng.normal(0, 5, m), rng.uniform(0, 360, m))) % 360
    lpw = pw[src] + rng.normal(0, 0.002, (m, 3))
    ld = np.stack([synth.flip_bits(rng, cd[i], 60) for i in src])
    far = rng.random(m) < 0.1
    lpw[far] = rng.uniform(-30, 30, (int(far.sum()), 3))
    valid = (rng.random(m) < 0.9).astype(np.uint8)
    ray = lpw - (-R.T @ t)
    dist = np.linalg.norm(ray, axis=1)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    lvl = np.clip(lk["octave"] + rng.integers(-1, 2, m), 0, 7)
    dmax = (dist * sf[lvl] * rng.uniform(0.85, 1.0, m)).astype(np.float32)
    dmin = (dmax / sf[7] * rng.uniform(0.5, 1.3, m)).astype(np.float32)
    nrm = ray / dist[:, None]
    flip = rng.random(m) < 0.15
    nrm[flip] = rng.normal(0, 1, (int(flip.sum()), 3))
    return ck, cd, T, lk, lpw, ld, valid, np.ascontiguousarray(np.stack([dmin, dmax], 1)), nrm, sf, (fx, fy, cx, cy)


@pytest.mark.parametrize("model", [0, 1])
def test_frame_and_keyframe_projection_matcher_second_restatement(oracle, model):
    """Rule 21's relocalisation matcher: the landmark's distance range through the float getters, predict_scale_level in float, the level window
    around the prediction, sequential claims, the caller's Hamming threshold."""
    rows, cols, n = (480, 960, 1500) if model == 1 else (480, 752, 1200)
    ck, cd, T, lk, lpw, ld, valid, dmm, _, sf, (fx, fy, cx, cy) = _keyframe_and_landmarks(model, rows, cols, n, 60 + model)
    occ = (np.random.default_rng(17).random(n) < 0.1).astype(np.uint8)
    cam = oracle.Camera(model, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    gp = oracle.grid_params(cols, rows)
    lsf = float(np.log(np.float32(1.2)))
    camt = (fx, fy, cx, cy, 0.0) if model == 0 else (cols, rows)
    for margin, thr, orient in ((10.0, 100, True), (20.0, 50, False)):
        want, wn = oracle.projection_match_frame_and_keyframe(cam, gp, ck, cd, T, lk, lpw, dmm, ld, sf, lsf, margin, thr, orient, curr_occupied=occ,
                                                              kf_valid=valid)
        got = nv.projection_match_frame_and_keyframe(model, camt, cols, rows, ck["x"], ck["y"], ck["octave"], ck["angle"], cd, T, lk["angle"], lpw, dmm, ld,
                                                     sf, lsf, margin, thr, orient, occ, valid)
        assert wn > n // 20 and np.array_equal(got, want), (model, margin, int((got != want).sum()), wn)


@pytest.mark.parametrize("model,setup", [(0, 0), (0, 1), (1, 0)])
def test_fuse_candidate_search_second_restatement(oracle, model, setup):
    """Rule 22: viewing-angle gate, every grid candidate of the predicted radius, levels [pred - 1, pred], the chi-square gate on the
    reprojection error (three components for a stereo keypoint), best Hamming <= 50, no claims."""
    rows, cols, n = (480, 960, 1500) if model == 1 else (480, 752, 1200)
    ck, cd, T, lk, lpw, ld, valid, dmm, nrm, sf, (fx, fy, cx, cy) = _keyframe_and_landmarks(model, rows, cols, n, 40 + model + setup)
    rng = np.random.default_rng(7)
    ils = (1.0 / (sf * sf)).astype(np.float32)
    cam = oracle.Camera(model, setup, fx, fy, cx, cy, 0.12 * fx, 0.12, cols, rows)
    gp = oracle.grid_params(cols, rows)
    lsf = float(np.log(np.float32(1.2)))
    xr = np.where(rng.random(n) < 0.6, ck["x"] - 0.12 * fx / rng.uniform(2, 20, n), -1.0).astype(np.float32) if setup else None
    camt = (fx, fy, cx, cy, 0.12 * fx) if model == 0 else (cols, rows)
    for margin in (3.0, 8.0):
        want, wn = oracle.fuse_replace_duplication(cam, gp, ck, cd, T, lpw, dmm, nrm, ld, sf, ils, lsf, margin, kf_stereo_x_right=xr, lm_valid=valid)
        got = nv.fuse_replace_duplication(model, camt, cols, rows, ck["x"], ck["y"], ck["octave"], cd, T, lpw, dmm, nrm, ld, sf, ils, lsf, margin, xr, valid)
        assert wn > n // 40 and np.array_equal(got, want), (model, setup, margin, int((got != want).sum()), wn, int((got >= 0).sum()))


@pytest.mark.parametrize("rows,cols,nfeat,seed", [(240, 400, 500, 3), (376, 620, 1000, 1)])
def test_stereo_matcher_second_restatement(oracle, rows, cols, nfeat, seed):
    """Rule 20: row bands, the octave and disparity gates, Hamming < 75, the 11 x 11 centred-window L1 search on the keypoint's pyramid level, the float
    parabola, the disparity range and the median outlier rule -- stereo_x_right and depth equal to the C oracle bit for bit, for the default
    factor 2.0 and the 2.1 variant."""
    left, right, _ = synth.synth_stereo_pair(rows, cols, seed=seed)
    oxl, oxr = oracle.OrbExtractor(oracle.make_params(nfeat)), oracle.OrbExtractor(oracle.make_params(nfeat))
    kl, dl = oxl.extract(left)
    kr, dr = oxr.extract(right)
    tabs = oracle.orb_tables(oxl.params)
    n_levels = oxl.params.num_levels
    pyr_l, pyr_r = [oxl.level_image(l) for l in range(n_levels)], [oxr.level_image(l) for l in range(n_levels)]
    for f21 in (False, True):
        wx, wd, wn = oracle.stereo_compute(oxl, oxr, kl, dl, kr, dr, 386.1448, 0.5372, outlier_factor_21=f21)
        gx, gd = nv.stereo_compute(pyr_l, pyr_r, kl, dl, kr, dr, tabs["scale_factors"], tabs["inv_scale_factors"], 386.1448, 0.5372, 2.1 if f21 else 2.0)
        assert wn > nfeat // 10 and np.array_equal(gx >= 0, wx >= 0), (int(((gx >= 0) != (wx >= 0)).sum()), wn)
        assert np.array_equal(gx.view(np.uint32), wx.view(np.uint32)) and np.array_equal(gd.view(np.uint32), wd.view(np.uint32))


@pytest.mark.parametrize("model", [0, 1])
def test_sim3_matchers_second_restatement(oracle, model):
    """Rule 27: the pose recovered from [sR | st], then fuse::detect_duplication (no chi-square gate, no claims) and
    projection::match_by_Sim3_transform (claims in landmark order, pre-occupied keypoints; a third of the landmarks duplicated so that claims decide)."""
    rows, cols, n = (480, 960, 1500) if model == 1 else (480, 752, 1200)
    ck, cd, T, lk, lpw, ld, valid, dmm, nrm, sf, (fx, fy, cx, cy) = _keyframe_and_landmarks(model, rows, cols, n, 80 + model)
    cam = oracle.Camera(model, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    gp = oracle.grid_params(cols, rows)
    lsf = float(np.log(np.float32(1.2)))
    camt = (fx, fy, cx, cy, 0.0) if model == 0 else (cols, rows)
    rng = np.random.default_rng(5)
    extra = rng.integers(0, len(lpw), len(lpw) // 3)
    lpw2 = np.concatenate([lpw, lpw[extra] + rng.normal(0, 0.002, (len(extra), 3))])
    dmm2, nrm2, valid2 = np.concatenate([dmm, dmm[extra]]), np.concatenate([nrm, nrm[extra]]), np.concatenate([valid, valid[extra]])
    ld2 = np.concatenate([ld, np.stack([synth.flip_bits(rng, ld[i], 6) for i in extra])])
    occ = (rng.random(n) < 0.1).astype(np.uint8)
    for scale, margin in ((1.7, 4.0), (0.6, 10.0)):
        S = np.concatenate([scale * T[:, :3], scale * T[:, 3:]], 1)
        want, wn = oracle.fuse_detect_duplication(cam, gp, ck, cd, S, lpw, dmm, nrm, ld, sf, lsf, margin, lm_valid=valid)
        got = nv.fuse_detect_duplication(model, camt, cols, rows, ck["x"], ck["y"], ck["octave"], cd, S, lpw, dmm, nrm, ld, sf, lsf, margin, valid)
        assert wn > n // 20 and np.array_equal(got, want), ("detect", model, scale, int((got != want).sum()))
        want, wn = oracle.projection_match_by_sim3_transform(cam, gp, ck, cd, S, lpw2, dmm2, nrm2, ld2, sf, lsf, margin, kf_occupied=occ, lm_valid=valid2)
        got = nv.projection_match_by_sim3_transform(model, camt, cols, rows, ck["x"], ck["y"], ck["octave"], cd, S, lpw2, dmm2, nrm2, ld2, sf, lsf, margin, occ,
                                                    valid2)
        assert wn > n // 20 and np.array_equal(got, want), ("by_sim3", model, scale, int((got != want).sum()))


@pytest.mark.parametrize("s_12", [1.0, 1.35])
def test_mutual_sim3_matcher_second_restatement(oracle, s_12):
    """Rule 27's projection::match_keyframes_mutually: two keyframes of the same 1500 points, keyframe 2's map s_12 times smaller; both directions
    through Sim3_12 and its inverse, agreement of the two picks."""
    rows, cols, n = 720, 1280, 1500
    rng = np.random.default_rng(22)
    fx = fy = 0.7 * cols
    cx, cy = cols / 2.0, rows / 2.0
    R1, t1 = np.eye(3), np.zeros(3)
    R2, t2 = _rot((0, 1, 0), 4.0) @ _rot((1, 0, 0), 1.5), np.array([-0.6, 0.05, 0.1])
    X = np.stack([rng.uniform(-4, 4, n), rng.uniform(-2.5, 2.5, n), rng.uniform(4, 15, n)], 1)
    p1, p2 = X @ R1.T + t1, X @ R2.T + t2
    k1, d1 = synth.synth_keypoints(n, rows, cols, seed=33)
    k2 = k1.copy()
    k1["x"], k1["y"] = fx * p1[:, 0] / p1[:, 2] + cx + rng.normal(0, 0.5, n), fy * p1[:, 1] / p1[:, 2] + cy + rng.normal(0, 0.5, n)
    k2["x"], k2["y"] = fx * p2[:, 0] / p2[:, 2] + cx + rng.normal(0, 0.5, n), fy * p2[:, 1] / p2[:, 2] + cy + rng.normal(0, 0.5, n)
    k2["octave"] = np.clip(k1["octave"] + rng.integers(-1, 2, n), 0, 7)
    d2 = np.stack([synth.flip_bits(rng, d1[i], 40) for i in range(n)])
    dup = rng.integers(0, n, n // 6)
    d2[dup] = np.stack([synth.flip_bits(rng, d1[(i + 7) % n], 25) for i in dup])
    perm = rng.permutation(n)
    k2, d2, X2, p2 = k2[perm], d2[perm], X[perm] / s_12, p2[perm]
    T1 = np.concatenate([R1, t1[:, None]], 1)
    T2 = np.concatenate([R2, (t2 / s_12)[:, None]], 1)
    R12, t12 = R1 @ R2.T, t1 - R1 @ R2.T @ t2
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    inv = np.argsort(perm)

    def ranges(dist, octave):
        lvl = np.clip(octave + rng.integers(0, 2, n), 0, 7)
        dmax = (dist * sf[lvl] * rng.uniform(0.85, 1.0, n)).astype(np.float32)
        return np.ascontiguousarray(np.stack([(dmax / sf[7] * rng.uniform(0.5, 1.3, n)).astype(np.float32), dmax], 1))

    dm1 = ranges(np.linalg.norm(p2[inv], axis=1) / s_12, k2["octave"][inv])
    dm2 = ranges(np.linalg.norm(p1[perm], axis=1), k1["octave"][perm])
    l1 = np.stack([synth.flip_bits(rng, d1[i], 10) for i in range(n)])
    l2 = np.stack([synth.flip_bits(rng, d2[i], 10) for i in range(n)])
    v1, v2 = (rng.random(n) < 0.85).astype(np.uint8), (rng.random(n) < 0.85).astype(np.uint8)
    cam = oracle.Camera(0, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    gp = oracle.grid_params(cols, rows)
    lsf = float(np.log(np.float32(1.2)))
    for margin in (7.5, 15.0):
        wn, want = oracle.projection_match_keyframes_mutually(cam, gp, k1, d1, T1, X, dm1, l1, v1, k2, d2, T2, X2, dm2, l2, v2, s_12, R12, t12, sf, lsf, margin)
        gn, got = nv.projection_match_keyframes_mutually((fx, fy, cx, cy), cols, rows, k1, d1, T1, X, dm1, l1, v1, k2, d2, T2, X2, dm2, l2, v2, s_12, R12, t12,
                                                         sf, lsf, margin)
        assert wn > n // 5 and gn == wn and np.array_equal(got, want), (s_12, margin, wn, gn, int((got != want).sum()))


def test_whole_extractor_in_numpy_reproduces_the_golden_vectors(oracle):
    """orb_extractor::extract as one numpy pipeline (tests/nversion_extract.py: tables, pyramid, cell loop + FAST, closed-form quad-tree,
    orientation, blur, rBRIEF, scaling) with no oracle C code in it: the committed golden files come out of it byte for byte -- all seven
    keypoint fields and the descriptors of 1008 + 1002 + the small frame's keypoints, and the per-level candidate counts."""
    import nversion_extract as nx
    pat = oracle.orb_pattern()          # (the 256 x 4 table is data, shared with the product through orb_pattern.inc)
    g = np.load(os.path.join(GOLDEN, "orb_752x480_seed0.npz"))
    for key, img in (("a", synth.synth_frame(480, 752, seed=0)), ("b", synth.synth_frame(480, 752, seed=0, shift=(5, 0), noise_seed=4242))):
        k, d, n_cand = nx.extract(img, pat, 1000)
        want = g["kps_" + key]
        assert len(k) == len(want) and k.tobytes() == np.ascontiguousarray(want).tobytes() and np.array_equal(d, g["desc_" + key]), key
        if key == "a":
            assert np.array_equal(n_cand, g["n_cand"])
    g2 = np.load(os.path.join(GOLDEN, "orb_331x203_seed5.npz"))
    k, d, _ = nx.extract(synth.synth_frame(203, 331, seed=5), pat, 300)
    assert k.tobytes() == np.ascontiguousarray(g2["kps"]).tobytes() and np.array_equal(d, g2["desc"])


@pytest.mark.parametrize("rows,cols,nfeat,scale,levels,seed", [(480, 640, 2000, 1.2, 8, 2), (500, 300, 700, 1.2, 8, 4), (97, 131, 200, 1.2, 8, 3),
                                                              (360, 480, 1500, 1.1, 12, 6), (240, 320, 400, 1.5, 4, 7), (45, 60, 50, 1.2, 8, 8)])
def test_whole_extractor_in_numpy_equals_the_oracle(oracle, rows, cols, nfeat, scale, levels, seed):
    """The same pipeline against the C oracle beyond the golden inputs: portrait and tiny frames (levels that end up without a cell), other scale
    factors / level counts / budgets (over- and under-subscribed levels), also with a low-contrast frame that makes cells fall back to min_fast_thr."""
    import nversion_extract as nx
    pat = oracle.orb_pattern()
    img = synth.synth_frame(rows, cols, seed=seed)
    flat = (100 + (img.astype(np.int32) - 128) // 6).astype(np.uint8)         # contrast / 6: most cells need the low threshold
    for im in (img, flat):
        p = oracle.make_params(nfeat)
        p.scale_factor, p.num_levels = scale, levels
        wk, wd = oracle.OrbExtractor(p).extract(im)
        k, d, _ = nx.extract(im, pat, nfeat, scale, levels)
        assert len(k) == len(wk) and k.tobytes() == np.ascontiguousarray(wk).tobytes() and np.array_equal(d, wd), (len(k), len(wk))


@pytest.mark.parametrize("k,depth", [(10, 4), (6, 3), (3, 6)])
def test_bow_transform_second_restatement(oracle, k, depth):
    """Rule 29 (DBoW2 transform): the level-synchronous numpy descent (bit-vector distances, first minimum in child order, ragged last level,
    zero-weight words) equals the oracle's per-feature loop for every levelsup, on descriptors near the words and on random ones."""
    vocab = synth.synth_vocabulary(k, depth, seed=k + depth)
    rng = np.random.default_rng(9)
    leaves = np.nonzero(vocab["word_id"] >= 0)[0]
    near = np.stack([synth.flip_bits(rng, vocab["desc"][i], 30) for i in rng.choice(leaves, 600)])
    desc = np.concatenate([near, rng.integers(0, 256, (200, 32), dtype=np.uint8), vocab["desc"][leaves[:50]]])
    for levelsup in range(0, depth + 2):
        w, wt, nd = oracle.bow_transform(vocab, desc, levelsup)
        gw, gwt, gnd = nv.bow_transform(vocab, desc, levelsup)
        assert np.array_equal(gw, w) and np.array_equal(gwt, wt) and np.array_equal(gnd, nd), (k, depth, levelsup)
    assert (w >= 0).all() and len(np.unique(w)) > 100 or k == 3


def _small_lba_scene(seed, stereo_frac, outlier_frac=0.04):
    from openvslam_amd.ba import EDGE_STEREO_DTYPE, quat_to_rot
    d = synth.synth_local_ba(n_pose=7, n_pt=260, obs_per_pose=150, seed=seed, pose_noise=0.03, point_noise=0.03, n_fixed=2)
    rng = np.random.default_rng(seed + 100)
    e = d["edges"].copy()
    bad = rng.random(len(e)) < outlier_frac
    e["obs_x"][bad] += rng.choice([-1, 1], int(bad.sum())) * rng.uniform(15, 60, int(bad.sum()))
    bf = 0.12 * d["cam"][0]
    is_st = rng.random(len(e)) < stereo_frac
    st = np.zeros(int(is_st.sum()), EDGE_STEREO_DTYPE)
    if len(st):
        es = e[is_st]
        for k in ("pose_idx", "point_idx", "obs_x", "obs_y", "inv_sigma_sq"):
            st[k] = es[k]
        z = np.empty(len(es))
        for p in np.unique(es["pose_idx"]):
            sel = es["pose_idx"] == p
            z[sel] = (d["points_true"][es["point_idx"][sel]] @ quat_to_rot(d["poses_true"][p, 3:]).T + d["poses_true"][p, :3])[:, 2]
        st["obs_x_right"] = es["obs_x"] - bf / z + rng.normal(0, 1, len(es))
    return d, np.ascontiguousarray(e[~is_st]), st, bf


@pytest.mark.parametrize("seed,stereo_frac", [(1, 0.0), (2, 0.35), (3, 1.0)])
def test_local_ba_second_restatement(oracle, seed, stereo_frac):
    """Rule 28 (local_bundle_adjuster::optimize behind the graph build) a second time, in another FORM: tests/nversion_pose.py keeps the landmarks
    in the system -- every Levenberg-Marquardt trial is one LAPACK solve over (free keyframes, landmarks) -- where the oracle and the library
    eliminate them; blocks from the numpy linearisation. Same iteration counts, same outlier flags, chi2 to 1e-9, states to 1e-7 (the bound the
    library's own solver is held to)."""
    import nversion_pose as npz
    from oracle import lba
    d, mono, st, bf = _small_lba_scene(seed, stereo_frac)
    if stereo_frac == 1.0:
        mono = mono[:0]
    want = lba.local_ba_optimize(d["poses"], d["pose_fixed"], d["points"], mono, d["cam"], st if len(st) else None, bf if len(st) else 0.0)
    got = npz.local_ba_optimize(d["poses"], d["pose_fixed"], d["points"], mono, d["cam"], st if len(st) else None, bf if len(st) else 0.0)
    assert np.array_equal(got["info"][4:], want["info"][4:]) and want["info"][4] >= 3, (got["info"], want["info"])
    assert np.allclose(got["info"][:4], want["info"][:4], rtol=1e-9)
    assert np.array_equal(got["mono_outlier"], want["mono_outlier"]) and np.array_equal(got["stereo_outlier"], want["stereo_outlier"])
    assert want["mono_outlier"].sum() + want["stereo_outlier"].sum() > 10
    Rw = npz._quat_rot(want["poses"][:, 3:])
    seen = np.zeros(len(d["points"]), bool)
    seen[mono["point_idx"]] = True
    seen[st["point_idx"]] = True
    assert np.abs(got["R"] - Rw).max() < 1e-7 and np.abs(got["t"] - want["poses"][:, :3]).max() < 1e-7
    assert np.abs(got["points"][seen] - want["points"][seen]).max() < 1e-7 and np.array_equal(got["points"][~seen], want["points"][~seen])
    print("local BA, two forms: pose", np.abs(got["t"] - want["poses"][:, :3]).max(), "points", np.abs(got["points"][seen] - want["points"][seen]).max())


def test_local_ba_equirect_second_restatement(oracle):
    """The same two forms over the equirectangular edge (rule 26): six keyframes inside a shell of 300 landmarks, exact observations plus noise,
    3 % displaced by 60 px."""
    import nversion_pose as npz
    from oracle import lba
    rng = np.random.default_rng(2)
    n_pose, n_pt, per, cols, rows = 6, 300, 160, 3840, 1920
    pts = rng.normal(size=(n_pt, 3))
    pts *= (rng.uniform(3.0, 9.0, n_pt) / np.linalg.norm(pts, axis=1))[:, None]
    poses = np.zeros((n_pose, 7))
    poses[:, 6] = 1.0
    poses[:, :3] = rng.normal(0, 0.5, (n_pose, 3))
    edges = np.zeros(n_pose * per, oracle.BA_EDGE_DTYPE)
    for i in range(n_pose):
        sel = rng.choice(n_pt, per, replace=False)
        u, v = synth.equirect_project(pts[sel] + poses[i, :3], cols, rows)
        e = edges[i * per:(i + 1) * per]
        e["pose_idx"], e["point_idx"], e["obs_x"], e["obs_y"], e["inv_sigma_sq"] = i, sel, u + rng.normal(0, 0.7, per), v + rng.normal(0, 0.7, per), 1.0
    edges["obs_y"][rng.random(len(edges)) < 0.03] += 60.0
    fixed = np.zeros(n_pose, np.uint8)
    fixed[:2] = 1
    p0, x0 = poses.copy(), pts + rng.normal(0, 0.01, pts.shape)
    p0[2:, :3] += rng.normal(0, 0.01, (n_pose - 2, 3))
    want = lba.local_ba_optimize_equirect(p0, fixed, x0, edges, cols, rows)
    got = npz.local_ba_optimize(p0, fixed, x0, edges, (float(cols), float(rows), 0.0, 0.0), None, 0.0, setup_type=0, equirect=True)
    assert np.array_equal(got["info"][4:], want["info"][4:]) and np.allclose(got["info"][:4], want["info"][:4], rtol=1e-9)
    assert np.array_equal(got["mono_outlier"], want["mono_outlier"]) and want["mono_outlier"].sum() > 10
    seen = np.zeros(n_pt, bool)
    seen[edges["point_idx"]] = True
    assert np.abs(got["R"] - npz._quat_rot(want["poses"][:, 3:])).max() < 1e-7 and np.abs(got["t"] - want["poses"][:, :3]).max() < 1e-7
    assert np.abs(got["points"][seen] - want["points"][seen]).max() < 1e-7


@pytest.mark.parametrize("factor,tie,taps", [(1, 0, 0), (3, 1, 0), (3, 0, 1), (1, 1, 1)])
def test_whole_extractor_in_numpy_follows_the_variant_switches(oracle, factor, tie, taps):
    """The run-time variants of rules 6, 7 and 10 (quad-tree switch factor 3 | 1, order of equal counts, blur taps) have a second form too: the
    numpy pipeline with the same three switches equals the oracle under ovo_orb_set_variant, and each switch does change the output."""
    import nversion_extract as nx
    pat = oracle.orb_pattern()
    img = synth.synth_frame(480, 752, seed=12)
    ox = oracle.OrbExtractor(oracle.make_params(1000))
    k0, d0 = ox.extract(img)
    ox.set_variant("tree_switch_factor", factor)
    ox.set_variant("tree_tie_order", tie)
    ox.set_variant("blur_taps", taps)
    wk, wd = ox.extract(img)
    k, d, _ = nx.extract(img, pat, 1000, tree_switch_factor=factor, tree_tie_order=tie, blur_taps=taps)
    assert len(k) == len(wk) and k.tobytes() == np.ascontiguousarray(wk).tobytes() and np.array_equal(d, wd)
    assert len(wk) != len(k0) or wk.tobytes() != np.ascontiguousarray(k0).tobytes() or not np.array_equal(wd, d0)


def test_remaining_variants_second_restatement(oracle):
    """The round-4 switches in their second form: rule 20's double parabola (stereo_x_right and depth bit for bit), rule 25 (iv)'s per-round
    vertex re-set (flags identical, pose to 2e-8), rule 17's ORB-SLAM2 keep rule (bin by bin on histograms with a dominant mode)."""
    import nversion_pose as nvp
    from openvslam_amd.synth import synth_pose_frame
    # stereo: parabola in double, with both outlier factors
    left, right, _ = synth.synth_stereo_pair(240, 400, seed=3)
    oxl, oxr = oracle.OrbExtractor(oracle.make_params(500)), oracle.OrbExtractor(oracle.make_params(500))
    kl, dl = oxl.extract(left)
    kr, dr = oxr.extract(right)
    tabs = oracle.orb_tables(oxl.params)
    pyr_l, pyr_r = [oxl.level_image(l) for l in range(8)], [oxr.level_image(l) for l in range(8)]
    for f21 in (False, True):
        wx, wd, wn = oracle.stereo_compute(oxl, oxr, kl, dl, kr, dr, 386.1448, 0.5372, outlier_factor_21=f21, parabola_double=True)
        gx, gd = nv.stereo_compute(pyr_l, pyr_r, kl, dl, kr, dr, tabs["scale_factors"], tabs["inv_scale_factors"], 386.1448, 0.5372, 2.1 if f21 else 2.0, True)
        assert wn > 50 and np.array_equal(gx.view(np.uint32), wx.view(np.uint32)) and np.array_equal(gd.view(np.uint32), wd.view(np.uint32))
    # (on these frames the double quotient rounds to the float one everywhere or moves it by an ulp: equality with the oracle is the point)
    # pose optimiser: the frame vertex re-set every round
    oracle.pose_set_variant("reset_each_round", 1)
    try:
        for n, sf, of, pe, seed in ((1500, 0.4, 0.1, 1.0, 1), (300, 0.0, 0.2, 2.0, 2)):
            T0, obs, cam, bf, _ = synth_pose_frame(oracle.POSE_OBS_DTYPE, n, seed, sf, of, pe)
            wT, wout, wnv = oracle.pose_optimize(T0, obs, cam, bf)
            T, out, nval = nvp.pose_optimize(T0, obs, cam, bf, reset_each_round=True)
            assert nval == wnv and np.array_equal(out, wout.astype(bool)) and np.allclose(T, wT, rtol=0, atol=2e-8)
    finally:
        oracle.pose_set_variant("reset_each_round", 0)
    # angle checker: ORB-SLAM2's keep rule
    rng = np.random.default_rng(4)
    oracle.match_set_variant("angle_keep_rule", 1)
    try:
        changed = 0
        for trial in range(40):
            n = int(rng.integers(20, 400))
            d = np.where(rng.random(n) < rng.uniform(0.6, 0.98), rng.normal(rng.uniform(0, 360), 6, n), rng.uniform(-360, 720, n)).astype(np.float32)
            want = oracle.angle_checker_invalid(d)
            assert np.array_equal(nv.angle_checker_invalid(d, 1), want)
            changed += int(not np.array_equal(nv.angle_checker_invalid(d, 0), want))
        assert changed > 5
    finally:
        oracle.match_set_variant("angle_keep_rule", 0)


def test_small_tie_heavy_cases_of_every_list_matcher(oracle):
    """600 small random cases (0 .. 40 keypoints, descriptors drawn from THREE base patterns with a few flipped bits, positions on a coarse lattice,
    angles from a handful of values): equal Hamming distances, equal histogram counts, candidates exactly on window edges and empty inputs
    everywhere -- the places where only the tie rules decide. Second restatements against the oracle: brute force (rule 14), area (19),
    both bow_tree matchers (19), frame-and-landmarks (18)."""
    rng = np.random.default_rng(2024)
    cols, rows = 200, 120
    gp = oracle.grid_params(cols, rows)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    base = rng.integers(0, 256, (3, 32), dtype=np.uint8)

    def frame(n):
        k = np.zeros(n, oracle.KP_DTYPE)
        k["x"] = (rng.integers(0, 41, n) * 5).astype(np.float32)          # lattice of 5 px: |dx| == margin happens
        k["y"] = (rng.integers(0, 25, n) * 5).astype(np.float32)
        k["octave"] = rng.integers(0, 3, n)
        k["angle"] = rng.choice(np.array([0, 15, 45, 100, 200, 355], np.float32), n)
        d = base[rng.integers(0, 3, n)].copy()
        for j in range(n):
            for b in rng.integers(0, 256, rng.integers(0, 4)):
                d[j, b >> 3] ^= np.uint8(1 << (b & 7))
        return k, d

    def bow(n):
        fv = {}
        for i, node in enumerate(rng.integers(0, 4, n)):
            fv.setdefault(int(node), []).append(i)
        return fv

    for case in range(600):
        n1, n2 = int(rng.integers(0, 41)), int(rng.integers(0, 41))
        k1, d1 = frame(n1)
        k2, d2 = frame(n2)
        ratio = float(rng.choice([0.6, 0.9, 1.0]))
        orient = bool(rng.integers(0, 2))
        if n1 and n2:
            v = (rng.random(n2) < 0.8).astype(np.uint8)
            assert np.array_equal(nv.robust_brute_force_match(d1, d2, v, ratio), oracle.robust_brute_force_match(d1, d2, v, ratio)), ("bf", case)
        margin = int(rng.choice([5, 10, 20]))
        prev_o = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], 1), np.float32).reshape(-1, 2)
        prev_n = prev_o.copy()
        wn, want = oracle.area_match_in_consistent_area(gp, k1, d1, k2, d2, prev_o, margin, ratio, orient)
        gn, got, prev_n = nv.area_match_in_consistent_area(k1["octave"], k1["angle"], d1, k2["x"], k2["y"], k2["octave"], k2["angle"], d2, prev_n, cols, rows,
                                                           margin, ratio, orient)
        assert gn == wn and np.array_equal(got, want) and np.array_equal(prev_n, prev_o), ("area", case)
        f1, f2 = bow(n1), bow(n2)
        l1, l2 = rng.random(n1) < 0.8, rng.random(n2) < 0.8
        wn, want = oracle.bow_match_frame_and_keyframe(k1, d1, f1, k2, d2, f2, ratio, orient, l1.astype(np.uint8))
        gn, got = nv.bow_match_frame_and_keyframe(k1["angle"], d1, f1, l1, k2["angle"], d2, f2, ratio, orient)
        assert gn == wn and np.array_equal(got, want), ("bow frame", case)
        wn, want = oracle.bow_match_keyframes(k1, d1, f1, k2, d2, f2, ratio, orient, l1.astype(np.uint8), l2.astype(np.uint8))
        gn, got = nv.bow_match_keyframes(k1["angle"], d1, f1, l1, k2["angle"], d2, f2, l2, ratio, orient)
        assert gn == wn and np.array_equal(got, want), ("bow keyframes", case)
        m = int(rng.integers(0, 30))
        lk, ld = frame(m)
        lm_xy = np.ascontiguousarray(np.stack([lk["x"], lk["y"]], 1), np.float32).reshape(-1, 2)
        lvl = lk["octave"].astype(np.int32)
        occ = (rng.random(n1) < 0.1).astype(np.uint8)
        want, nm = oracle.projection_match_frame_and_landmarks(gp, k1, d1, sf, lm_xy, lvl, ld, float(margin), ratio, None, occ, None, None)
        got = nv.projection_match_frame_and_landmarks(k1["x"], k1["y"], k1["octave"], d1, sf, lm_xy, lvl, ld, cols, rows, float(margin), ratio, None, occ, None, None)
        assert np.array_equal(got, want), ("frame and landmarks", case)


def test_triangulation_matcher_ties(oracle):
    """Rule 23 where only its tie rule decides: every point lies in ONE epipolar plane (y = 0 with a baseline along x), so every pairing passes
    check_epipolar_constraint, and the descriptors come from three base patterns -- many candidates at EQUAL distance, of which the LAST in
    node order must win ('d <= best'), and a target taken earlier is gone for later keypoints."""
    rng = np.random.default_rng(77)
    fx = fy = 400.0
    cx, cy = 320.0, 240.0
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    base = rng.integers(0, 256, (3, 32), dtype=np.uint8)
    R2, t2 = np.eye(3), np.array([-0.5, 0.0, 0.0])
    E12 = np.array([[0, 0, 0], [0, 0, -0.5], [0, 0.5, 0]]) * -1.0          # [t12]x R12 with t12 = (0.5, 0, 0), R12 = I
    ep = t2 / np.linalg.norm(t2)
    n_ties = 0
    for case in range(200):
        n = int(rng.integers(1, 30))
        X = np.stack([rng.uniform(-3, 3, n), np.zeros(n), rng.uniform(3, 9, n)], 1)
        P2 = X + t2
        k1, k2 = np.zeros(n, oracle.KP_DTYPE), np.zeros(n, oracle.KP_DTYPE)
        k1["x"], k1["y"] = fx * X[:, 0] / X[:, 2] + cx, cy
        k2["x"], k2["y"] = fx * P2[:, 0] / P2[:, 2] + cx, cy
        k1["octave"], k2["octave"] = rng.integers(0, 4, n), rng.integers(0, 4, n)
        k1["angle"] = rng.choice(np.array([0, 20, 100, 250], np.float32), n)
        k2["angle"] = rng.choice(np.array([0, 20, 100, 250], np.float32), n)
        d1, d2 = base[rng.integers(0, 3, n)].copy(), base[rng.integers(0, 3, n)].copy()
        for d in (d1, d2):
            for j in range(n):
                for b in rng.integers(0, 256, rng.integers(0, 3)):
                    d[j, b >> 3] ^= np.uint8(1 << (b & 7))
        perm = rng.permutation(n)
        k2, d2 = k2[perm], d2[perm]

        def bearings(k):
            b = np.stack([(k["x"].astype(np.float64) - cx) / fx, (k["y"].astype(np.float64) - cy) / fy, np.ones(len(k))], 1)
            return b / np.linalg.norm(b, axis=1)[:, None]

        b1, b2 = bearings(k1), bearings(k2)
        fv1, fv2 = {}, {}
        for i, node in enumerate(rng.integers(0, 2, n)):
            fv1.setdefault(int(node), []).append(i)
        for i, node in enumerate(rng.integers(0, 2, n)):
            fv2.setdefault(int(node), []).append(i)
        h1, h2 = (rng.random(n) < 0.2).astype(np.uint8), (rng.random(n) < 0.2).astype(np.uint8)
        orient = bool(rng.integers(0, 2))
        wn, want = oracle.robust_match_for_triangulation(k1, d1, fv1, b1, k2, d2, fv2, b2, E12, ep, sf, orient, h1, h2, None, None)
        gn, got = nv.robust_match_for_triangulation(k1["angle"], k1["octave"], d1, fv1, b1, h1, None, k2["angle"], d2, fv2, b2, h2, None, E12, ep, sf, orient)
        assert gn == wn and np.array_equal(got, want), (case, n, got, want)
        D = nv.hamming_matrix(d1, d2)
        for i1 in np.nonzero(want >= 0)[0]:
            n_ties += int((D[i1] == D[i1, want[i1]]).sum() > 1)
    assert n_ties > 200      # the accepted distance was shared by another keypoint of keyframe 2 in many of the matches
