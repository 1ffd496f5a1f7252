"""GPU parity: the grid candidate generator (D1) and the windowed matchers (M3 projection::match_frame_and_landmarks, M5
area::match_in_consistent_area, M7 bow_tree::match_frame_and_keyframe, angle_checker) through the C ABI == CPU oracle, exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def match():
    from openvslam_amd import match
    return match


@pytest.fixture(scope="module")
def synth():
    from openvslam_amd import synth
    return synth


@pytest.mark.parametrize("n,rows,cols", [(4000, 1920, 3840), (1000, 480, 752), (1, 480, 752), (0, 480, 752), (2500, 376, 1241)])
def test_assign_keypoints_to_grid(match, synth, oracle, n, rows, cols):
    k, _ = synth.synth_keypoints(n, rows, cols, seed=n + 1)
    if n > 10:   # keypoints outside the image bounds fall in no cell; some exactly on cell borders (cvRound ties)
        k["x"][:5] = [-3.0, cols + 7.0, cols / 64 * 2.5, cols / 64 * 3.5, 0.0]
        k["y"][:5] = [10.0, 10.0, rows / 48 * 0.5, rows / 48 * 1.5, rows + 1.0]
    gp = match.grid_params(cols, rows)
    w = match.projection(0.8, False, max_targets=4096, max_queries=16)
    start, items = w.assign_keypoints_to_grid(gp, k)
    want_start, want_items = oracle.assign_keypoints_to_grid(oracle.grid_params(cols, rows), k)
    assert np.array_equal(start, want_start) and np.array_equal(items, want_items)


@pytest.mark.parametrize("n,m,rows,cols,stereo", [(4000, 10000, 1920, 3840, False), (2000, 3000, 1080, 1920, True), (300, 50, 480, 752, False),
                                                 (5, 2000, 480, 752, True)])
@pytest.mark.parametrize("ratio", [0.8, 0.6])
def test_projection_match_frame_and_landmarks(match, synth, oracle, n, m, rows, cols, stereo, ratio):
    """BASELINE config 4 geometry (3840x1920 / 4000 keypoints / 10 000 landmarks, margin 5) and smaller / stereo cases."""
    k, d = synth.synth_keypoints(n, rows, cols, seed=11 * n + m)
    lm = synth.synth_landmarks(k, d, m, rows, cols, seed=m, n_from_frame=min(m, int(1.3 * n)), with_stereo=stereo)
    rng = np.random.default_rng(m)
    sf = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)
    occ = (rng.random(n) < 0.1).astype(np.uint8)
    xr = None
    if stereo:   # some keypoints carry a right-image x; landmarks derived from them mostly agree
        xr = np.where(rng.random(n) < 0.7, k["x"] - rng.uniform(2, 60, n), -1.0).astype(np.float32)
        from_frame = lm["src"] >= 0
        agree = from_frame & (rng.random(m) < 0.8)
        lm["x_right"][agree] = xr[lm["src"][agree]] + rng.normal(0, 1.5, int(agree.sum())).astype(np.float32)
    gp, ogp = match.grid_params(cols, rows), oracle.grid_params(cols, rows)
    w = match.projection(ratio, True, max_targets=4096, max_queries=10240)
    for margin in (5.0, 15.0):
        got, gn = w.match_frame_and_landmarks(gp, k, d, sf, lm["xy"], lm["level"], lm["desc"], margin, frm_stereo_x_right=xr,
                                              frm_occupied=occ, lm_x_right=lm.get("x_right"), lm_valid=lm["valid"])
        want, wn = oracle.projection_match_frame_and_landmarks(ogp, k, d, sf, lm["xy"], lm["level"], lm["desc"], margin, ratio,
                                                               frm_stereo_x_right=xr, frm_occupied=occ, lm_x_right=lm.get("x_right"),
                                                               lm_valid=lm["valid"])
        assert gn == wn and np.array_equal(got, want)
        # the frame side resident in HBM (ovs_frame_dev: uploaded and indexed once, then handed to the matcher): identical pairs
        fd = match.frame_dev(gp, k, d, xr)
        for _ in range(2):
            got_f, gn_f = w.match_frame_and_landmarks(gp, fd, None, sf, lm["xy"], lm["level"], lm["desc"], margin, frm_occupied=occ,
                                                      lm_x_right=lm.get("x_right"), lm_valid=lm["valid"])
            assert gn_f == wn and np.array_equal(got_f, want)
        if n >= 2000:
            assert wn > n // 4   # the construction yields real matches and real collisions
            assigned = want[want >= 0]
            assert len(np.unique(assigned)) == len(assigned)


def _two_frames(oracle, synth, rows=480, cols=752, nfeat=1000, shift=(5, 0)):
    a = synth.synth_frame(rows, cols, seed=21)
    b = synth.synth_frame(rows, cols, seed=21, shift=shift, noise_seed=77)
    ox = oracle.OrbExtractor(oracle.make_params(nfeat))
    ka, da = ox.extract(a)
    kb, db = ox.extract(b)
    return ka, da, kb, db


@pytest.mark.parametrize("check_orientation", [True, False])
@pytest.mark.parametrize("ratio,margin", [(0.9, 100), (0.7, 30), (1.0, 200)])
def test_area_match_in_consistent_area(match, synth, oracle, check_orientation, ratio, margin):
    """BASELINE config 1: 752x480, 1000 features, initialisation matcher with margin 100 between two frames 5 px apart."""
    ka, da, kb, db = _two_frames(oracle, synth)
    gp, ogp = match.grid_params(752, 480), oracle.grid_params(752, 480)
    w = match.area(ratio, check_orientation, max_targets=2048, max_queries=2048)
    prev_g = np.ascontiguousarray(np.stack([ka["x"], ka["y"]], 1), np.float32)
    prev_o = prev_g.copy()
    for it in range(2):   # second call starts from the updated prev_matched_pts, as module::initializer does
        gn, got = w.match_in_consistent_area(gp, ka, da, kb, db, prev_g, margin)
        wn, want = oracle.area_match_in_consistent_area(ogp, ka, da, kb, db, prev_o, margin, ratio, check_orientation)
        assert gn == wn and np.array_equal(got, want) and np.array_equal(prev_g.view(np.uint32), prev_o.view(np.uint32))
    # both frames resident (the initializer matches its init frame against every incoming frame)
    f1, f2 = match.frame_dev(gp, ka, da), match.frame_dev(gp, kb, db)
    prev_f = np.ascontiguousarray(np.stack([ka["x"], ka["y"]], 1), np.float32)
    prev_o = prev_f.copy()
    for it in range(2):
        gn, got = w.match_in_consistent_area(gp, f1, None, f2, None, prev_f, margin)
        wn, want = oracle.area_match_in_consistent_area(ogp, ka, da, kb, db, prev_o, margin, ratio, check_orientation)
        assert gn == wn and np.array_equal(got, want) and np.array_equal(prev_f.view(np.uint32), prev_o.view(np.uint32))
    assert wn > 50
    assert (ka["octave"][want >= 0] == 0).all()


def test_area_steals_and_orientation(match, synth, oracle):
    """Many near-duplicate level-0 descriptors in a small area: later queries steal targets from earlier ones, and the
    orientation histogram (entries of stolen matches stay in it) removes the off-mode matches."""
    rng = np.random.default_rng(5)
    n = 400
    k1, _ = synth.synth_keypoints(n, 480, 752, seed=3)
    k1["octave"] = 0
    k1["x"] = rng.uniform(100, 300, n).astype(np.float32)
    k1["y"] = rng.uniform(100, 300, n).astype(np.float32)
    k2 = k1.copy()
    k2["x"] += rng.normal(0, 3, n).astype(np.float32)
    k2["angle"] = np.where(rng.random(n) < 0.7, k1["angle"] + rng.normal(0, 4, n), rng.uniform(0, 360, n)).astype(np.float32) % 360
    base = rng.integers(0, 256, size=(40, 32), dtype=np.uint8)
    d1 = np.stack([synth.flip_bits(rng, base[i % 40], 14) for i in range(n)])
    d2 = np.stack([synth.flip_bits(rng, base[i % 40], 14) for i in range(n)])
    gp, ogp = match.grid_params(752, 480), oracle.grid_params(752, 480)
    for ratio in (0.9, 1.0):
        for co in (True, False):
            w = match.area(ratio, co, max_targets=512, max_queries=512)
            pg = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], 1), np.float32)
            po = pg.copy()
            gn, got = w.match_in_consistent_area(gp, k1, d1, k2, d2, pg, 60)
            wn, want = oracle.area_match_in_consistent_area(ogp, k1, d1, k2, d2, po, 60, ratio, co)
            assert gn == wn and np.array_equal(got, want) and np.array_equal(pg, po)


@pytest.mark.parametrize("check_orientation", [True, False])
@pytest.mark.parametrize("ratio", [0.75, 0.6])
def test_bow_match_frame_and_keyframe(match, synth, oracle, check_orientation, ratio):
    ka, da, kb, db = _two_frames(oracle, synth, shift=(3, 2))
    fa, fb = synth.synth_bow(da, seed=1, n_nodes=120), synth.synth_bow(db, seed=1, n_nodes=120)
    fb.pop(sorted(fb)[3])   # node present on one side only: exercises the lower_bound skips
    has_lm = (np.random.default_rng(2).random(len(ka)) < 0.85).astype(np.uint8)
    w = match.bow_tree(ratio, check_orientation, max_targets=2048, max_queries=2048)
    gn, got = w.match_frame_and_keyframe(ka, da, fa, kb, db, fb, has_lm)
    wn, want = oracle.bow_match_frame_and_keyframe(ka, da, fa, kb, db, fb, ratio, check_orientation, has_lm)
    assert gn == wn and np.array_equal(got, want)
    assert wn > 30
    # keyframe and frame resident (round 4): same result, nothing but the landmark flags and the BoW vectors travels
    gp = match.grid_params(752, 480)
    fka, fkb = match.frame_dev(gp, ka, da), match.frame_dev(gp, kb, db)
    for _ in range(2):   # the handles are reusable
        gn_f, got_f = w.match_frame_and_keyframe(fka, None, fa, fkb, None, fb, has_lm)
        assert gn_f == wn and np.array_equal(got_f, want)


def _rot(axis, deg):
    a = np.radians(deg)
    c, s = np.cos(a), np.sin(a)
    x, y, z = axis
    return np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                     [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                     [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])


def _last_and_current(synth, model, rows, cols, n, seed, forward_z):
    """A current frame of n keypoints and a last frame whose landmarks are the current keypoints back-projected at random depths
    (so they reproject onto them up to noise), plus distractors; the last camera sits `forward_z` metres behind along z."""
    rng = np.random.default_rng(seed)
    ck, cd = synth.synth_keypoints(n, rows, cols, seed=seed)
    R = _rot((0, 1, 0), 2.0) @ _rot((1, 0, 0), -1.0)
    t = np.array([0.05, -0.02, 0.3])
    Tc = np.concatenate([R, t[:, None]], 1)
    Tl = np.concatenate([np.eye(3), np.array([[0.0], [0.0], [-forward_z]])], 1)   # pos_l = pos_w - (0,0,forward_z)... last camera pose
    fx = fy = 0.6 * cols
    cx, cy = cols / 2.0, rows / 2.0
    depth = rng.uniform(2.0, 20.0, n)
    u = ck["x"].astype(np.float64) + rng.normal(0, 1.5, n)
    v = ck["y"].astype(np.float64) + rng.normal(0, 1.5, n)
    if model == 0:
        pc = np.stack([(u - cx) / fx * depth, (v - cy) / fy * depth, depth], 1)
    else:
        lon = (u / cols - 0.5) * 2 * np.pi
        lat = -(v / rows - 0.5) * np.pi
        pc = np.stack([np.cos(lat) * np.sin(lon), -np.sin(lat), np.cos(lat) * np.cos(lon)], 1) * depth[:, None]
    pw = (pc - t) @ R          # R^T (pc - t)
    m = int(1.2 * n)
    src = np.concatenate([rng.permutation(n), rng.integers(0, n, m - n)])
    lk = ck[src].copy()
    lk["angle"] = (lk["angle"] + np.where(rng.random(m) < 0.8, rng.normal(0, 5, m), rng.uniform(0, 360, m))) % 360
    lk["octave"] = np.clip(lk["octave"] + rng.integers(-1, 2, m), 0, 7)
    lpw = pw[src] + rng.normal(0, 0.002, (m, 3))
    ld = np.stack([synth.flip_bits(rng, cd[i], 60) for i in src])
    far = rng.random(m) < 0.1   # some landmarks behind the camera / far outside the image
    lpw[far] = rng.uniform(-30, 30, (int(far.sum()), 3))
    valid = (rng.random(m) < 0.9).astype(np.uint8)
    return ck, cd, Tc, lk, lpw, ld, Tl, valid, (fx, fy, cx, cy)


@pytest.mark.parametrize("model,setup,forward_z", [(0, 0, 0.0), (0, 1, 2.0), (0, 1, -2.0), (0, 2, 0.0), (1, 0, 0.0)])
@pytest.mark.parametrize("check_orientation", [True, False])
def test_projection_match_current_and_last_frames(match, synth, oracle, model, setup, forward_z, check_orientation):
    from openvslam_amd import _lib
    rows, cols, n = (960, 1920, 3000) if model == 1 else (720, 1280, 2000)
    ck, cd, Tc, lk, lpw, ld, Tl, valid, (fx, fy, cx, cy) = _last_and_current(synth, model, rows, cols, n, 5 + model + setup, forward_z)
    cam = _lib.Camera(model, setup, fx, fy, cx, cy, 0.12 * fx, 0.12, cols, rows)
    ocam = oracle.Camera(model, setup, fx, fy, cx, cy, 0.12 * fx, 0.12, cols, rows)
    gp, ogp = match.grid_params(cols, rows), oracle.grid_params(cols, rows)
    sf = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)
    rng = np.random.default_rng(1)
    occ = (rng.random(n) < 0.05).astype(np.uint8)
    xr = None
    if setup:
        xr = np.where(rng.random(n) < 0.6, ck["x"] - rng.uniform(1, 40, n), -1.0).astype(np.float32)
    w = match.projection(0.9, check_orientation, max_targets=4096, max_queries=4096)
    for margin in (7.0, 15.0):
        got, gn = w.match_current_and_last_frames(cam, gp, ck, cd, Tc, lk, lpw, ld, Tl, sf, margin, curr_stereo_x_right=xr,
                                                  curr_occupied=occ, last_valid=valid)
        want, wn = oracle.projection_match_current_and_last_frames(ocam, ogp, ck, cd, Tc, lk, lpw, ld, Tl, sf, margin, check_orientation,
                                                                   curr_stereo_x_right=xr, curr_occupied=occ, last_valid=valid)
        assert gn == wn and np.array_equal(got, want)
        fd = match.frame_dev(gp, ck, cd, xr)   # the current frame resident
        got_f, gn_f = w.match_current_and_last_frames(cam, gp, fd, None, Tc, lk, lpw, ld, Tl, sf, margin, curr_occupied=occ, last_valid=valid)
        assert gn_f == wn and np.array_equal(got_f, want)
    assert wn > n // 10


@pytest.mark.parametrize("model,setup", [(0, 0), (0, 1), (1, 0)])
def test_fuse_replace_duplication(match, synth, oracle, model, setup):
    """fuse::replace_duplication's candidate search: landmarks back-projected from the keyframe's own keypoints (some with wrong
    depth range / viewing direction / level) plus distractors."""
    from openvslam_amd import _lib
    rows, cols, n = (960, 1920, 3000) if model == 1 else (720, 1280, 2000)
    ck, cd, Tc, lk, lpw, ld, _, valid, (fx, fy, cx, cy) = _last_and_current(synth, model, rows, cols, n, 40 + model + setup, 0.0)
    m = len(lk)
    rng = np.random.default_rng(7)
    R, t = Tc[:, :3], Tc[:, 3]
    cc = -R.T @ t
    v = lpw - cc
    dist = np.linalg.norm(v, axis=1)
    sf = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)
    ils = (1.0 / (sf * sf)).astype(np.float32)
    lvl = np.clip(lk["octave"] + rng.integers(0, 2, m), 0, 7)
    dmax = (dist * sf[lvl] * rng.uniform(0.85, 1.0, m)).astype(np.float32)      # predict_scale_level inverts this
    dmin = (dmax / sf[7] * rng.uniform(0.5, 1.3, m)).astype(np.float32)
    dmm = np.ascontiguousarray(np.stack([dmin, dmax], 1))
    nrm = v / dist[:, None]
    flip = rng.random(m) < 0.15
    nrm[flip] = rng.normal(0, 1, (int(flip.sum()), 3))
    cam = _lib.Camera(model, setup, fx, fy, cx, cy, 0.12 * fx, 0.12, cols, rows)
    ocam = oracle.Camera(model, setup, fx, fy, cx, cy, 0.12 * fx, 0.12, cols, rows)
    gp, ogp = match.grid_params(cols, rows), oracle.grid_params(cols, rows)
    xr = None
    if setup:
        xr = np.where(rng.random(n) < 0.6, ck["x"] - 0.12 * fx / rng.uniform(2, 20, n), -1.0).astype(np.float32)
    w = match.fuse(0.6, max_targets=4096, max_queries=4096)
    for margin in (3.0, 8.0):
        got, gn = w.replace_duplication(cam, gp, ck, cd, Tc, lpw, dmm, nrm, ld, sf, ils, float(np.log(np.float32(1.2))), margin,
                                        keyfrm_stereo_x_right=xr, lm_valid=valid)
        want, wn = oracle.fuse_replace_duplication(ocam, ogp, ck, cd, Tc, lpw, dmm, nrm, ld, sf, ils, float(np.log(np.float32(1.2))), margin,
                                                   kf_stereo_x_right=xr, lm_valid=valid)
        assert gn == wn and np.array_equal(got, want)
        kf = match.frame_dev(gp, ck, cd, xr)   # the keyframe resident
        got_f, gn_f = w.replace_duplication(cam, None, kf, None, Tc, lpw, dmm, nrm, ld, sf, ils, float(np.log(np.float32(1.2))), margin, lm_valid=valid)
        assert gn_f == wn and np.array_equal(got_f, want)
    assert wn > n // 20


@pytest.mark.parametrize("check_orientation", [True, False])
def test_bow_match_keyframes(match, synth, oracle, check_orientation):
    ka, da, kb, db = _two_frames(oracle, synth, shift=(2, 3))
    fa, fb = synth.synth_bow(da, seed=4, n_nodes=90), synth.synth_bow(db, seed=4, n_nodes=90)
    fa.pop(sorted(fa)[5])
    rng = np.random.default_rng(3)
    v1 = (rng.random(len(ka)) < 0.8).astype(np.uint8)
    v2 = (rng.random(len(kb)) < 0.8).astype(np.uint8)
    w = match.bow_tree(0.75, check_orientation, max_targets=2048, max_queries=2048)
    gn, got = w.match_keyframes(ka, da, fa, kb, db, fb, v1, v2)
    wn, want = oracle.bow_match_keyframes(ka, da, fa, kb, db, fb, 0.75, check_orientation, v1, v2)
    assert gn == wn and np.array_equal(got, want)
    gp = match.grid_params(752, 480)
    gn_f, got_f = w.match_keyframes(match.frame_dev(gp, ka, da), None, fa, match.frame_dev(gp, kb, db), None, fb, v1, v2)   # both keyframes resident
    assert gn_f == wn and np.array_equal(got_f, want)
    assert wn > 20 and (v1[want >= 0] == 1).all() and (v2[want[want >= 0]] == 1).all()


@pytest.mark.parametrize("model", [0, 1])
@pytest.mark.parametrize("check_orientation,thr", [(True, 100), (False, 50)])
def test_projection_match_frame_and_keyframe(match, synth, oracle, model, check_orientation, thr):
    from openvslam_amd import _lib
    rows, cols, n = (960, 1920, 3000) if model == 1 else (720, 1280, 2000)
    ck, cd, Tc, lk, lpw, ld, _, valid, (fx, fy, cx, cy) = _last_and_current(synth, model, rows, cols, n, 60 + model, 0.0)
    m = len(lk)
    rng = np.random.default_rng(17)
    R, t = Tc[:, :3], Tc[:, 3]
    dist = np.linalg.norm(lpw - (-R.T @ t), axis=1)
    sf = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)
    lvl = np.clip(lk["octave"] + rng.integers(-1, 2, m), 0, 7)
    dmax = (dist * sf[lvl] * rng.uniform(0.85, 1.0, m)).astype(np.float32)
    dmin = (dmax / sf[7] * rng.uniform(0.5, 1.3, m)).astype(np.float32)
    dmm = np.ascontiguousarray(np.stack([dmin, dmax], 1))
    occ = (rng.random(n) < 0.1).astype(np.uint8)
    cam = _lib.Camera(model, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    ocam = oracle.Camera(model, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    gp, ogp = match.grid_params(cols, rows), oracle.grid_params(cols, rows)
    lsf = float(np.log(np.float32(1.2)))
    w = match.projection(0.9, check_orientation, max_targets=4096, max_queries=4096)
    for margin in (10.0, 20.0):
        got, gn = w.match_frame_and_keyframe(cam, gp, ck, cd, Tc, lk, lpw, dmm, ld, sf, lsf, margin, thr, curr_occupied=occ, kf_valid=valid)
        want, wn = oracle.projection_match_frame_and_keyframe(ocam, ogp, ck, cd, Tc, lk, lpw, dmm, ld, sf, lsf, margin, thr, check_orientation,
                                                              curr_occupied=occ, kf_valid=valid)
        assert gn == wn and np.array_equal(got, want)
        got_f, gn_f = w.match_frame_and_keyframe(cam, None, match.frame_dev(gp, ck, cd), None, Tc, lk, lpw, dmm, ld, sf, lsf, margin, thr,
                                                 curr_occupied=occ, kf_valid=valid)   # the current frame resident
        assert gn_f == wn and np.array_equal(got_f, want)
    assert wn > n // 20


def _two_view_geometry(rows, cols, n, seed):
    """Two keyframes observing the same random 3D points (plus unrelated keypoints): keypoints, descriptors, unit bearings, the
    essential matrix E_12 (x1^T E_12 x2 = 0 for bearings) and keyframe 1's centre as a bearing in keyframe 2."""
    rng = np.random.default_rng(seed)
    fx = fy = 0.7 * cols
    cx, cy = cols / 2.0, rows / 2.0
    R1, t1 = np.eye(3), np.zeros(3)
    R2 = _rot((0, 1, 0), 4.0) @ _rot((1, 0, 0), 1.5)
    t2 = np.array([-0.6, 0.05, 0.1])
    X = np.stack([rng.uniform(-4, 4, n), rng.uniform(-2.5, 2.5, n), rng.uniform(4, 15, n)], 1)

    def project(R, t):
        pc = X @ R.T + t
        return np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1), pc

    return fx, fy, cx, cy, R1, t1, R2, t2, X, project


@pytest.mark.parametrize("check_orientation", [True, False])
@pytest.mark.parametrize("stereo", [False, True])
def test_robust_match_for_triangulation(match, synth, oracle, check_orientation, stereo):
    rows, cols, n = 720, 1280, 1500
    fx, fy, cx, cy, R1, t1, R2, t2, X, project = _two_view_geometry(rows, cols, n, 9)
    rng = np.random.default_rng(10)
    u1, _ = project(R1, t1)
    u2, _ = project(R2, t2)
    k1, d1 = synth.synth_keypoints(n, rows, cols, seed=31)
    k2 = k1.copy()
    k1["x"], k1["y"] = u1[:, 0], u1[:, 1]
    k2["x"], k2["y"] = u2[:, 0] + rng.normal(0, 0.4, n), u2[:, 1] + rng.normal(0, 0.4, n)
    off = rng.random(n) < 0.15                      # violate the epipolar constraint
    k2["y"][off] += rng.uniform(8, 40, int(off.sum()))
    k2["angle"] = (k1["angle"] + np.where(rng.random(n) < 0.8, rng.normal(0, 4, n), rng.uniform(0, 360, n))) % 360
    d2 = np.stack([synth.flip_bits(rng, d1[i], 45) for i in range(n)])
    dup = rng.integers(0, n, n // 5)                # near-duplicate descriptors: competing candidates
    d2[dup] = np.stack([synth.flip_bits(rng, d1[(i + 1) % n], 20) for i in dup])
    perm = rng.permutation(n)
    k2, d2 = k2[perm], d2[perm]

    def bearings(k):
        b = np.stack([(k["x"].astype(np.float64) - cx) / fx, (k["y"].astype(np.float64) - cy) / fy, np.ones(len(k))], 1)
        return b / np.linalg.norm(b, axis=1)[:, None]

    b1, b2 = bearings(k1), bearings(k2)
    R12, t12 = R1 @ R2.T, t1 - R1 @ R2.T @ t2       # x1 = R12 x2 + t12
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    E12 = tx @ R12
    c1_in_2 = R2 @ (-R1.T @ t1) + t2
    epipole = c1_in_2 / np.linalg.norm(c1_in_2)
    fv1, fv2 = synth.synth_bow(d1, seed=2, n_nodes=60), synth.synth_bow(d2, seed=2, n_nodes=60)
    h1 = (rng.random(n) < 0.3).astype(np.uint8)
    h2 = (rng.random(n) < 0.3).astype(np.uint8)
    x1 = x2 = None
    if stereo:
        x1 = np.where(rng.random(n) < 0.4, k1["x"] - 10, -1.0).astype(np.float32)
        x2 = np.where(rng.random(n) < 0.4, k2["x"] - 10, -1.0).astype(np.float32)
    sf = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)
    w = match.robust_triangulation(0.6, check_orientation, max_targets=2048, max_queries=2048)
    gn, pairs = w.match_for_triangulation(k1, d1, fv1, b1, k2, d2, fv2, b2, E12, epipole, sf, h1, h2, x1, x2)
    wn, want = oracle.robust_match_for_triangulation(k1, d1, fv1, b1, k2, d2, fv2, b2, E12, epipole, sf, check_orientation, h1, h2, x1, x2)
    idx = np.nonzero(want >= 0)[0]
    assert gn == wn and np.array_equal(pairs, np.stack([idx, want[idx]], 1))
    # both keyframes resident, stereo_x_right and bearings included
    gp = match.grid_params(cols, rows, min_x=-200.0, min_y=-200.0)
    f1 = match.frame_dev(gp, k1, d1, x1).attach_bearings(b1)
    f2 = match.frame_dev(gp, k2, d2, x2).attach_bearings(b2)
    gn_f, pairs_f = w.match_for_triangulation(f1, None, fv1, None, f2, None, fv2, None, E12, epipole, sf, h1, h2)
    assert gn_f == wn and np.array_equal(pairs_f, pairs)
    assert wn > 50 and (h1[idx] == 0).all() and (h2[want[idx]] == 0).all()


def _sim3_scene(synth, model, seed, scale=1.7):
    """The fuse scene (landmarks back-projected from the keyframe's keypoints, some with a wrong range / normal / level) seen through
    a Sim3 pose [sR | st]: the decomposition has to recover (R, t)."""
    rows, cols, n = (960, 1920, 3000) if model == 1 else (720, 1280, 2000)
    ck, cd, Tc, lk, lpw, ld, _, valid, intr = _last_and_current(synth, model, rows, cols, n, seed, 0.0)
    m = len(lk)
    rng = np.random.default_rng(seed + 1)
    R, t = Tc[:, :3], Tc[:, 3]
    v = lpw - (-R.T @ t)
    dist = np.linalg.norm(v, axis=1)
    sf = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)
    lvl = np.clip(lk["octave"] + rng.integers(0, 2, m), 0, 7)
    dmax = (dist * sf[lvl] * rng.uniform(0.85, 1.0, m)).astype(np.float32)
    dmin = (dmax / sf[7] * rng.uniform(0.5, 1.3, m)).astype(np.float32)
    dmm = np.ascontiguousarray(np.stack([dmin, dmax], 1))
    nrm = v / dist[:, None]
    flip = rng.random(m) < 0.15
    nrm[flip] = rng.normal(0, 1, (int(flip.sum()), 3))
    S = np.concatenate([scale * R, (scale * t)[:, None]], 1)
    return rows, cols, n, ck, cd, S, lpw, dmm, nrm, ld, valid, sf, intr


@pytest.mark.parametrize("model", [0, 1])
def test_fuse_detect_duplication(match, synth, oracle, model):
    from openvslam_amd import _lib
    rows, cols, n, ck, cd, S, lpw, dmm, nrm, ld, valid, sf, (fx, fy, cx, cy) = _sim3_scene(synth, model, 80 + model)
    cam = _lib.Camera(model, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    ocam = oracle.Camera(model, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    gp, ogp = match.grid_params(cols, rows), oracle.grid_params(cols, rows)
    lsf = float(np.log(np.float32(1.2)))
    w = match.fuse(0.6, max_targets=4096, max_queries=4096)
    for margin in (4.0, 10.0):
        got, gn = w.detect_duplication(cam, gp, ck, cd, S, lpw, dmm, nrm, ld, sf, lsf, margin, lm_valid=valid)
        want, wn = oracle.fuse_detect_duplication(ocam, ogp, ck, cd, S, lpw, dmm, nrm, ld, sf, lsf, margin, lm_valid=valid)
        assert gn == wn and np.array_equal(got, want)
        got_f, gn_f = w.detect_duplication(cam, None, match.frame_dev(gp, ck, cd), None, S, lpw, dmm, nrm, ld, sf, lsf, margin, lm_valid=valid)
        assert gn_f == wn and np.array_equal(got_f, want)
    assert wn > n // 20


@pytest.mark.parametrize("model", [0, 1])
def test_projection_match_by_sim3_transform(match, synth, oracle, model):
    from openvslam_amd import _lib
    rows, cols, n, ck, cd, S, lpw, dmm, nrm, ld, valid, sf, (fx, fy, cx, cy) = _sim3_scene(synth, model, 90 + model, scale=0.6)
    rng = np.random.default_rng(5)
    # several landmarks per keypoint (duplicates later in the list) so that the sequential claim decides
    extra = rng.integers(0, len(lpw), len(lpw) // 3)
    lpw2 = np.concatenate([lpw, lpw[extra] + rng.normal(0, 0.002, (len(extra), 3))])
    dmm2, nrm2, valid2 = np.concatenate([dmm, dmm[extra]]), np.concatenate([nrm, nrm[extra]]), np.concatenate([valid, valid[extra]])
    ld2 = np.concatenate([ld, np.stack([synth.flip_bits(rng, ld[i], 6) for i in extra])])
    occ = (rng.random(n) < 0.1).astype(np.uint8)
    cam = _lib.Camera(model, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    ocam = oracle.Camera(model, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    gp, ogp = match.grid_params(cols, rows), oracle.grid_params(cols, rows)
    lsf = float(np.log(np.float32(1.2)))
    w = match.projection(0.9, False, max_targets=4096, max_queries=8192, max_entries=1 << 20)
    for margin in (5.0, 10.0):
        got, gn = w.match_by_Sim3_transform(cam, gp, ck, cd, S, lpw2, dmm2, nrm2, ld2, sf, lsf, margin, keyfrm_occupied=occ, lm_valid=valid2)
        want, wn = oracle.projection_match_by_sim3_transform(ocam, ogp, ck, cd, S, lpw2, dmm2, nrm2, ld2, sf, lsf, margin, kf_occupied=occ,
                                                             lm_valid=valid2)
        assert gn == wn and np.array_equal(got, want)
        got_f, gn_f = w.match_by_Sim3_transform(cam, None, match.frame_dev(gp, ck, cd), None, S, lpw2, dmm2, nrm2, ld2, sf, lsf, margin,
                                                keyfrm_occupied=occ, lm_valid=valid2)
        assert gn_f == wn and np.array_equal(got_f, want)
    assert wn > n // 20
    hit = want[want >= 0]
    assert len(np.unique(hit)) == len(hit) and not occ[hit].any()


@pytest.mark.parametrize("s_12", [1.0, 1.35])
def test_projection_match_keyframes_mutually(match, synth, oracle, s_12):
    """Two keyframes of the same points; keyframe 2's map is s_12 times smaller (scale drift), Sim3_12 = (s_12, R_12, t_12) undoes it."""
    from openvslam_amd import _lib
    rows, cols, n = 720, 1280, 1800
    fx, fy, cx, cy, R1, t1, R2, t2, X, project = _two_view_geometry(rows, cols, n, 21)
    rng = np.random.default_rng(22)
    u1, p1 = project(R1, t1)
    u2, p2 = project(R2, t2)
    k1, d1 = synth.synth_keypoints(n, rows, cols, seed=33)
    k2 = k1.copy()
    k1["x"], k1["y"] = u1[:, 0] + rng.normal(0, 0.5, n), u1[:, 1] + rng.normal(0, 0.5, n)
    k2["x"], k2["y"] = u2[:, 0] + rng.normal(0, 0.5, n), u2[:, 1] + rng.normal(0, 0.5, n)
    k2["octave"] = np.clip(k1["octave"] + rng.integers(-1, 2, n), 0, 7)
    d2 = np.stack([synth.flip_bits(rng, d1[i], 40) for i in range(n)])
    dup = rng.integers(0, n, n // 6)                 # confusable descriptors: the two directions disagree for some
    d2[dup] = np.stack([synth.flip_bits(rng, d1[(i + 7) % n], 25) for i in dup])
    perm = rng.permutation(n)
    k2, d2, X2, p2 = k2[perm], d2[perm], X[perm] / s_12, p2[perm]
    T1 = np.concatenate([R1, t1[:, None]], 1)
    T2 = np.concatenate([R2, (t2 / s_12)[:, None]], 1)
    R12, t12 = R1 @ R2.T, t1 - R1 @ R2.T @ t2
    sf = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)
    inv = np.argsort(perm)
    # landmark of keypoint i in keyframe 1 lands on keypoint inv[i] of keyframe 2 at distance |p2| / s_12; and the other way round |p1|
    dist_1_in_2 = np.linalg.norm(p2[inv], axis=1) / s_12
    dist_2_in_1 = np.linalg.norm(p1[perm], axis=1)

    def ranges(dist, octave):
        lvl = np.clip(octave + rng.integers(0, 2, n), 0, 7)
        dmax = (dist * sf[lvl] * rng.uniform(0.85, 1.0, n)).astype(np.float32)
        dmin = (dmax / sf[7] * rng.uniform(0.5, 1.3, n)).astype(np.float32)
        return np.ascontiguousarray(np.stack([dmin, dmax], 1))

    dm1, dm2 = ranges(dist_1_in_2, k2["octave"][inv]), ranges(dist_2_in_1, k1["octave"][perm])
    l1 = np.stack([synth.flip_bits(rng, d1[i], 10) for i in range(n)])
    l2 = np.stack([synth.flip_bits(rng, d2[i], 10) for i in range(n)])
    v1, v2 = (rng.random(n) < 0.85).astype(np.uint8), (rng.random(n) < 0.85).astype(np.uint8)
    cam = _lib.Camera(0, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    ocam = oracle.Camera(0, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    gp, ogp = match.grid_params(cols, rows), oracle.grid_params(cols, rows)
    lsf = float(np.log(np.float32(1.2)))
    w = match.projection(0.9, False, max_targets=4096, max_queries=4096)
    for margin in (7.5, 15.0):
        gn, got = w.match_keyframes_mutually(cam, gp, k1, d1, T1, X, dm1, l1, v1, k2, d2, T2, X2, dm2, l2, v2, s_12, R12, t12, sf, lsf, margin)
        wn, want = oracle.projection_match_keyframes_mutually(ocam, ogp, k1, d1, T1, X, dm1, l1, v1, k2, d2, T2, X2, dm2, l2, v2, s_12, R12, t12,
                                                              sf, lsf, margin)
        assert gn == wn and np.array_equal(got, want)
        gn_f, got_f = w.match_keyframes_mutually(cam, None, match.frame_dev(gp, k1, d1), None, T1, X, dm1, l1, v1, match.frame_dev(gp, k2, d2), None, T2,
                                                 X2, dm2, l2, v2, s_12, R12, t12, sf, lsf, margin)
        assert gn_f == wn and np.array_equal(got_f, want)
    assert wn > n // 5
    ok = want >= 0
    assert (want[ok] == inv[ok]).mean() > 0.9 and v1[ok].all() and v2[want[ok]].all()


@pytest.mark.parametrize("n_nodes,ratio,check_orientation", [(3, 0.9, False), (6, 0.75, True), (12, 0.6, True), (3, 1.0, False)])
def test_resolver_under_heavy_contention(match, synth, oracle, n_nodes, ratio, check_orientation):
    """The sequential-claim resolvers where almost every query competes for the same targets: bow_tree::match_frame_and_keyframe over a
    vocabulary of 3-12 nodes (lists of hundreds of shared candidates per query -- the shape on which an interim commit rule of round 3
    failed, fuzz seed 901) and area::match_in_consistent_area with a window covering the image. Fixed form of
    `tools/fuzz_parity.py --contention` (40 random cases of it: profiles/r04a_fuzz.txt)."""
    rows, cols = 336, 292
    a = synth.synth_frame(rows, cols, seed=4003)
    b = synth.synth_frame(rows, cols, seed=4003, shift=(7, 3), noise_seed=11)
    ox = oracle.OrbExtractor(oracle.make_params(1000))
    ka, da = ox.extract(a)
    kb, db = ox.extract(b)
    fa, fb = synth.synth_bow(da, seed=1, n_nodes=n_nodes), synth.synth_bow(db, seed=1, n_nodes=n_nodes)
    has_lm = (np.random.default_rng(5).random(len(ka)) < 0.9).astype(np.uint8)
    w = match.bow_tree(ratio, check_orientation, max_targets=4096, max_queries=4096)
    gn, got = w.match_frame_and_keyframe(ka, da, fa, kb, db, fb, has_lm)
    wn, want = oracle.bow_match_frame_and_keyframe(ka, da, fa, kb, db, fb, ratio, check_orientation, has_lm)
    assert gn == wn and np.array_equal(got, want)
    assert wn > 50 and max(len(v) for v in fa.values()) > len(ka) // (2 * n_nodes)
    gp, ogp = match.grid_params(cols, rows), oracle.grid_params(cols, rows)
    wa = match.area(ratio, check_orientation, max_targets=4096, max_queries=4096)
    pg = np.ascontiguousarray(np.stack([ka["x"], ka["y"]], 1), np.float32)
    po = pg.copy()
    gn, got = wa.match_in_consistent_area(gp, ka, da, kb, db, pg, 400)
    wn, want = oracle.area_match_in_consistent_area(ogp, ka, da, kb, db, po, 400, ratio, check_orientation)
    assert gn == wn and np.array_equal(got, want) and np.array_equal(pg.view(np.uint32), po.view(np.uint32))


def test_angle_tie_order_variant_on_both_sides(match, oracle):
    """ORACLE_SPEC rule 17's tie order (equally full bins: lower first | higher first) as a run-time variant of both sides, on a case built so
    that ONLY the tie decides: area matching with orientation check, four groups of exact-copy matches whose rotation differences fall into
    bins 1 (12 matches) and 3, 6, 8 (5 matches each). Default keeps bins 1, 3, 6; the variant keeps 1, 6, 8; device == oracle in both."""
    rng = np.random.default_rng(5)
    groups = [(1, 12), (3, 5), (6, 5), (8, 5)]
    n = sum(c for _, c in groups)
    ka = np.zeros(n, oracle.KP_DTYPE)
    kb = np.zeros(n, oracle.KP_DTYPE)
    ka["x"] = kb["x"] = 40.0 + 25.0 * (np.arange(n) % 9)
    ka["y"] = kb["y"] = 40.0 + 25.0 * (np.arange(n) // 9)
    ka["size"] = kb["size"] = 31.0
    da = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    db = da.copy()
    i = 0
    for b, c in groups:
        ka["angle"][i:i + c] = 30.0 * b + 5.0      # delta = angle_a - angle_b = 30 b degrees exactly -> bin b
        kb["angle"][i:i + c] = 5.0
        i += c
    gp, ogp = match.grid_params(320, 240), oracle.grid_params(320, 240)
    res = {}
    try:
        for order in (0, 1):
            match.set_variant("angle_tie_order", order)
            oracle.match_set_variant("angle_tie_order", order)
            w = match.area(0.9, True, max_targets=256, max_queries=256)
            pg = np.ascontiguousarray(np.stack([ka["x"], ka["y"]], 1), np.float32)
            po = pg.copy()
            gn, got = w.match_in_consistent_area(gp, ka, da, kb, db, pg, 10)
            wn, want = oracle.area_match_in_consistent_area(ogp, ka, da, kb, db, po, 10, 0.9, True)
            assert gn == wn == 22 and np.array_equal(got, want)
            res[order] = got.copy()
    finally:
        match.set_variant("angle_tie_order", 0)
        oracle.match_set_variant("angle_tie_order", 0)
    assert (res[0][12:22] >= 0).all() and (res[0][22:] < 0).all()                                  # bins 1, 3, 6 kept
    assert (res[1][12:17] < 0).all() and (res[1][17:] >= 0).all() and (res[1][:12] >= 0).all()     # bins 1, 6, 8 kept


def test_angle_keep_rule_variant_on_both_sides(match, synth, oracle):
    """ORACLE_SPEC rule 17's alternative (ORB-SLAM2's 0.1 x max rule in angle_checker) as a process-wide run-time variant of BOTH sides
    (ovs_match_set_variant / ovo_match_set_variant): the device resolver and the oracle agree in either setting, and on a frame pair with one
    dominant rotation bin the rule removes matches the default keeps."""
    ka, da, kb, db = _two_frames(oracle, synth, shift=(3, 2))
    gp, ogp = match.grid_params(752, 480), oracle.grid_params(752, 480)
    fa, fb = synth.synth_bow(da, seed=1, n_nodes=120), synth.synth_bow(db, seed=1, n_nodes=120)
    has_lm = np.ones(len(ka), np.uint8)
    res = {}
    try:
        for rule in (0, 1):
            match.set_variant("angle_keep_rule", rule)
            oracle.match_set_variant("angle_keep_rule", rule)
            w = match.area(0.9, True, max_targets=2048, max_queries=2048)
            pg = np.ascontiguousarray(np.stack([ka["x"], ka["y"]], 1), np.float32)
            po = pg.copy()
            gn, got = w.match_in_consistent_area(gp, ka, da, kb, db, pg, 100)
            wn, want = oracle.area_match_in_consistent_area(ogp, ka, da, kb, db, po, 100, 0.9, True)
            assert gn == wn and np.array_equal(got, want)
            wb = match.bow_tree(0.75, True, max_targets=2048, max_queries=2048)
            bn, bgot = wb.match_frame_and_keyframe(ka, da, fa, kb, db, fb, has_lm)
            own, owant = oracle.bow_match_frame_and_keyframe(ka, da, fa, kb, db, fb, 0.75, True, has_lm)
            assert bn == own and np.array_equal(bgot, owant)
            res[rule] = (wn, own)
    finally:
        match.set_variant("angle_keep_rule", 0)
        oracle.match_set_variant("angle_keep_rule", 0)
    assert res[1][0] <= res[0][0] and res[1][1] <= res[0][1] and res[1] != res[0]
