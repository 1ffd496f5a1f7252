"""Knife-edge parity: the float decisions that used to go through two different maths libraries
(landmark::predict_scale_level's logf + ceil, the equirectangular asin / atan2 on the +-180 degree seam and at the poles) are driven
with inputs placed ON the decision boundary and a few ulp either side. Both sides evaluate include/ovs_detmath.h, so the match pairs
must be identical here too -- not only on inputs that keep clear of the edge."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SF = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)
LSF = float(np.log(np.float32(1.2)))


def _nudge(x, j):
    x = x.copy()
    for _ in range(abs(int(j))):
        x = np.nextafter(x, np.float32(np.inf if j > 0 else 0), dtype=np.float32)
    return x


def _knife_dmax(oracle, dist, rng):
    """max_valid_dist_ values whose ratio to the (float) camera distance is (float)1.2^k nudged by -2..+2 ulp: logf(ratio) / log(1.2f) sits
    on an integer, where one ulp of logf decides ceil()."""
    m = len(dist)
    df = dist.astype(np.float32)
    k = rng.integers(0, 9, m)
    j = rng.integers(-2, 3, m)
    powers = np.concatenate([[np.float32(1.0)], np.cumprod(np.full(9, np.float32(1.2), np.float32))]).astype(np.float32)
    target = powers[k].copy()
    for jj in (-2, -1, 1, 2):
        sel = j == jj
        target[sel] = _nudge(target[sel], jj)
    dmax = (target * df).astype(np.float32)
    ratio = (dmax / df).astype(np.float32)
    lg = oracle.detmath_eval(oracle.DETMATH_LOGF, ratio.astype(np.float64)).astype(np.float32)
    pred = np.ceil((lg / np.float32(LSF)).astype(np.float32)).astype(np.int64)
    # the construction really straddles the edge: for most exponents both ceil outcomes occur among the nudged copies
    straddle = sum(1 for kk in range(1, 8) if len(set(pred[k == kk].tolist())) > 1)
    assert straddle >= 5, straddle
    return dmax


def _scene(synth, model, rows, cols, n, seed):
    from test_gpu_window import _last_and_current
    return _last_and_current(synth, model, rows, cols, n, seed, 0.0)


@pytest.mark.parametrize("model", [0, 1])
def test_predict_scale_level_on_the_ceil_edge(oracle, model):
    from openvslam_amd import _lib, match, synth
    rows, cols, n = (960, 1920, 3000) if model == 1 else (720, 1280, 2000)
    ck, cd, Tc, lk, lpw, ld, _, valid, (fx, fy, cx, cy) = _scene(synth, model, rows, cols, n, 90 + model)
    m = len(lk)
    rng = np.random.default_rng(23)
    R, t = Tc[:, :3], Tc[:, 3]
    cc = -R.T @ t
    v = lpw - cc
    dist = np.linalg.norm(v, axis=1)
    dmax = _knife_dmax(oracle, dist, rng)
    dmin = (dmax / SF[7] * 0.5).astype(np.float32)
    dmm = np.ascontiguousarray(np.stack([dmin, dmax], 1))
    nrm = v / dist[:, None]
    ils = (1.0 / (SF * SF)).astype(np.float32)
    cam = _lib.Camera(model, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    ocam = oracle.Camera(model, 0, fx, fy, cx, cy, 0.0, 0.0, cols, rows)
    gp, ogp = match.grid_params(cols, rows), oracle.grid_params(cols, rows)
    w = match.fuse(0.6, max_targets=4096, max_queries=4096)
    for margin in (3.0, 8.0):
        got, gn = w.replace_duplication(cam, gp, ck, cd, Tc, lpw, dmm, nrm, ld, SF, ils, LSF, margin, lm_valid=valid)
        want, wn = oracle.fuse_replace_duplication(ocam, ogp, ck, cd, Tc, lpw, dmm, nrm, ld, SF, ils, LSF, margin, lm_valid=valid)
        assert gn == wn and np.array_equal(got, want)
    assert wn > 20
    w = match.projection(0.9, True, max_targets=4096, max_queries=4096)
    for margin in (10.0, 20.0):
        got, gn = w.match_frame_and_keyframe(cam, gp, ck, cd, Tc, lk, lpw, dmm, ld, SF, LSF, margin, 100, kf_valid=valid)
        want, wn = oracle.projection_match_frame_and_keyframe(ocam, ogp, ck, cd, Tc, lk, lpw, dmm, ld, SF, LSF, margin, 100, True, kf_valid=valid)
        assert gn == wn and np.array_equal(got, want)
    assert wn > 20


def test_equirectangular_seam_and_poles(oracle):
    """Landmarks exactly on / a hair either side of the +-180 degree seam (x = +-tiny, z < 0) and at the poles (asin(+-1), atan2(0, 0)),
    identity pose so the camera-frame coordinates are the world coordinates bit for bit; keypoints sit along both image borders."""
    from openvslam_amd import _lib, match, synth
    rows, cols, n = 960, 1920, 3000
    rng = np.random.default_rng(5)
    ck, cd = synth.synth_keypoints(n, rows, cols, seed=77)
    nb = 600   # keypoints hugging the left / right border and the top / bottom rows
    ck["x"][:nb // 2] = rng.integers(0, 6, nb // 2).astype(np.float32)
    ck["x"][nb // 2:nb] = (cols - 1 - rng.integers(0, 6, nb - nb // 2)).astype(np.float32)
    ck["y"][nb:nb + 100] = rng.integers(0, 4, 100).astype(np.float32)
    ck["y"][nb + 100:nb + 200] = (rows - 1 - rng.integers(0, 4, 100)).astype(np.float32)
    Tc = np.concatenate([np.eye(3), np.zeros((3, 1))], 1)
    Tl = Tc.copy()
    eps = np.array([0.0, -0.0, 5e-324, -5e-324, 1e-300, -1e-300, 1e-17, -1e-17, 1e-12, -1e-12, 1e-6, -1e-6, 1e-3, -1e-3])
    src = rng.integers(0, nb, 2000)
    depth = rng.uniform(2, 20, len(src))
    lat = -(ck["y"][src].astype(np.float64) / rows - 0.5) * np.pi
    x = eps[rng.integers(0, len(eps), len(src))] * depth
    lpw = np.stack([x, -np.sin(lat) * depth, -np.cos(lat) * depth], 1)       # longitude = +-pi up to the nudge
    poles = np.array([[0.0, 3.0, 0.0], [0.0, -3.0, 0.0], [-0.0, 2.0, -0.0], [1e-300, 5.0, 0.0], [0.0, 5.0, 1e-300], [0.0, -5.0, -1e-300]])
    psrc = np.concatenate([np.arange(nb, nb + 3), np.arange(nb + 100, nb + 103)])
    lpw = np.concatenate([lpw, poles])
    src = np.concatenate([src, psrc])
    lk = ck[src].copy()
    ld = np.stack([synth.flip_bits(rng, cd[i], 30) for i in src])
    cam = _lib.Camera(1, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, cols, rows)
    ocam = oracle.Camera(1, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, cols, rows)
    gp, ogp = match.grid_params(cols, rows), oracle.grid_params(cols, rows)
    # the projections themselves: +pi -> u = cols, -pi -> u = 0, poles -> v = 0 / rows
    uv = np.array([oracle.reproject_to_image(ocam, ogp, Tc, X)[1] for X in lpw[-6:]])
    assert set(np.round(uv[:, 1]).tolist()) == {0.0, float(rows)}
    w = match.projection(0.9, False, max_targets=4096, max_queries=4096)
    total = 0
    for margin in (7.0, 15.0):
        got, gn = w.match_current_and_last_frames(cam, gp, ck, cd, Tc, lk, lpw, ld, Tl, SF, margin)
        want, wn = oracle.projection_match_current_and_last_frames(ocam, ogp, ck, cd, Tc, lk, lpw, ld, Tl, SF, margin, False)
        assert gn == wn and np.array_equal(got, want)
        total += wn
    assert total > 50   # border keypoints really are found from both sides of the seam
