"""The three rules of the extraction that cannot be pinned without upstream's sources / an OpenCV build (oracle/ORACLE_SPEC.md rules 6, 7, 10) as
run-time variants of the CPU oracle (the HIP library has the same switches: tests/test_gpu_orb.py::test_extract_bit_exact_under_every_variant).
Known answers worked by hand for the alternatives, and the defaults are shown to be the defaults."""
import numpy as np
import pytest


def test_blur_taps_variants_by_hand(oracle):
    img = np.zeros((15, 15), np.uint8)
    img[7, 7] = 255
    a, b = oracle.gaussian_blur(img, 0), oracle.gaussian_blur(img, 1)
    g0, g1 = np.array([18, 34, 48, 56, 48, 34, 18]), np.array([18, 34, 49, 55, 49, 34, 18])
    for out, g in ((a, g0), (b, g1)):
        want = (np.outer(g, g) * 255 + 32768) >> 16          # impulse response: row pass 255 g_j, column pass g_i (255 g_j), round half up
        assert np.array_equal(out[4:11, 4:11], want)
    assert np.array_equal(oracle.gaussian_blur(img), a)       # default = error-diffused taps
    flat = np.full((12, 12), 255, np.uint8)
    assert (oracle.gaussian_blur(flat, 0) == 255).all()       # taps sum to 256: exact
    assert (oracle.gaussian_blur(flat, 1) == 255).all()       # taps sum to 257: 255 * 257^2 / 65536 = 257 -> saturates
    v = np.full((12, 12), 200, np.uint8)
    assert (oracle.gaussian_blur(v, 0) == 200).all() and (oracle.gaussian_blur(v, 1) == 202).all()   # (200 * 66049 + 32768) >> 16 = 202


def _clustered(seed, n):
    rng = np.random.default_rng(seed)
    c = rng.uniform(60, 540, (12, 2))
    p = c[rng.integers(0, 12, n)] + rng.normal(0, 25, (n, 2))
    xs = np.clip(np.rint(p[:, 0]), 20, 579).astype(np.float32)
    ys = np.clip(np.rint(p[:, 1]), 20, 379).astype(np.float32)
    return xs, ys, rng.integers(5, 90, n).astype(np.float32)


@pytest.mark.parametrize("factor", [3, 1])
@pytest.mark.parametrize("tie", [False, True])
def test_tree_variants_are_valid_selections(oracle, factor, tie):
    """Whatever the switch factor and the tie order: at most one keypoint per final node, at least min(N, distinct positions) of them when the
    tree can grow that far, every selected keypoint is the strongest of its node -- and the default arguments are (3, later-created first)."""
    xs, ys, rs = _clustered(3, 4000)
    for N in (50, 300, 1200):
        sel = oracle.distribute_via_tree(xs, ys, rs, 19, 581, 19, 381, N, switch_factor=factor, tie_earlier_first=tie)
        # factor 3: the last single split adds at most 3 nodes; factor 1: the last all-at-once pass starts from size + pool <= N and adds <= 2 pool <= N
        assert len(set(sel.tolist())) == len(sel) and N <= len(sel) <= (N + 3 if factor == 3 else 2 * N)
        if factor == 3 and not tie:
            assert np.array_equal(sel, oracle.distribute_via_tree(xs, ys, rs, 19, 581, 19, 381, N))


def test_tree_variants_differ_where_they_should(oracle):
    """The alternatives are not no-ops: on clustered input with many equal counts the tie order changes which nodes the sorted phase splits
    first, and the switch factor changes when that phase starts."""
    differ_tie = differ_factor = 0
    for seed in range(6):
        xs, ys, rs = _clustered(10 + seed, 3000)
        base = oracle.distribute_via_tree(xs, ys, rs, 19, 581, 19, 381, 400)
        differ_tie += not np.array_equal(base, oracle.distribute_via_tree(xs, ys, rs, 19, 581, 19, 381, 400, tie_earlier_first=True))
        differ_factor += not np.array_equal(base, oracle.distribute_via_tree(xs, ys, rs, 19, 581, 19, 381, 400, switch_factor=1))
    assert differ_tie >= 1 and differ_factor >= 1


def test_trig_variant_descriptor_by_hand(oracle):
    """rule 11's alternative on one keypoint: the descriptor under `trig` = 1 equals a plain-Python evaluation with libm's float sin / cos."""
    import ctypes
    import math
    rng = np.random.default_rng(9)
    blurred = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    libm = ctypes.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = ctypes.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [ctypes.c_float]
    pat = oracle.orb_pattern()
    for angle_deg in (0.0, 33.3, 123.456, 271.0, 359.99):
        a32 = np.float32(angle_deg)
        rad = np.float32(np.float64(a32) * math.pi / 180.0)
        s, c = np.float32(libm.sinf(rad)), np.float32(libm.cosf(rad))
        want = np.zeros(32, np.uint8)
        for i in range(256):
            v = []
            for (px, py) in ((pat[i][0], pat[i][1]), (pat[i][2], pat[i][3])):
                fx, fy = np.float32(px), np.float32(py)
                dy = int(np.rint(np.float32(np.float32(fx * s) + np.float32(fy * c))))
                dx = int(np.rint(np.float32(np.float32(fx * c) - np.float32(fy * s))))
                v.append(int(blurred[32 + dy, 32 + dx]))
            want[i // 8] |= (v[0] < v[1]) << (i % 8)
        assert np.array_equal(oracle.orb_descriptor(blurred, 32, 32, float(a32), trig_variant=1), want), angle_deg
        if angle_deg:
            assert np.array_equal(oracle.orb_descriptor(blurred, 32, 32, float(a32)), oracle.orb_descriptor(blurred, 32, 32, float(a32), trig_variant=0))


def test_extractor_variants_change_only_their_stage(oracle):
    from openvslam_amd.synth import synth_frame
    img = synth_frame(240, 320, seed=5)
    ox = oracle.OrbExtractor(oracle.make_params(500))
    k0, d0 = ox.extract(img)
    ox.set_variant("blur_taps", 1)
    k1, d1 = ox.extract(img)
    assert np.array_equal(k0.view(np.uint8), k1.view(np.uint8)) and not np.array_equal(d0, d1)   # same keypoints and angles, other descriptors
    ox.set_variant("blur_taps", 0)
    ox.set_variant("tree_switch_factor", 1)
    k2, d2 = ox.extract(img)
    assert len(k2) > 0 and (len(k2) != len(k0) or not np.array_equal(k0.view(np.uint8), k2.view(np.uint8)))
    ox.set_variant("tree_switch_factor", 3)
    k3, d3 = ox.extract(img)
    assert np.array_equal(k0.view(np.uint8), k3.view(np.uint8)) and np.array_equal(d0, d3)
    ox.set_variant("trig", 1)   # libm's cosf / sinf instead of util::cos / util::sin: same keypoints and angles, other descriptor bits
    k4, d4 = ox.extract(img)
    assert np.array_equal(k0.view(np.uint8), k4.view(np.uint8)) and not np.array_equal(d0, d4)
    flipped = np.unpackbits(d0 ^ d4, axis=1).sum(1)
    assert 0 < flipped.mean() < 40   # a ~1e-3 change of sin / cos moves a rounded sample position now and then: some bits, not all
    ox.set_variant("trig", 0)
    k5, d5 = ox.extract(img)
    assert np.array_equal(d0, d5)
    with pytest.raises(AssertionError):
        ox.set_variant("tree_switch_factor", 2)


def test_stereo_variants_are_what_they_say(oracle):
    """ORACLE_SPEC rule 20's alternatives on the oracle alone: the 2.1 outlier factor keeps a superset of the 2.0 matches with identical values,
    the double parabola moves stereo_x_right by at most a few float ulp and never changes which keypoints match."""
    from openvslam_amd import synth
    left, right, _ = synth.synth_stereo_pair(240, 400, seed=3)
    oxl, oxr = oracle.OrbExtractor(oracle.make_params(500)), oracle.OrbExtractor(oracle.make_params(500))
    kl, dl = oxl.extract(left)
    kr, dr = oxr.extract(right)
    x0, d0, n0 = oracle.stereo_compute(oxl, oxr, kl, dl, kr, dr, 386.1448, 0.5372)
    x1, d1, n1 = oracle.stereo_compute(oxl, oxr, kl, dl, kr, dr, 386.1448, 0.5372, outlier_factor_21=True)
    keep0 = x0 >= 0
    assert n1 >= n0 > 10 and np.array_equal(x1[keep0], x0[keep0]) and np.array_equal(d1[keep0], d0[keep0])
    x2, d2, n2 = oracle.stereo_compute(oxl, oxr, kl, dl, kr, dr, 386.1448, 0.5372, parabola_double=True)
    assert n2 == n0 and np.array_equal(x2 >= 0, keep0)
    ulp = np.abs(x2[keep0].view(np.int32).astype(np.int64) - x0[keep0].view(np.int32).astype(np.int64))
    assert ulp.max() <= 4


def test_pose_reset_each_round_variant_on_the_oracle(oracle):
    """rule 25 (iv): with the frame vertex re-set every round the four rounds all start from the input pose; the result differs from the
    default schedule in the last digits only (same optimum), and the switch is process-wide and restorable."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_pose import make_frame
    T0, obs, cam, bf, _ = make_frame(oracle.POSE_OBS_DTYPE, 800, 800, 0.3, 0.15, 1.5)
    Td, od, nd = oracle.pose_optimize(T0, obs, cam, bf)
    try:
        oracle.pose_set_variant("reset_each_round", 1)
        Tr, orr, nr = oracle.pose_optimize(T0, obs, cam, bf)
    finally:
        oracle.pose_set_variant("reset_each_round", 0)
    assert not np.array_equal(Td, Tr) and np.allclose(Td, Tr, rtol=0, atol=1e-4) and abs(nd - nr) <= 3
    T2, _, _ = oracle.pose_optimize(T0, obs, cam, bf)
    assert np.array_equal(T2, Td)


def test_angle_keep_rule_variant_by_hand(oracle):
    """rule 17's alternative: bins of 40 / 3 / 2 entries -- the default keeps all three; with ORB-SLAM2's rule the second (3 < 0.1 * 40) drops out
    and takes the third with it. Bins 40 / 10 / 3: only the third drops. Process-wide switch, restorable."""
    def deltas(counts):   # bin b <-> delta = 30 b degrees
        return np.concatenate([np.full(c, 30.0 * b, np.float32) for b, c in counts.items()])
    d1 = deltas({2: 40, 5: 3, 9: 2, 11: 1})
    d2 = deltas({2: 40, 5: 10, 9: 3, 11: 1})
    inv = oracle.angle_checker_invalid
    assert inv(d1).sum() == 1 and inv(d2).sum() == 1                     # default: top three stay, only the 1-entry bin goes
    try:
        oracle.match_set_variant("angle_keep_rule", 1)
        assert inv(d1).sum() == 3 + 2 + 1 and inv(d1)[:40].sum() == 0    # only the fullest bin survives
        assert inv(d2).sum() == 3 + 1 and inv(d2)[:50].sum() == 0        # 10 >= 4 stays, 3 < 4 goes
    finally:
        oracle.match_set_variant("angle_keep_rule", 0)
    assert inv(d1).sum() == 1


def test_angle_tie_order_variant_by_hand(oracle):
    """rule 17's tie order: bins 2 / 5 / 9 / 11 hold 40 / 7 / 7 / 7 entries -- three candidates for the two places behind the fullest bin. Default:
    the LOWER bins win (5 and 9 stay, 11 goes); variant: the higher ones (9 and 11 stay, 5 goes). No tie -> no difference. Restorable."""
    def deltas(counts):   # bin b <-> delta = 30 b degrees
        return np.concatenate([np.full(c, 30.0 * b, np.float32) for b, c in counts.items()])
    d = deltas({2: 40, 5: 7, 9: 7, 11: 7})
    inv = oracle.angle_checker_invalid
    assert inv(d)[:54].sum() == 0 and inv(d)[54:].sum() == 7              # bins 2, 5, 9 kept
    no_tie = deltas({2: 40, 5: 9, 9: 8, 11: 7})
    base = inv(no_tie).copy()
    try:
        oracle.match_set_variant("angle_tie_order", 1)
        got = inv(d)
        assert got[:40].sum() == 0 and got[40:47].sum() == 7 and got[47:].sum() == 0   # bin 5 dropped, 9 and 11 kept
        assert np.array_equal(inv(no_tie), base)
    finally:
        oracle.match_set_variant("angle_tie_order", 0)
    assert inv(d)[54:].sum() == 7
