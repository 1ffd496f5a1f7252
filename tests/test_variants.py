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
    with pytest.raises(AssertionError):
        ox.set_variant("tree_switch_factor", 2)
