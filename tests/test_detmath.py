"""include/ovs_detmath.h: the deterministic log / asin / acos / atan2 that replace libm wherever a float decides a match pair
(VERDICT round 1, weak #2). CPU part: pinned against glibc through numpy (an independent implementation), so a transcription error
in a coefficient shows up here and not as a silently shared bug. GPU part: gfx950 and the host produce identical bits."""
import numpy as np
import pytest

from oracle import binding as ob


def _ulp_diff(a, b):
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia[ia < 0] = np.int64(-2**63) - ia[ia < 0]   # monotone map of the sign-magnitude encoding
    ib[ib < 0] = np.int64(-2**63) - ib[ib < 0]
    return np.abs(ia - ib)


def _samples(rng):
    x = np.concatenate([rng.uniform(-1, 1, 400000), np.linspace(-1, 1, 20001), np.array([0.5, -0.5, 0.975, -0.975, 1.0, -1.0, 0.0, -0.0]),
                        np.nextafter(np.array([0.5, -0.5, 0.975, 1.0, -1.0]), 0.0), rng.uniform(-1e-6, 1e-6, 1000),
                        1.0 - np.geomspace(1e-16, 1e-2, 2000), -1.0 + np.geomspace(1e-16, 1e-2, 2000)])
    return x


def test_asin_acos_within_one_ulp_of_glibc():
    x = _samples(np.random.default_rng(1))
    got = ob.detmath_eval(ob.DETMATH_ASIN, x)
    assert _ulp_diff(got, np.arcsin(x)).max() <= 1
    got = ob.detmath_eval(ob.DETMATH_ACOS, x)
    assert _ulp_diff(got, np.arccos(x)).max() <= 1
    assert np.isnan(ob.detmath_eval(ob.DETMATH_ASIN, np.array([1.0000001, -2.0, np.nan]))).all()


def test_atan2_within_two_ulp_of_glibc():
    rng = np.random.default_rng(2)
    y = np.concatenate([rng.normal(size=400000), rng.normal(size=20000) * 1e-9, np.zeros(8), rng.normal(size=20000)])
    x = np.concatenate([rng.normal(size=400000), rng.normal(size=20000), np.array([1, -1, 0, -0.0, 2, -2, 1e300, -1e300]),
                        rng.normal(size=20000) * 1e-9])
    got = ob.detmath_eval(ob.DETMATH_ATAN2, y, x)
    want = np.arctan2(y, x)
    d = _ulp_diff(got, want)
    # y / x is rounded before the arctangent: worst case ~1.5 ulp from the true value, i.e. up to 2 ulp from glibc's result
    assert d.max() <= 2 and (d > 1).mean() < 1e-4 and (d == 0).mean() > 0.8
    # the +-180 degree seam of the equirectangular projection: x < 0, y = +-tiny / +-0
    ys = np.array([0.0, -0.0, 1e-300, -1e-300, 1e-17, -1e-17])
    xs = -np.ones(6)
    got = ob.detmath_eval(ob.DETMATH_ATAN2, ys, xs)
    assert np.array_equal(got, np.arctan2(ys, xs)) and got[0] == np.pi and got[1] == -np.pi
    # special values
    inf = np.inf
    ys = np.array([inf, -inf, inf, -inf, 1.0, -1.0, 1.0, inf, 0.0, 3.0])
    xs = np.array([inf, inf, -inf, -inf, inf, inf, -inf, 1.0, 5.0, 0.0])
    assert np.array_equal(ob.detmath_eval(ob.DETMATH_ATAN2, ys, xs), np.arctan2(ys, xs))


def test_logf_equals_glibc_exhaustively():
    """landmark::predict_scale_level calls std::log(float) = glibc logf; ovs_det_logf restates glibc's algorithm (table + cubic in double).
    Every positive finite float (2^31 - 2^23 - 1 bit patterns) against this machine's libm: zero mismatches => the rule is pinned."""
    import ctypes as C
    f = ob.lib().ovo_detmath_logf_vs_libm
    f.restype = C.c_longlong
    f.argtypes = [C.c_uint32, C.c_uint32]
    assert f(1, 0x7F7FFFFF) == 0
    sp = ob.detmath_eval(ob.DETMATH_LOGF, np.array([0.0, -1.0, np.inf, np.nan, 1.0]))
    assert sp[0] == -np.inf and np.isnan(sp[1]) and sp[2] == np.inf and np.isnan(sp[3]) and sp[4] == 0.0


def test_deg2rad_single_multiply_is_exact():
    """k_describe converts the keypoint angle with ONE double multiplication by RN(pi / 180); upstream's expression (rule 11) is a double
    product and a double quotient. Equal after the rounding to float for EVERY float in [0, 360] -- fastAtan2's whole range."""
    assert ob.deg2rad_mismatches(0.0, 360.0) == 0


def test_sinf_cosf_equal_glibc_exhaustively():
    """ovs_det_sinf / ovs_det_cosf (glibc >= 2.28's sinf / cosf restated: the OVS_VARIANT_TRIG = 1 steering) against this machine's libm on
    EVERY float of [0, 6.3] -- all angles a keypoint can have, 1.09e9 values, ~12 s of C -- and hand-checked special values."""
    assert ob.trig_mismatches_vs_libm(0.0, 6.3) == 0
    f = ob.lib()
    assert f.ovo_det_cosf(0.0) == 1.0 and f.ovo_det_sinf(0.0) == 0.0
    assert f.ovo_det_sinf(np.float32(np.pi / 2)) == 1.0 and abs(f.ovo_det_cosf(np.float32(np.pi))) == 1.0
    # and they differ from util::cos / util::sin (the default steering) by the polynomial's ~1e-3, not by rounding
    a = np.float32(0.7)
    assert 1e-5 < abs(f.ovo_det_cosf(a) - f.ovo_util_cos(a)) < 2e-3


@pytest.mark.gpu
def test_device_and_host_agree_bit_for_bit():
    import ctypes as C
    from openvslam_amd import _lib
    L = _lib.lib()
    _lib.require_device()
    rng = np.random.default_rng(4)

    def dev(fn, a, b=None):
        a = np.ascontiguousarray(a, np.float64)
        out = np.zeros_like(a)
        pb = np.ascontiguousarray(b, np.float64) if b is not None else None
        _lib.check(L.ovs_detmath_eval(0, fn, a.ctypes.data, pb.ctypes.data if pb is not None else None, out.ctypes.data, a.size),
                   "ovs_detmath_eval")
        return out

    x = _samples(rng)
    for fn in (ob.DETMATH_ASIN, ob.DETMATH_ACOS):
        assert np.array_equal(dev(fn, x).view(np.uint64), ob.detmath_eval(fn, x).view(np.uint64))
    y = np.concatenate([rng.normal(size=500000), np.array([0.0, -0.0, 1e-300, -1e-300, 1e-17, -1e-17])])
    xx = np.concatenate([rng.normal(size=500000), -np.ones(6)])
    assert np.array_equal(dev(ob.DETMATH_ATAN2, y, xx).view(np.uint64), ob.detmath_eval(ob.DETMATH_ATAN2, y, xx).view(np.uint64))
    # logf: every float in three binades around 1 (2^23 mantissas x 3 exponents would be 25M; take every 7th) + wide range + 1.2^k +- 2 ulp
    m = (np.arange(0, 3 << 23, 7, dtype=np.uint32) + np.uint32(0x3F000000)).view(np.float32)
    k = (np.float64(1.2) ** np.arange(-40, 41)).astype(np.float32)
    k = np.concatenate([k, np.nextafter(k, np.float32(0)), np.nextafter(k, np.float32(1e30)),
                        np.nextafter(np.nextafter(k, np.float32(0)), np.float32(0)), np.nextafter(np.nextafter(k, np.float32(1e30)), np.float32(1e30))])
    xs = np.concatenate([m, np.exp(rng.uniform(-80, 80, 500000)).astype(np.float32), k]).astype(np.float64)
    assert np.array_equal(dev(ob.DETMATH_LOGF, xs).view(np.uint64), ob.detmath_eval(ob.DETMATH_LOGF, xs).view(np.uint64))
    # sinf / cosf (the `trig` variant's steering): every 5th float of [0, 2 pi] + the quadrant boundaries +- 3 ulp
    u = np.arange(0, int(np.float32(6.2832).view(np.uint32)), 5, dtype=np.uint32).view(np.float32)
    q = np.float32([np.pi / 4, np.pi / 2, 3 * np.pi / 4, np.pi, 5 * np.pi / 4, 3 * np.pi / 2, 7 * np.pi / 4, 2 * np.pi])
    nb = [q]
    for _ in range(3):
        nb += [np.nextafter(nb[-1], np.float32(0)), np.nextafter(nb[-1 if len(nb) == 1 else -2], np.float32(10))]
    ang = np.concatenate([u] + nb).astype(np.float64)
    for fn in (ob.DETMATH_SINF, ob.DETMATH_COSF):
        assert np.array_equal(dev(fn, ang).view(np.uint64), ob.detmath_eval(fn, ang).view(np.uint64))


def test_equirectangular_decisions_do_not_depend_on_the_shared_asin_atan2():
    """The kernels and the oracle share ovs_det_asin / ovs_det_atan2 (<= 1 / <= 2 ulp from glibc): parity between them on the equirectangular
    reprojection holds by construction, so it cannot show whether an upstream build (libm) would decide differently. This test can: the
    DECISIONS that hang on those values -- the pixel a bearing reprojects to, the grid cell of assign_keypoints_to_grid it falls in, the
    window test |u - u_kp| <= margin -- are evaluated through the shared functions and through numpy (glibc) for two million bearings, incl. the
    +-180 degree seam and the poles. They differ only where the libm value itself sits within a few ulp of the decision threshold."""
    rng = np.random.default_rng(5)
    n = 2_000_000
    p = rng.normal(size=(n, 3))
    p[:50000, 0] = rng.normal(size=50000) * 1e-6      # the seam (x ~ 0 behind the camera) and the forward direction
    p[:50000, 2] = -np.abs(p[:50000, 2])
    p[50000:100000, [0, 2]] *= 1e-5                   # the poles
    cols, rows = 3840.0, 1920.0
    norm = np.sqrt((p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + p[:, 2] * p[:, 2])
    s = p[:, 1] / norm
    s = np.clip(s, -1.0, 1.0)

    def project(atan2, asin):
        u = cols * (0.5 + atan2(p[:, 0], p[:, 2]) / (2 * np.pi))
        v = rows * (0.5 + asin(s) / np.pi)
        return u, v

    u_d, v_d = project(lambda y, x: ob.detmath_eval(ob.DETMATH_ATAN2, y, x), lambda a: ob.detmath_eval(ob.DETMATH_ASIN, a))
    u_l, v_l = project(np.arctan2, np.arcsin)
    assert np.abs(u_d - u_l).max() < 4e-12 * cols and np.abs(v_d - v_l).max() < 4e-12 * rows        # a few ulp of a pixel coordinate
    # grid cell of data::assign_keypoints_to_grid (64 x 48 cells over the image) and a +-15 px window around a keypoint at the image centre
    cw, ch = cols / 64, rows / 48
    cell_d = np.floor(u_d / cw) + 64 * np.floor(v_d / ch)
    cell_l = np.floor(u_l / cw) + 64 * np.floor(v_l / ch)
    flips = cell_d != cell_l
    win_d = (np.abs(u_d - cols / 2) <= 15.0) & (np.abs(v_d - rows / 2) <= 15.0)
    win_l = (np.abs(u_l - cols / 2) <= 15.0) & (np.abs(v_l - rows / 2) <= 15.0)
    flips |= win_d != win_l
    assert flips.sum() <= 2            # (measure-zero boundaries; a flip needs a pixel coordinate within ~1e-12 of a cell edge)
    if flips.any():                    # and where one happens, libm's own value is on the edge to within rounding
        edge = np.minimum(np.abs(u_l[flips] / cw - np.rint(u_l[flips] / cw)), np.abs(v_l[flips] / ch - np.rint(v_l[flips] / ch)))
        assert (edge < 1e-9).all() or (np.abs(np.abs(u_l[flips] - cols / 2) - 15.0) < 1e-9).all()


def test_asin_atan2_against_the_correctly_rounded_value():
    """The same functions against mpmath (50 digits, rounded to nearest double): an implementation-independent statement of their error --
    asin / acos within 1 ulp, atan2 within 2 ulp of the correctly rounded result -- so the glibc comparisons above are not the only pin."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1, 1, 6000), 1.0 - np.geomspace(1e-16, 1e-2, 500), -1.0 + np.geomspace(1e-16, 1e-2, 500),
                        rng.uniform(-1e-5, 1e-5, 500), np.array([0.5, -0.5, 0.975, 1.0, -1.0, 0.0])])
    want = np.array([float(mp.asin(mp.mpf(float(v)))) for v in x])
    assert _ulp_diff(ob.detmath_eval(ob.DETMATH_ASIN, x), want).max() <= 1
    want = np.array([float(mp.acos(mp.mpf(float(v)))) for v in x])
    assert _ulp_diff(ob.detmath_eval(ob.DETMATH_ACOS, x), want).max() <= 1
    y = np.concatenate([rng.normal(size=6000), rng.normal(size=1000) * 1e-9, rng.normal(size=1000)])
    xx = np.concatenate([rng.normal(size=6000), rng.normal(size=1000), rng.normal(size=1000) * 1e-9])
    want = np.array([float(mp.atan2(mp.mpf(float(a)), mp.mpf(float(b)))) for a, b in zip(y, xx)])
    d = _ulp_diff(ob.detmath_eval(ob.DETMATH_ATAN2, y, xx), want)
    assert d.max() <= 2 and (d == 0).mean() > 0.8
