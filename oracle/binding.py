"""ctypes binding of oracle/liboracle.so (CPU oracle, test infrastructure; PARITY UNPINNED -- see ovo_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class OrbParams(C.Structure):
    _fields_ = [("max_num_keypts", C.c_int32), ("scale_factor", C.c_float), ("num_levels", C.c_int32),
                ("ini_fast_thr", C.c_int32), ("min_fast_thr", C.c_int32)]


def _lib_name():
    # OVS_ORACLE_LIB=liboracle_asan.so: the sanitised build (make -C oracle asan), loaded by tests/test_sanitizers.py in a subprocess that
    # preloads the sanitizer runtime
    return os.environ.get("OVS_ORACLE_LIB", "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["asan"] if _lib_name() == "liboracle_asan.so" else []))


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, _lib_name())
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    u8p, i32p, f32p, u16p = (C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_uint16))
    vp = C.c_void_p
    L.ovo_orb_tables.argtypes = [C.POINTER(OrbParams), vp, vp, vp, vp, vp, vp]
    L.ovo_pyramid_sizes.argtypes = [C.POINTER(OrbParams), C.c_int, C.c_int, vp, vp]
    L.ovo_resize_linear_u8.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_int, C.c_int, C.c_size_t]
    L.ovo_fast9_16.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, vp, vp, vp, C.c_int]
    L.ovo_distribute_via_tree.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.ovo_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.ovo_fast_atan2.restype = C.c_float
    L.ovo_ic_angle.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp]
    L.ovo_ic_angle.restype = C.c_float
    L.ovo_gaussian_blur_7x7.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t]
    L.ovo_util_cos.argtypes = [C.c_float]
    L.ovo_util_cos.restype = C.c_float
    L.ovo_util_sin.argtypes = [C.c_float]
    L.ovo_util_sin.restype = C.c_float
    L.ovo_orb_descriptor.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_float, vp]
    L.ovo_orb_descriptor_v.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_float, vp, C.c_int]
    L.ovo_trig_mismatches_vs_libm.argtypes = [C.c_uint32, C.c_uint32]
    L.ovo_trig_mismatches_vs_libm.restype = C.c_long
    L.ovo_deg2rad_mismatches.argtypes = [C.c_uint32, C.c_uint32]
    L.ovo_deg2rad_mismatches.restype = C.c_long
    for fn in (L.ovo_det_sinf, L.ovo_det_cosf, L.ovo_util_cos, L.ovo_util_sin):
        fn.argtypes = [C.c_float]
        fn.restype = C.c_float
    L.ovo_orb_pattern.restype = C.POINTER(C.c_int8)
    L.ovo_orb_create.argtypes = [C.POINTER(OrbParams)]
    L.ovo_orb_create.restype = vp
    L.ovo_orb_destroy.argtypes = [vp]
    L.ovo_orb_set_threads.argtypes = [vp, C.c_int]
    L.ovo_orb_set_variant.argtypes = [vp, C.c_int, C.c_int]
    L.ovo_distribute_via_tree_v.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.ovo_gaussian_blur_7x7_v.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t, C.c_int]
    L.ovo_orb_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_size_t, vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.ovo_orb_level_size.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ovo_orb_level_image.argtypes = [vp, C.c_int]
    L.ovo_orb_level_image.restype = vp
    L.ovo_orb_level_blurred.argtypes = [vp, C.c_int]
    L.ovo_orb_level_blurred.restype = vp
    L.ovo_orb_level_num_candidates.argtypes = [vp, C.c_int]
    L.ovo_orb_level_candidates.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
    L.ovo_orb_level_num_keypts.argtypes = [vp, C.c_int]
    L.ovo_descriptor_distance_32.argtypes = [vp, vp]
    L.ovo_descriptor_distance_32.restype = C.c_uint32
    L.ovo_robust_brute_force_match.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, C.c_float, vp, C.c_int]
    L.ovo_hamming_best2.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]
    for name, at in _OPTIONAL.items():
        if hasattr(L, name):
            getattr(L, name).argtypes = at[0]
            if at[1] is not None:
                getattr(L, name).restype = at[1]
    _LIB = L
    return L


_OPTIONAL = {"ovo_ba_linearize": ([C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_double,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int)}


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


DETMATH_LOGF, DETMATH_ASIN, DETMATH_ACOS, DETMATH_ATAN2, DETMATH_SINF, DETMATH_COSF = 0, 1, 2, 3, 4, 5


def detmath_eval(fn, a, b=None):
    """include/ovs_detmath.h on the host (the op sequence the kernels share)."""
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64) if b is not None else None
    out = np.zeros(a.shape, np.float64)
    f = lib().ovo_detmath_eval
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    assert f(fn, _p(a), _p(b), _p(out), a.size) == 0
    return out


def make_params(max_num_keypts=2000, scale_factor=1.2, num_levels=8, ini_fast_thr=20, min_fast_thr=7):
    return OrbParams(max_num_keypts, scale_factor, num_levels, ini_fast_thr, min_fast_thr)


def orb_tables(params):
    L = params.num_levels
    sf, isf, ls, ils = (np.zeros(L, np.float32) for _ in range(4))
    npl = np.zeros(L, np.int32)
    um = np.zeros(16, np.int32)
    assert lib().ovo_orb_tables(C.byref(params), _p(sf), _p(isf), _p(ls), _p(ils), _p(npl), _p(um)) == 0
    return dict(scale_factors=sf, inv_scale_factors=isf, level_sigma_sq=ls, inv_level_sigma_sq=ils, num_keypts_per_level=npl,
                u_max=um)


def pyramid_sizes(params, rows, cols):
    lr = np.zeros(params.num_levels, np.int32)
    lc = np.zeros(params.num_levels, np.int32)
    assert lib().ovo_pyramid_sizes(C.byref(params), rows, cols, _p(lr), _p(lc)) == 0
    return lr, lc


def resize_linear(src, drows, dcols):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((drows, dcols), np.uint8)
    assert lib().ovo_resize_linear_u8(_p(src), src.shape[0], src.shape[1], src.strides[0], _p(dst), drows, dcols, dcols) == 0
    return dst


def fast9_16(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().ovo_fast9_16(_p(img), img.shape[0], img.shape[1], img.strides[0], threshold, int(nonmax), _p(xs), _p(ys), _p(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def distribute_via_tree(xs, ys, responses, min_x, max_x, min_y, max_y, num_keypts, switch_factor=3, tie_earlier_first=False):
    """switch_factor / tie_earlier_first: ORACLE_SPEC rules 6 / 7 as run-time variants (defaults = the rules as fixed)."""
    xs = np.ascontiguousarray(xs, np.float32)
    ys = np.ascontiguousarray(ys, np.float32)
    rs = np.ascontiguousarray(responses, np.float32)
    out = np.zeros(max(len(xs), 1), np.int32)
    n = lib().ovo_distribute_via_tree_v(_p(xs), _p(ys), _p(rs), len(xs), min_x, max_x, min_y, max_y, num_keypts, int(switch_factor),
                                        int(bool(tie_earlier_first)), _p(out), len(out))
    return out[:n].copy()


def gaussian_blur(img, taps_variant=0):
    """taps_variant: ORACLE_SPEC rule 10 as a run-time variant (0 = 18 34 48 56 ..., 1 = 18 34 49 55 ... saturating)."""
    img = np.ascontiguousarray(img, np.uint8)
    dst = np.zeros_like(img)
    assert lib().ovo_gaussian_blur_7x7_v(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(dst), dst.strides[0], int(taps_variant)) == 0
    return dst


def ic_angle(img, x, y, u_max):
    img = np.ascontiguousarray(img, np.uint8)
    um = np.ascontiguousarray(u_max, np.int32)
    return float(lib().ovo_ic_angle(_p(img), img.strides[0], x, y, _p(um)))


def orb_descriptor(blurred, x, y, angle_deg, trig_variant=0):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    d = np.zeros(32, np.uint8)
    assert lib().ovo_orb_descriptor_v(_p(blurred), blurred.strides[0], x, y, angle_deg, _p(d), int(trig_variant)) == 0
    return d


def deg2rad_mismatches(lo, hi):
    """ovs_det_deg2rad (one f64 multiply, what k_describe evaluates) against `(float)((double)a * M_PI / 180.0)` on every float of [lo, hi]."""
    lo_b, hi_b = (int(np.float32(v).view(np.uint32)) for v in (lo, hi))
    return int(lib().ovo_deg2rad_mismatches(lo_b, hi_b))


def trig_mismatches_vs_libm(lo, hi):
    """ovs_det_sinf / ovs_det_cosf against this machine's libm on every float in [lo, hi] (C loop): number of differing values."""
    lo_b, hi_b = (int(np.float32(v).view(np.uint32)) for v in (lo, hi))
    return int(lib().ovo_trig_mismatches_vs_libm(lo_b, hi_b))


def orb_pattern():
    p = lib().ovo_orb_pattern()
    return np.ctypeslib.as_array(p, shape=(256, 4)).copy()


class OrbExtractor:
    """Oracle restatement of feature::orb_extractor (expected: src/openvslam/feature/orb_extractor.h)."""

    def __init__(self, params=None, threads=1):
        self.params = params or make_params()
        self._h = lib().ovo_orb_create(C.byref(self.params))
        if not self._h:
            raise ValueError("bad orb_params")
        lib().ovo_orb_set_threads(self._h, threads)
        self.threads = threads

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ovo_orb_destroy(self._h)
            self._h = None

    def set_variant(self, which, value):
        """ORACLE_SPEC rules 6 / 7 / 10 as run-time variants (ovo_orb_set_variant): same names and values as feature.orb_extractor.set_variant."""
        idx = {"tree_switch_factor": 0, "tree_tie_order": 1, "blur_taps": 2, "trig": 3}[which]
        assert lib().ovo_orb_set_variant(self._h, idx, int(value)) == 0, (which, value)

    def extract(self, img, mask=None):
        img = np.ascontiguousarray(img, np.uint8)
        cap = 2 * self.params.max_num_keypts + 260 * self.params.num_levels + 64   # a level may return 4 nodes per root patch (<= 64 patches); 2 N under tree_switch_factor = 1
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
        rc = lib().ovo_orb_extract(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0], _p(mask),
                                   mask.strides[0] if mask is not None else 0, _p(kps), _p(desc), cap, C.byref(n))
        assert rc == 0, rc
        return kps[:n.value].copy(), desc[:n.value].copy()

    def level_image(self, level):
        r, c = C.c_int(), C.c_int()
        assert lib().ovo_orb_level_size(self._h, level, C.byref(r), C.byref(c)) == 0
        ptr = lib().ovo_orb_level_image(self._h, level)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(r.value, c.value)).copy()

    def level_blurred(self, level):
        r, c = C.c_int(), C.c_int()
        assert lib().ovo_orb_level_size(self._h, level, C.byref(r), C.byref(c)) == 0
        ptr = lib().ovo_orb_level_blurred(self._h, level)
        if not ptr:
            return None
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(r.value, c.value)).copy()

    def level_candidates(self, level):
        n = lib().ovo_orb_level_num_candidates(self._h, level)
        xs, ys, sc = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        lib().ovo_orb_level_candidates(self._h, level, _p(xs), _p(ys), _p(sc), n)
        return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()

    def level_num_keypts(self, level):
        return lib().ovo_orb_level_num_keypts(self._h, level)


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return int(lib().ovo_descriptor_distance_32(_p(a), _p(b)))


def robust_brute_force_match(desc_frm, desc_kf, kf_valid=None, lowe_ratio=0.8, frm_valid=None):
    desc_frm = np.ascontiguousarray(desc_frm, np.uint8)
    desc_kf = np.ascontiguousarray(desc_kf, np.uint8)
    if kf_valid is not None:
        kf_valid = np.ascontiguousarray(kf_valid, np.uint8)
    pairs = np.zeros((max(len(desc_kf), 1), 2), np.int32)
    if frm_valid is not None:
        frm_valid = np.ascontiguousarray(frm_valid, np.uint8)
    n = lib().ovo_robust_brute_force_match(_p(desc_frm), len(desc_frm), _p(frm_valid), _p(desc_kf), len(desc_kf), _p(kf_valid),
                                           lowe_ratio, _p(pairs), len(pairs))
    return pairs[:n].copy()


def hamming_best2(q, t, t_valid=None):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    if t_valid is not None:
        t_valid = np.ascontiguousarray(t_valid, np.uint8)
    bi = np.zeros(len(q), np.int32)
    b = np.zeros(len(q), np.uint16)
    s = np.zeros(len(q), np.uint16)
    lib().ovo_hamming_best2(_p(q), len(q), _p(t), len(t), _p(t_valid), _p(bi), _p(b), _p(s))
    return bi, b, s


class GridParams(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float), ("cols", C.c_int32),
                ("rows", C.c_int32)]


def grid_params(cols, rows, num_grid_cols=64, num_grid_rows=48, min_x=0.0, min_y=0.0):
    return GridParams(min_x, min_y, float(cols), float(rows), num_grid_cols, num_grid_rows)


def _soa(kps):
    k = np.ascontiguousarray(kps, KP_DTYPE)
    return (np.ascontiguousarray(k["x"]), np.ascontiguousarray(k["y"]), np.ascontiguousarray(k["octave"]), np.ascontiguousarray(k["angle"]))


def assign_keypoints_to_grid(gp, kps):
    xs, ys, _, _ = _soa(kps)
    start = np.zeros(gp.cols * gp.rows + 1, np.int32)
    items = np.zeros(max(len(xs), 1), np.int32)
    L = lib()
    L.ovo_assign_keypoints_to_grid.restype = C.c_int
    n = L.ovo_assign_keypoints_to_grid(C.byref(gp), _p(xs), _p(ys), len(xs), _p(start), _p(items))
    return start, items[:n].copy()


def get_keypoints_in_cell(gp, kps, ref_x, ref_y, margin, min_level=-1, max_level=-1):
    xs, ys, oc, _ = _soa(kps)
    out = np.zeros(max(len(xs), 1), np.int32)
    n = lib().ovo_get_keypoints_in_cell(C.byref(gp), _p(xs), _p(ys), _p(oc), len(xs), C.c_float(ref_x), C.c_float(ref_y), C.c_float(margin),
                                        int(min_level), int(max_level), _p(out), len(out))
    return out[:n].copy()


def match_set_variant(which, value):
    """ovo_match_set_variant, process-wide (ORACLE_SPEC rule 17): "angle_keep_rule" (0 top-3 | 1 top-3 with ORB-SLAM2's 0.1 x max rule),
    "angle_tie_order" (0 lower of two equally full bins first | 1 higher first)."""
    assert lib().ovo_match_set_variant({"angle_keep_rule": 0, "angle_tie_order": 1}[which], int(value)) == 0


def angle_checker_invalid(delta_angles):
    d = np.ascontiguousarray(delta_angles, np.float32)
    inv = np.zeros(max(len(d), 1), np.uint8)
    lib().ovo_angle_checker_invalid(_p(d), len(d), _p(inv))
    return inv[:len(d)].astype(bool)


def projection_match_frame_and_landmarks(gp, frm_kps, frm_desc, scale_factors, lm_reproj, lm_level, lm_desc, margin=5.0, lowe_ratio=0.6,
                                         frm_stereo_x_right=None, frm_occupied=None, lm_x_right=None, lm_valid=None):
    xs, ys, oc, _ = _soa(frm_kps)
    d = np.ascontiguousarray(frm_desc, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    xy = np.ascontiguousarray(lm_reproj, np.float32).reshape(-1, 2)
    lx, ly = np.ascontiguousarray(xy[:, 0]), np.ascontiguousarray(xy[:, 1])
    lv = np.ascontiguousarray(lm_level, np.int32)
    ld = np.ascontiguousarray(lm_desc, np.uint8).reshape(-1, 32)
    xr = None if frm_stereo_x_right is None else np.ascontiguousarray(frm_stereo_x_right, np.float32)
    occ = np.zeros(len(xs), np.uint8) if frm_occupied is None else np.ascontiguousarray(frm_occupied, np.uint8)
    lxr = None if lm_x_right is None else np.ascontiguousarray(lm_x_right, np.float32)
    val = None if lm_valid is None else np.ascontiguousarray(lm_valid, np.uint8)
    assigned = np.full(max(len(xy), 1), -1, np.int32)
    n = lib().ovo_projection_match_frame_and_landmarks(C.byref(gp), _p(xs), _p(ys), _p(oc), _p(xr), _p(d), _p(occ), len(xs), _p(lx), _p(ly),
                                                       _p(lxr), _p(lv), _p(ld), _p(val), len(xy), _p(sf), C.c_float(margin),
                                                       C.c_float(lowe_ratio), _p(assigned))
    return assigned[:len(xy)].copy(), n


def area_match_in_consistent_area(gp, kps_1, desc_1, kps_2, desc_2, prev_matched_pts, margin=10, lowe_ratio=0.9, check_orientation=True):
    _, _, oc1, an1 = _soa(kps_1)
    xs2, ys2, oc2, an2 = _soa(kps_2)
    d1 = np.ascontiguousarray(desc_1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(desc_2, np.uint8).reshape(-1, 32)
    assert prev_matched_pts.dtype == np.float32 and prev_matched_pts.flags.c_contiguous
    matched = np.full(max(len(oc1), 1), -1, np.int32)
    n = lib().ovo_area_match_in_consistent_area(C.byref(gp), _p(oc1), _p(an1), _p(d1), len(oc1), _p(xs2), _p(ys2), _p(oc2), _p(an2), _p(d2),
                                                len(xs2), _p(prev_matched_pts), _p(matched), int(margin), C.c_float(lowe_ratio),
                                                int(check_orientation))
    return n, matched[:len(oc1)].copy()


def _flatten_bow(feat_vec):
    ids = np.array(sorted(feat_vec), np.int32)
    start = np.zeros(len(ids) + 1, np.int32)
    items = []
    for k, i in enumerate(ids):
        items.extend(feat_vec[int(i)])
        start[k + 1] = len(items)
    return ids, start, np.array(items, np.int32)


def bow_match_frame_and_keyframe(kf_kps, kf_desc, kf_feat_vec, frm_kps, frm_desc, frm_feat_vec, lowe_ratio=0.6, check_orientation=True,
                                 kf_has_landmark=None):
    _, _, _, ka = _soa(kf_kps)
    _, _, _, fa = _soa(frm_kps)
    kd = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32)
    fd = np.ascontiguousarray(frm_desc, np.uint8).reshape(-1, 32)
    v = None if kf_has_landmark is None else np.ascontiguousarray(kf_has_landmark, np.uint8)
    ki, ks, kit = _flatten_bow(kf_feat_vec)
    fi, fs, fit = _flatten_bow(frm_feat_vec)
    out = np.full(max(len(fa), 1), -1, np.int32)
    n = lib().ovo_bow_match_frame_and_keyframe(_p(kd), _p(ka), _p(v), len(ka), _p(ki), _p(ks), _p(kit), len(ki), _p(fd), _p(fa), len(fa),
                                               _p(fi), _p(fs), _p(fit), len(fi), C.c_float(lowe_ratio), int(check_orientation), _p(out))
    return n, out[:len(fa)].copy()


class Camera(C.Structure):
    _fields_ = [("model", C.c_int32), ("setup", C.c_int32), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("focal_x_baseline", C.c_double), ("true_baseline", C.c_double), ("cols", C.c_int32), ("rows", C.c_int32)]


def _pose12(pose_cw):
    T = np.asarray(pose_cw, np.float64)
    return np.ascontiguousarray(np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]))


def reproject_to_image(cam, gp, pose_cw, pos_w):
    out = np.zeros(2)
    xr = C.c_float()
    p = _pose12(pose_cw)
    x = np.ascontiguousarray(pos_w, np.float64)
    ok = lib().ovo_reproject_to_image(C.byref(cam), C.byref(gp), _p(p), _p(x), _p(out), C.byref(xr))
    return bool(ok), out, xr.value


def projection_match_current_and_last_frames(cam, gp, curr_kps, curr_desc, pose_cw_curr, last_kps, last_pos_w, last_lm_desc, pose_cw_last,
                                             scale_factors, margin, check_orientation=True, curr_stereo_x_right=None, curr_occupied=None,
                                             last_valid=None):
    xs, ys, oc, an = _soa(curr_kps)
    _, _, loc, lan = _soa(last_kps)
    cd = np.ascontiguousarray(curr_desc, np.uint8).reshape(-1, 32)
    pw = np.ascontiguousarray(last_pos_w, np.float64).reshape(-1, 3)
    ld = np.ascontiguousarray(last_lm_desc, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    xr = None if curr_stereo_x_right is None else np.ascontiguousarray(curr_stereo_x_right, np.float32)
    occ = None if curr_occupied is None else np.ascontiguousarray(curr_occupied, np.uint8)
    val = None if last_valid is None else np.ascontiguousarray(last_valid, np.uint8)
    pc, pl = _pose12(pose_cw_curr), _pose12(pose_cw_last)
    assigned = np.full(max(len(loc), 1), -1, np.int32)
    n = lib().ovo_projection_match_current_and_last_frames(C.byref(cam), C.byref(gp), _p(xs), _p(ys), _p(oc), _p(an), _p(xr), _p(cd), _p(occ),
                                                           len(xs), _p(pc), _p(loc), _p(lan), _p(pw), _p(ld), _p(val), len(loc), _p(pl),
                                                           _p(sf), len(sf), C.c_float(margin), int(check_orientation), _p(assigned))
    return assigned[:len(loc)].copy(), n


def bow_match_keyframes(kps_1, desc_1, feat_vec_1, kps_2, desc_2, feat_vec_2, lowe_ratio=0.6, check_orientation=True, has_lm_1=None,
                        has_lm_2=None):
    _, _, _, a1 = _soa(kps_1)
    _, _, _, a2 = _soa(kps_2)
    d1 = np.ascontiguousarray(desc_1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(desc_2, np.uint8).reshape(-1, 32)
    v1 = None if has_lm_1 is None else np.ascontiguousarray(has_lm_1, np.uint8)
    v2 = None if has_lm_2 is None else np.ascontiguousarray(has_lm_2, np.uint8)
    i1, s1, t1 = _flatten_bow(feat_vec_1)
    i2, s2, t2 = _flatten_bow(feat_vec_2)
    out = np.full(max(len(a1), 1), -1, np.int32)
    n = lib().ovo_bow_match_keyframes(_p(d1), _p(a1), _p(v1), len(a1), _p(i1), _p(s1), _p(t1), len(i1), _p(d2), _p(a2), _p(v2), len(a2), _p(i2),
                                      _p(s2), _p(t2), len(i2), C.c_float(lowe_ratio), int(check_orientation), _p(out))
    return n, out[:len(a1)].copy()


def projection_match_frame_and_keyframe(cam, gp, curr_kps, curr_desc, pose_cw_curr, kf_kps, kf_pos_w, kf_dist_min_max, kf_lm_desc,
                                        scale_factors, log_scale_factor, margin, hamm_dist_thr, check_orientation=True, curr_occupied=None,
                                        kf_valid=None):
    xs, ys, oc, an = _soa(curr_kps)
    _, _, _, kan = _soa(kf_kps)
    cd = np.ascontiguousarray(curr_desc, np.uint8).reshape(-1, 32)
    pw = np.ascontiguousarray(kf_pos_w, np.float64).reshape(-1, 3)
    dm = np.ascontiguousarray(kf_dist_min_max, np.float32).reshape(-1, 2)
    ld = np.ascontiguousarray(kf_lm_desc, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    occ = None if curr_occupied is None else np.ascontiguousarray(curr_occupied, np.uint8)
    val = None if kf_valid is None else np.ascontiguousarray(kf_valid, np.uint8)
    assigned = np.full(max(len(kan), 1), -1, np.int32)
    n = lib().ovo_projection_match_frame_and_keyframe(C.byref(cam), C.byref(gp), _p(xs), _p(ys), _p(oc), _p(an), _p(cd), _p(occ), len(xs),
                                                      _p(_pose12(pose_cw_curr)), _p(kan), _p(pw), _p(dm), _p(ld), _p(val), len(kan), _p(sf),
                                                      len(sf), C.c_float(log_scale_factor), C.c_float(margin), C.c_uint(hamm_dist_thr),
                                                      int(check_orientation), _p(assigned))
    return assigned[:len(kan)].copy(), n


def robust_match_for_triangulation(kps_1, desc_1, feat_vec_1, bearings_1, kps_2, desc_2, feat_vec_2, bearings_2, E_12, epipole_in_2,
                                   scale_factors, check_orientation=True, has_lm_1=None, has_lm_2=None, x_right_1=None, x_right_2=None):
    _, _, o1, a1 = _soa(kps_1)
    _, _, _, a2 = _soa(kps_2)
    d1 = np.ascontiguousarray(desc_1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(desc_2, np.uint8).reshape(-1, 32)
    b1 = np.ascontiguousarray(bearings_1, np.float64).reshape(-1, 3)
    b2 = np.ascontiguousarray(bearings_2, np.float64).reshape(-1, 3)
    h1 = None if has_lm_1 is None else np.ascontiguousarray(has_lm_1, np.uint8)
    h2 = None if has_lm_2 is None else np.ascontiguousarray(has_lm_2, np.uint8)
    x1 = None if x_right_1 is None else np.ascontiguousarray(x_right_1, np.float32)
    x2 = None if x_right_2 is None else np.ascontiguousarray(x_right_2, np.float32)
    E = np.ascontiguousarray(E_12, np.float64).reshape(9)
    ep = np.ascontiguousarray(epipole_in_2, np.float64).reshape(3)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    i1, s1, t1 = _flatten_bow(feat_vec_1)
    i2, s2, t2 = _flatten_bow(feat_vec_2)
    out = np.full(max(len(a1), 1), -1, np.int32)
    n = lib().ovo_robust_match_for_triangulation(_p(d1), _p(a1), _p(o1), _p(h1), _p(x1), _p(b1), len(a1), _p(i1), _p(s1), _p(t1), len(i1),
                                                 _p(d2), _p(a2), _p(h2), _p(x2), _p(b2), len(a2), _p(i2), _p(s2), _p(t2), len(i2), _p(E), _p(ep),
                                                 _p(sf), int(check_orientation), _p(out))
    return n, out[:len(a1)].copy()


def fuse_replace_duplication(cam, gp, kf_kps, kf_desc, pose_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc, scale_factors,
                             inv_level_sigma_sq, log_scale_factor, margin=3.0, kf_stereo_x_right=None, lm_valid=None):
    xs, ys, oc, _ = _soa(kf_kps)
    d = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32)
    pw = np.ascontiguousarray(lm_pos_w, np.float64).reshape(-1, 3)
    dm = np.ascontiguousarray(lm_dist_min_max, np.float32).reshape(-1, 2)
    nr = np.ascontiguousarray(lm_normal, np.float64).reshape(-1, 3)
    ld = np.ascontiguousarray(lm_desc, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    ils = np.ascontiguousarray(inv_level_sigma_sq, np.float32)
    xr = None if kf_stereo_x_right is None else np.ascontiguousarray(kf_stereo_x_right, np.float32)
    val = None if lm_valid is None else np.ascontiguousarray(lm_valid, np.uint8)
    best = np.full(max(len(pw), 1), -1, np.int32)
    n = lib().ovo_fuse_replace_duplication(C.byref(cam), C.byref(gp), _p(xs), _p(ys), _p(oc), _p(xr), _p(d), len(xs), _p(_pose12(pose_cw)),
                                           _p(pw), _p(dm), _p(nr), _p(ld), _p(val), len(pw), _p(sf), _p(ils), len(sf),
                                           C.c_float(log_scale_factor), C.c_float(margin), _p(best))
    return best[:len(pw)].copy(), n


def stereo_compute(ox_left, ox_right, kps_left, desc_left, kps_right, desc_right, focal_x_baseline, true_baseline, outlier_factor_21=False,
                   parabola_double=False):
    """stereo::compute on the pyramids of two OrbExtractor instances (their last extract). Returns (stereo_x_right, depths, n_valid).
    outlier_factor_21 / parabola_double: ORACLE_SPEC rule 20's alternatives (ovs_stereo_set_variant on the HIP side)."""
    L = lib()
    tabs = orb_tables(ox_left.params)
    nl = ox_left.params.num_levels
    imgs_l = [np.ascontiguousarray(ox_left.level_image(l)) for l in range(nl)]
    imgs_r = [np.ascontiguousarray(ox_right.level_image(l)) for l in range(nl)]
    pl = (C.c_void_p * nl)(*[a.ctypes.data for a in imgs_l])
    pr = (C.c_void_p * nl)(*[a.ctypes.data for a in imgs_r])
    rows = np.array([a.shape[0] for a in imgs_l], np.int32)
    cols = np.array([a.shape[1] for a in imgs_l], np.int32)
    sl = (C.c_size_t * nl)(*[a.shape[1] for a in imgs_l])
    sr = (C.c_size_t * nl)(*[a.shape[1] for a in imgs_r])
    kl = np.ascontiguousarray(kps_left, KP_DTYPE)
    kr = np.ascontiguousarray(kps_right, KP_DTYPE)
    dl = np.ascontiguousarray(desc_left, np.uint8).reshape(-1, 32)
    dr = np.ascontiguousarray(desc_right, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(tabs["scale_factors"], np.float32)
    isf = np.ascontiguousarray(tabs["inv_scale_factors"], np.float32)
    xr = np.full(max(len(kl), 1), -1, np.float32)
    dp = np.full(max(len(kl), 1), -1, np.float32)
    n = L.ovo_stereo_compute_v(pl, pr, _p(rows), _p(cols), sl, sr, nl, _p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), _p(sf), _p(isf),
                               C.c_float(focal_x_baseline), C.c_float(true_baseline), _p(xr), _p(dp),
                               C.c_int((1 if outlier_factor_21 else 0) | (2 if parabola_double else 0)))
    return xr[:len(kl)].copy(), dp[:len(kl)].copy(), n


BA_EDGE_DTYPE = np.dtype([("pose_idx", "<i4"), ("point_idx", "<i4"), ("obs_x", "<f8"), ("obs_y", "<f8"), ("inv_sigma_sq", "<f8")])


def ba_linearize(poses, pose_fixed, points, edges, cam, huber_delta):
    """Oracle restatement of B1-B3 (ovo_ba.cc). Returns dict(Hpp, bp, Hll, bl, Hpl, chi2)."""
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    fixed = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
    n_pose, n_pt, n_edge = len(poses), len(points), len(edges)
    out = dict(Hpp=np.zeros((n_pose, 6, 6)), bp=np.zeros((n_pose, 6)), Hll=np.zeros((n_pt, 3, 3)), bl=np.zeros((n_pt, 3)),
               Hpl=np.zeros((max(n_edge, 1), 6, 3)), chi2=np.zeros(2))
    c = np.array(cam, np.float64)
    rc = lib().ovo_ba_linearize(_p(poses), _p(fixed), n_pose, _p(points), n_pt, _p(edges), n_edge, _p(c), float(huber_delta), _p(out["Hpp"]),
                                _p(out["bp"]), _p(out["Hll"]), _p(out["bl"]), _p(out["Hpl"]), _p(out["chi2"]))
    assert rc == 0, rc
    out["Hpl"] = out["Hpl"][:n_edge]
    return out


BA_EDGE_STEREO_DTYPE = np.dtype([("pose_idx", "<i4"), ("point_idx", "<i4"), ("obs_x", "<f8"), ("obs_y", "<f8"), ("obs_x_right", "<f8"),
                                 ("inv_sigma_sq", "<f8")])


def ba_linearize_stereo(poses, pose_fixed, points, edges, cam, focal_x_baseline, huber_delta, accumulate_into=None):
    """Oracle restatement of the stereo reprojection edge (ovo_ba.cc). Returns dict(Hpp, bp, Hll, bl, Hpl, chi2)."""
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    edges = np.ascontiguousarray(edges, BA_EDGE_STEREO_DTYPE)
    fixed = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
    n_pose, n_pt, n_edge = len(poses), len(points), len(edges)
    if accumulate_into is None:
        out = dict(Hpp=np.zeros((n_pose, 6, 6)), bp=np.zeros((n_pose, 6)), Hll=np.zeros((n_pt, 3, 3)), bl=np.zeros((n_pt, 3)), chi2=np.zeros(2))
    else:
        out = {k: accumulate_into[k].copy() for k in ("Hpp", "bp", "Hll", "bl", "chi2")}
    out["Hpl"] = np.zeros((max(n_edge, 1), 6, 3))
    c = np.array(cam, np.float64)
    L = lib()
    L.ovo_ba_linearize_stereo.restype = C.c_int
    rc = L.ovo_ba_linearize_stereo(_p(poses), _p(fixed), n_pose, _p(points), n_pt, _p(edges), n_edge, _p(c), C.c_double(focal_x_baseline),
                                   C.c_double(huber_delta), int(accumulate_into is not None), _p(out["Hpp"]), _p(out["bp"]), _p(out["Hll"]),
                                   _p(out["bl"]), _p(out["Hpl"]), _p(out["chi2"]))
    assert rc == 0, rc
    out["Hpl"] = out["Hpl"][:n_edge]
    return out


POSE_OBS_DTYPE = np.dtype([("pos_w", "<f8", (3,)), ("obs_x", "<f8"), ("obs_y", "<f8"), ("obs_x_right", "<f8"), ("inv_sigma_sq", "<f8"),
                           ("is_stereo", "<i4"), ("pad", "<i4")])
assert POSE_OBS_DTYPE.itemsize == 64


def pose_set_variant(which, value):
    """ovo_pose_set_variant: "reset_each_round" (0 | 1), process-wide (ORACLE_SPEC rule 25 (iv))."""
    assert lib().ovo_pose_set_variant({"reset_each_round": 0}[which], int(value)) == 0


def pose_optimize(pose_cw, obs, cam, focal_x_baseline=0.0, setup_type=None):
    """optimize::pose_optimizer::optimize (ovo_pose.cc). Returns (pose_cw 3x4, outlier flags, num_valid)."""
    o = np.ascontiguousarray(obs, POSE_OBS_DTYPE)
    pin = _pose12(pose_cw)
    pout = np.zeros(12)
    out = np.zeros(max(len(o), 1), np.uint8)
    nv = C.c_int()
    c = np.array(cam, np.float64)
    if setup_type is None:
        setup_type = 1 if focal_x_baseline != 0.0 else 0
    rc = lib().ovo_pose_optimize(_p(pin), _p(o), len(o), _p(c), C.c_double(focal_x_baseline), C.c_int(int(setup_type)), _p(pout), _p(out),
                                 C.byref(nv))
    assert rc == 0
    return np.concatenate([pout[:9].reshape(3, 3), pout[9:, None]], 1), out[:len(o)].astype(bool), nv.value


def ba_edge_chi2_equirect(poses, points, edges, cols, rows):
    """chi2 of every equirectangular edge at (poses n x 7, points) -- ovo_ba_edge_chi2_equirect."""
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    out = np.zeros(max(len(edges), 1))
    L = lib()
    L.ovo_ba_edge_chi2_equirect.restype = C.c_int
    rc = L.ovo_ba_edge_chi2_equirect(_p(poses), len(poses), _p(points), len(points), _p(edges), len(edges), int(cols), int(rows), _p(out))
    assert rc == 0, rc
    return out[:len(edges)]


def pose_optimize_equirect(pose_cw, obs, cols, rows):
    """pose_optimizer::optimize for an equirectangular frame (ovo_pose.cc: equirectangular_pose_opt_edge). Returns (pose 3x4, flags, num_valid)."""
    o = np.ascontiguousarray(obs, POSE_OBS_DTYPE)
    pin = _pose12(pose_cw)
    pout = np.zeros(12)
    out = np.zeros(max(len(o), 1), np.uint8)
    nv = C.c_int()
    rc = lib().ovo_pose_optimize_equirect(_p(pin), _p(o), len(o), C.c_int(int(cols)), C.c_int(int(rows)), _p(pout), _p(out), C.byref(nv))
    assert rc == 0
    return np.concatenate([pout[:9].reshape(3, 3), pout[9:, None]], 1), out[:len(o)].astype(bool), nv.value


def ba_linearize_equirect(poses, pose_fixed, points, edges, cols, rows, huber_delta):
    """Oracle restatement of the equirectangular reprojection edge (ovo_ba.cc). Returns dict(Hpp, bp, Hll, bl, Hpl, chi2)."""
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    fixed = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
    n_pose, n_pt, n_edge = len(poses), len(points), len(edges)
    out = dict(Hpp=np.zeros((n_pose, 6, 6)), bp=np.zeros((n_pose, 6)), Hll=np.zeros((n_pt, 3, 3)), bl=np.zeros((n_pt, 3)),
               Hpl=np.zeros((max(n_edge, 1), 6, 3)), chi2=np.zeros(2))
    L = lib()
    L.ovo_ba_linearize_equirect.restype = C.c_int
    rc = L.ovo_ba_linearize_equirect(_p(poses), _p(fixed), n_pose, _p(points), n_pt, _p(edges), n_edge, int(cols), int(rows),
                                     C.c_double(huber_delta), _p(out["Hpp"]), _p(out["bp"]), _p(out["Hll"]), _p(out["bl"]), _p(out["Hpl"]),
                                     _p(out["chi2"]))
    assert rc == 0, rc
    out["Hpl"] = out["Hpl"][:n_edge]
    return out


def fuse_detect_duplication(cam, gp, kf_kps, kf_desc, sim3_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc, scale_factors,
                            log_scale_factor, margin, lm_valid=None):
    xs, ys, oc, _ = _soa(kf_kps)
    d = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32)
    pw = np.ascontiguousarray(lm_pos_w, np.float64).reshape(-1, 3)
    dm = np.ascontiguousarray(lm_dist_min_max, np.float32).reshape(-1, 2)
    nr = np.ascontiguousarray(lm_normal, np.float64).reshape(-1, 3)
    ld = np.ascontiguousarray(lm_desc, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    val = None if lm_valid is None else np.ascontiguousarray(lm_valid, np.uint8)
    best = np.full(max(len(pw), 1), -1, np.int32)
    n = lib().ovo_fuse_detect_duplication(C.byref(cam), C.byref(gp), _p(xs), _p(ys), _p(oc), _p(d), len(xs), _p(_pose12(sim3_cw)), _p(pw),
                                          _p(dm), _p(nr), _p(ld), _p(val), len(pw), _p(sf), len(sf), C.c_float(log_scale_factor),
                                          C.c_float(margin), _p(best))
    return best[:len(pw)].copy(), n


def projection_match_by_sim3_transform(cam, gp, kf_kps, kf_desc, sim3_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc, scale_factors,
                                       log_scale_factor, margin, kf_occupied=None, lm_valid=None):
    xs, ys, oc, _ = _soa(kf_kps)
    d = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32)
    pw = np.ascontiguousarray(lm_pos_w, np.float64).reshape(-1, 3)
    dm = np.ascontiguousarray(lm_dist_min_max, np.float32).reshape(-1, 2)
    nr = np.ascontiguousarray(lm_normal, np.float64).reshape(-1, 3)
    ld = np.ascontiguousarray(lm_desc, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    occ = None if kf_occupied is None else np.ascontiguousarray(kf_occupied, np.uint8)
    val = None if lm_valid is None else np.ascontiguousarray(lm_valid, np.uint8)
    assigned = np.full(max(len(pw), 1), -1, np.int32)
    n = lib().ovo_projection_match_by_sim3_transform(C.byref(cam), C.byref(gp), _p(xs), _p(ys), _p(oc), _p(d), _p(occ), len(xs),
                                                     _p(_pose12(sim3_cw)), _p(pw), _p(dm), _p(nr), _p(ld), _p(val), len(pw), _p(sf), len(sf),
                                                     C.c_float(log_scale_factor), C.c_float(margin), _p(assigned))
    return assigned[:len(pw)].copy(), n


def projection_match_keyframes_mutually(cam, gp, kps_1, desc_1, pose_cw_1, lm_pos_w_1, lm_dist_1, lm_desc_1, lm_valid_1, kps_2, desc_2,
                                        pose_cw_2, lm_pos_w_2, lm_dist_2, lm_desc_2, lm_valid_2, s_12, rot_12, trans_12, scale_factors,
                                        log_scale_factor, margin):
    x1, y1, o1, _ = _soa(kps_1)
    x2, y2, o2, _ = _soa(kps_2)
    d1 = np.ascontiguousarray(desc_1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(desc_2, np.uint8).reshape(-1, 32)
    p1 = np.ascontiguousarray(lm_pos_w_1, np.float64).reshape(-1, 3)
    p2 = np.ascontiguousarray(lm_pos_w_2, np.float64).reshape(-1, 3)
    m1 = np.ascontiguousarray(lm_dist_1, np.float32).reshape(-1, 2)
    m2 = np.ascontiguousarray(lm_dist_2, np.float32).reshape(-1, 2)
    l1 = np.ascontiguousarray(lm_desc_1, np.uint8).reshape(-1, 32)
    l2 = np.ascontiguousarray(lm_desc_2, np.uint8).reshape(-1, 32)
    v1 = None if lm_valid_1 is None else np.ascontiguousarray(lm_valid_1, np.uint8)
    v2 = None if lm_valid_2 is None else np.ascontiguousarray(lm_valid_2, np.uint8)
    R = np.ascontiguousarray(rot_12, np.float64).reshape(9)
    t = np.ascontiguousarray(trans_12, np.float64).reshape(3)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.full(max(len(x1), 1), -1, np.int32)
    n = lib().ovo_projection_match_keyframes_mutually(
        C.byref(cam), C.byref(gp), _p(x1), _p(y1), _p(o1), _p(d1), len(x1), _p(_pose12(pose_cw_1)), _p(p1), _p(m1), _p(l1), _p(v1),
        C.byref(cam), C.byref(gp), _p(x2), _p(y2), _p(o2), _p(d2), len(x2), _p(_pose12(pose_cw_2)), _p(p2), _p(m2), _p(l2), _p(v2),
        C.c_double(s_12), _p(R), _p(t), _p(sf), len(sf), C.c_float(log_scale_factor), C.c_float(margin), _p(out))
    return n, out[:len(x1)].copy()


def bow_transform(vocab, desc, levelsup=4):
    """DBoW2 transform, per feature: (word_id, weight, node_id). vocab = dict(child_start, children, desc, weight, word_id, depth)."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    cs = np.ascontiguousarray(vocab["child_start"], np.int32)
    ch = np.ascontiguousarray(vocab["children"], np.int32)
    nd = np.ascontiguousarray(vocab["desc"], np.uint8)
    nw = np.ascontiguousarray(vocab["weight"], np.float64)
    wi = np.ascontiguousarray(vocab["word_id"], np.int32)
    n = len(d)
    word, weight, node = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1)), np.zeros(max(n, 1), np.int32)
    lib().ovo_bow_transform(len(wi), _p(cs), _p(ch), _p(nd), _p(nw), _p(wi), int(vocab["depth"]), _p(d), n, int(levelsup), _p(word), _p(weight),
                            _p(node))
    return word[:n].copy(), weight[:n].copy(), node[:n].copy()
