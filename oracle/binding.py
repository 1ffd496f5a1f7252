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


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
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
    L.ovo_orb_pattern.restype = C.POINTER(C.c_int8)
    L.ovo_orb_create.argtypes = [C.POINTER(OrbParams)]
    L.ovo_orb_create.restype = vp
    L.ovo_orb_destroy.argtypes = [vp]
    L.ovo_orb_set_threads.argtypes = [vp, C.c_int]
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
    L.ovo_robust_brute_force_match.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_float, vp, C.c_int]
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


def distribute_via_tree(xs, ys, responses, min_x, max_x, min_y, max_y, num_keypts):
    xs = np.ascontiguousarray(xs, np.float32)
    ys = np.ascontiguousarray(ys, np.float32)
    rs = np.ascontiguousarray(responses, np.float32)
    out = np.zeros(max(len(xs), 1), np.int32)
    n = lib().ovo_distribute_via_tree(_p(xs), _p(ys), _p(rs), len(xs), min_x, max_x, min_y, max_y, num_keypts, _p(out), len(out))
    return out[:n].copy()


def gaussian_blur(img):
    img = np.ascontiguousarray(img, np.uint8)
    dst = np.zeros_like(img)
    assert lib().ovo_gaussian_blur_7x7(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(dst), dst.strides[0]) == 0
    return dst


def ic_angle(img, x, y, u_max):
    img = np.ascontiguousarray(img, np.uint8)
    um = np.ascontiguousarray(u_max, np.int32)
    return float(lib().ovo_ic_angle(_p(img), img.strides[0], x, y, _p(um)))


def orb_descriptor(blurred, x, y, angle_deg):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().ovo_orb_descriptor(_p(blurred), blurred.strides[0], x, y, angle_deg, _p(d))
    return d


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

    def extract(self, img, mask=None):
        img = np.ascontiguousarray(img, np.uint8)
        cap = self.params.max_num_keypts + 4 * self.params.num_levels + 64
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


def robust_brute_force_match(desc_frm, desc_kf, kf_valid=None, lowe_ratio=0.8):
    desc_frm = np.ascontiguousarray(desc_frm, np.uint8)
    desc_kf = np.ascontiguousarray(desc_kf, np.uint8)
    if kf_valid is not None:
        kf_valid = np.ascontiguousarray(kf_valid, np.uint8)
    pairs = np.zeros((max(len(desc_kf), 1), 2), np.int32)
    n = lib().ovo_robust_brute_force_match(_p(desc_frm), len(desc_frm), _p(desc_kf), len(desc_kf), _p(kf_valid),
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
