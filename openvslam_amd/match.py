"""Host-side mirror of match::base / match::robust (expected: src/openvslam/match/base.h, robust.{h,cc}) over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib

HAMMING_DIST_THR_LOW = 50
HAMMING_DIST_THR_HIGH = 100
MAX_HAMMING_DIST = 256


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class _matcher_ctx:
    def __init__(self, max_n1=4096, max_n2=4096, max_batch=1, device=0):
        self._L = _lib.lib()
        _lib.require_device()
        h = C.c_void_p()
        _lib.check(self._L.ovs_matcher_create(max_n1, max_n2, max_batch, device, C.byref(h)), "ovs_matcher_create")
        self._h = h
        self.max_n1, self.max_n2, self.max_batch = max_n1, max_n2, max_batch

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.ovs_matcher_destroy(self._h)
            self._h = None


class robust(_matcher_ctx):
    """match::robust(lowe_ratio, check_orientation). Only brute_force_match (the all-pairs path) is device code so far."""

    def __init__(self, lowe_ratio=0.6, check_orientation=True, **kw):
        super().__init__(**kw)
        self.lowe_ratio_ = float(lowe_ratio)
        self.check_orientation_ = bool(check_orientation)

    def brute_force_match(self, frm_descriptors, keyfrm_descriptors, keyfrm_has_landmark=None):
        """robust::brute_force_match(frm, keyfrm, matches): returns matches as an (n, 2) int32 array of (idx_1, idx_2)."""
        d1 = np.ascontiguousarray(frm_descriptors, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(keyfrm_descriptors, np.uint8).reshape(-1, 32)
        v = None
        if keyfrm_has_landmark is not None:
            v = np.ascontiguousarray(keyfrm_has_landmark, np.uint8)
            if len(v) != len(d2):
                raise ValueError("keyfrm_has_landmark must have one entry per keyframe keypoint")
        pairs = np.zeros((max(len(d2), 1), 2), np.int32)
        n = C.c_int32()
        _lib.check(self._L.ovs_robust_brute_force_match(self._h, _p(d1), len(d1), _p(d2), len(d2), _p(v), self.lowe_ratio_, _p(pairs),
                                                        len(pairs), C.byref(n)), "ovs_robust_brute_force_match")
        return pairs[:n.value].copy()

    def brute_force_match_batch_dev(self, d_desc_1, d_n1, d_desc_2, d_n2, d_pairs, d_counts, stream=None, d_valid_2=None):
        """Device-resident batch: d_desc_1 (B, cap1, 32) u8, d_desc_2 (B, cap2, 32) u8, d_n1/d_n2 (B,) int32,
        d_pairs (B, cap, 2) int32, d_counts (B,) int32 -- torch CUDA tensors."""
        B = d_desc_1.shape[0]
        _lib.check(self._L.ovs_robust_brute_force_match_batch_dev(
            self._h, d_desc_1.data_ptr(), d_desc_1.stride(0), d_n1.data_ptr(), d_desc_2.data_ptr(), d_desc_2.stride(0), d_n2.data_ptr(),
            d_valid_2.data_ptr() if d_valid_2 is not None else None, B, self.lowe_ratio_, d_pairs.data_ptr(), d_counts.data_ptr(),
            d_pairs.shape[1], stream), "ovs_robust_brute_force_match_batch_dev")


def hamming_best2(ctx, q, t, t_valid=None):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    if t_valid is not None:
        t_valid = np.ascontiguousarray(t_valid, np.uint8)
    bi = np.zeros(len(q), np.int32)
    b = np.zeros(len(q), np.uint16)
    s = np.zeros(len(q), np.uint16)
    _lib.check(ctx._L.ovs_hamming_best2(ctx._h, _p(q), len(q), _p(t), len(t), _p(t_valid), _p(bi), _p(b), _p(s)), "ovs_hamming_best2")
    return bi, b, s
