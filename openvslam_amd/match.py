"""Host-side mirror of match::base / robust / projection / area / bow_tree (expected: src/openvslam/match/*.{h,cc}) and of
data::assign_keypoints_to_grid (src/openvslam/data/common.{h,cc}) over the C ABI."""
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

    NEAR_PATHS = {"matrix": 0, "popcount": 1}   # OVS_NEAR_PATH_MATRIX / OVS_NEAR_PATH_POPCOUNT

    def __init__(self, lowe_ratio=0.6, check_orientation=True, near_path="matrix", **kw):
        super().__init__(**kw)
        self.lowe_ratio_ = float(lowe_ratio)
        self.check_orientation_ = bool(check_orientation)
        self.set_near_path(near_path)

    def set_near_path(self, near_path):
        """Implementation of the all-pairs stage: "matrix" (i8 dot products on the matrix cores, default) or "popcount" (xor / popcount on
        the vector ALU). Bit-identical results."""
        _lib.check(self._L.ovs_matcher_set_near_path(self._h, self.NEAR_PATHS[near_path]), "ovs_matcher_set_near_path")
        self.near_path_ = near_path

    def brute_force_match(self, frm_descriptors, keyfrm_descriptors, keyfrm_has_landmark=None, frm_valid=None):
        """robust::brute_force_match(frm, keyfrm, matches): returns matches as an (n, 2) int32 array of (idx_1, idx_2).
        frm_valid: optional per-frame-keypoint mask (0 = never a candidate), see ovs_robust_brute_force_match."""
        d1 = np.ascontiguousarray(frm_descriptors, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(keyfrm_descriptors, np.uint8).reshape(-1, 32)
        v = None
        if keyfrm_has_landmark is not None:
            v = np.ascontiguousarray(keyfrm_has_landmark, np.uint8)
            if len(v) != len(d2):
                raise ValueError("keyfrm_has_landmark must have one entry per keyframe keypoint")
        v1 = None
        if frm_valid is not None:
            v1 = np.ascontiguousarray(frm_valid, np.uint8)
            if len(v1) != len(d1):
                raise ValueError("frm_valid must have one entry per frame keypoint")
        pairs = np.zeros((max(len(d2), 1), 2), np.int32)
        n = C.c_int32()
        _lib.check(self._L.ovs_robust_brute_force_match(self._h, _p(d1), len(d1), _p(v1), _p(d2), len(d2), _p(v), self.lowe_ratio_, _p(pairs),
                                                        len(pairs), C.byref(n)), "ovs_robust_brute_force_match")
        return pairs[:n.value].copy()

    def brute_force_match_batch_dev(self, d_desc_1, d_n1, d_desc_2, d_n2, d_pairs, d_counts, stream=None, d_valid_2=None, d_valid_1=None):
        """Device-resident batch: d_desc_1 (B, cap1, 32) u8, d_desc_2 (B, cap2, 32) u8, d_n1/d_n2 (B,) int32,
        d_pairs (B, cap, 2) int32, d_counts (B,) int32 -- torch CUDA tensors."""
        B = d_desc_1.shape[0]
        _lib.check(self._L.ovs_robust_brute_force_match_batch_dev(
            self._h, d_desc_1.data_ptr(), d_desc_1.stride(0), d_n1.data_ptr(), d_valid_1.data_ptr() if d_valid_1 is not None else None,
            d_desc_2.data_ptr(), d_desc_2.stride(0), d_n2.data_ptr(),
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


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
                     ("class_id", "<i4")])   # cv::KeyPoint / ovs_keypoint


def set_variant(which, value):
    """ovs_match_set_variant, process-wide: "angle_keep_rule" (0 top-3 | 1 top-3 with the 0.1 x max rule), "angle_tie_order" (0 lower of two equally
    full bins first | 1 higher first) -- oracle/ORACLE_SPEC.md rule 17 --, "bf_frame_mask" (0 | 1: robust::brute_force_match's class shim also skips
    frame keypoints that own a landmark, rule 14)."""
    _lib.check(_lib.lib().ovs_match_set_variant({"angle_keep_rule": 0, "angle_tie_order": 1, "bf_frame_mask": 2}[which], int(value)), "ovs_match_set_variant")


def grid_params(cols, rows, num_grid_cols=64, num_grid_rows=48, min_x=0.0, min_y=0.0):
    """camera::base for an undistorted image: img_bounds_ = [0, cols] x [0, rows], 64 x 48 cells."""
    return _lib.GridParams(min_x, min_y, float(cols), float(rows), num_grid_cols, num_grid_rows)


def flatten_bow(feat_vec):
    """std::map<node id, std::vector<unsigned>> (a dict here) -> CSR over ascending node ids."""
    ids = np.array(sorted(feat_vec), np.int32)
    start = np.zeros(len(ids) + 1, np.int32)
    items = []
    for k, i in enumerate(ids):
        items.extend(feat_vec[int(i)])
        start[k + 1] = len(items)
    return ids, start, np.array(items, np.int32)


class _window_ctx:
    def __init__(self, max_targets=8192, max_queries=16384, max_entries=1 << 21, device=0):
        self._L = _lib.lib()
        _lib.require_device()
        h = C.c_void_p()
        _lib.check(self._L.ovs_wmatcher_create(max_targets, max_queries, max_entries, device, C.byref(h)), "ovs_wmatcher_create")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.ovs_wmatcher_destroy(self._h)
            self._h = None

    def assign_keypoints_to_grid(self, gp, keypts):
        """data::assign_keypoints_to_grid -> (cell_start[cols*rows+1], items); cell id = cx*rows + cy."""
        k = np.ascontiguousarray(keypts, KP_DTYPE)
        nc = gp.cols * gp.rows
        start = np.zeros(nc + 1, np.int32)
        items = np.zeros(max(len(k), 1), np.int32)
        n = C.c_int32()
        _lib.check(self._L.ovs_assign_keypoints_to_grid(self._h, C.byref(gp), _p(k), len(k), _p(start), _p(items), C.byref(n)),
                   "ovs_assign_keypoints_to_grid")
        return start, items[:n.value].copy()


class frame_dev:
    """A frame's matcher-side data resident in HBM (ovs_frame_dev): undistorted keypoints, descriptors, stereo_x_right and the keypoint grid,
    uploaded and indexed once; pass it as `frm_keypts` / `curr_keypts` / `keypts_1` / `keypts_2` to the matchers below instead of host arrays."""

    def __init__(self, gp, keypts, desc, stereo_x_right=None, device=0):
        self._L = _lib.lib()
        _lib.require_device()
        k = np.ascontiguousarray(keypts, KP_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        xr = None if stereo_x_right is None else np.ascontiguousarray(stereo_x_right, np.float32)
        h = C.c_void_p()
        _lib.check(self._L.ovs_frame_dev_create(device, C.byref(gp), _p(k), _p(d), _p(xr), len(k), C.byref(h)), "ovs_frame_dev_create")
        self._h = h
        self.num_keypts = len(k)
        self.has_stereo = xr is not None

    def attach_bearings(self, bearings):
        """3 doubles per keypoint (keyframe::bearings_), uploaded once: robust_triangulation.match_for_triangulation needs them."""
        b = np.ascontiguousarray(bearings, np.float64).reshape(-1, 3)
        assert len(b) == self.num_keypts
        _lib.check(self._L.ovs_frame_dev_attach_bearings(self._h, _p(b)), "ovs_frame_dev_attach_bearings")
        return self

    @property
    def device(self):
        return self._L.ovs_frame_dev_device(self._h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.ovs_frame_dev_destroy(h)


class projection(_window_ctx):
    """match::projection(lowe_ratio, check_orientation)."""

    def __init__(self, lowe_ratio=0.6, check_orientation=True, **kw):
        super().__init__(**kw)
        self.lowe_ratio_ = float(lowe_ratio)
        self.check_orientation_ = bool(check_orientation)

    def match_frame_and_landmarks(self, gp, frm_keypts, frm_desc, scale_factors, lm_reproj, lm_level, lm_desc, margin=5.0,
                                  frm_stereo_x_right=None, frm_occupied=None, lm_x_right=None, lm_valid=None):
        """projection::match_frame_and_landmarks(frm, local_landmarks, margin): returns (assigned, num_matches) where
        assigned[l] is the frame keypoint that receives landmark l (frm.landmarks_[assigned[l]] = local_landmarks[l]) or -1."""
        sf = np.ascontiguousarray(scale_factors, np.float32)
        xy = np.ascontiguousarray(lm_reproj, np.float32).reshape(-1, 2)
        if isinstance(frm_keypts, frame_dev):   # the frame side is resident: gp / frm_desc / frm_stereo_x_right are the handle's
            lv = np.ascontiguousarray(lm_level, np.int32)
            ld = np.ascontiguousarray(lm_desc, np.uint8).reshape(-1, 32)
            occ = None if frm_occupied is None else np.ascontiguousarray(frm_occupied, np.uint8)
            lxr = None if lm_x_right is None else np.ascontiguousarray(lm_x_right, np.float32)
            val = None if lm_valid is None else np.ascontiguousarray(lm_valid, np.uint8)
            assigned = np.full(max(len(xy), 1), -1, np.int32)
            n = C.c_int32()
            _lib.check(self._L.ovs_projection_match_frame_and_landmarks_f(
                self._h, frm_keypts._h, _p(occ), _p(xy), _p(lxr), _p(lv), _p(ld), _p(val), len(xy), _p(sf), len(sf), float(margin), self.lowe_ratio_,
                _p(assigned), C.byref(n)), "ovs_projection_match_frame_and_landmarks_f")
            return assigned[:len(xy)].copy(), n.value
        k = np.ascontiguousarray(frm_keypts, KP_DTYPE)
        d = np.ascontiguousarray(frm_desc, np.uint8).reshape(-1, 32)
        lv = np.ascontiguousarray(lm_level, np.int32)
        ld = np.ascontiguousarray(lm_desc, np.uint8).reshape(-1, 32)
        xr = None if frm_stereo_x_right is None else np.ascontiguousarray(frm_stereo_x_right, np.float32)
        occ = None if frm_occupied is None else np.ascontiguousarray(frm_occupied, np.uint8)
        lxr = None if lm_x_right is None else np.ascontiguousarray(lm_x_right, np.float32)
        val = None if lm_valid is None else np.ascontiguousarray(lm_valid, np.uint8)
        assigned = np.full(max(len(xy), 1), -1, np.int32)
        n = C.c_int32()
        _lib.check(self._L.ovs_projection_match_frame_and_landmarks(
            self._h, C.byref(gp), _p(k), _p(d), _p(xr), _p(occ), len(k), _p(xy), _p(lxr), _p(lv), _p(ld), _p(val), len(xy), _p(sf), len(sf),
            float(margin), self.lowe_ratio_, _p(assigned), C.byref(n)), "ovs_projection_match_frame_and_landmarks")
        return assigned[:len(xy)].copy(), n.value


    def match_current_and_last_frames(self, cam, gp, curr_keypts, curr_desc, pose_cw_curr, last_keypts, last_pos_w, last_lm_desc,
                                      pose_cw_last, scale_factors, margin, curr_stereo_x_right=None, curr_occupied=None, last_valid=None):
        """projection::match_current_and_last_frames(curr_frm, last_frm, margin): returns (assigned, num_matches); assigned[i] is the
        current keypoint that receives last_frm.landmarks_[i], or -1. Poses are 3x4 [R|t] world->camera."""
        lk = np.ascontiguousarray(last_keypts, KP_DTYPE)
        pw = np.ascontiguousarray(last_pos_w, np.float64).reshape(-1, 3)
        ld = np.ascontiguousarray(last_lm_desc, np.uint8).reshape(-1, 32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        pc = _pose12(pose_cw_curr)
        pl = _pose12(pose_cw_last)
        if isinstance(curr_keypts, frame_dev):   # the current frame is resident
            occ = None if curr_occupied is None else np.ascontiguousarray(curr_occupied, np.uint8)
            val = None if last_valid is None else np.ascontiguousarray(last_valid, np.uint8)
            assigned = np.full(max(len(lk), 1), -1, np.int32)
            n = C.c_int32()
            _lib.check(self._L.ovs_projection_match_current_and_last_frames_f(
                self._h, C.byref(cam), curr_keypts._h, _p(occ), _p(pc), _p(lk), _p(pw), _p(ld), _p(val), len(lk), _p(pl), _p(sf), len(sf), float(margin),
                1 if self.check_orientation_ else 0, _p(assigned), C.byref(n)), "ovs_projection_match_current_and_last_frames_f")
            return assigned[:len(lk)].copy(), n.value
        ck = np.ascontiguousarray(curr_keypts, KP_DTYPE)
        cd = np.ascontiguousarray(curr_desc, np.uint8).reshape(-1, 32)
        xr = None if curr_stereo_x_right is None else np.ascontiguousarray(curr_stereo_x_right, np.float32)
        occ = None if curr_occupied is None else np.ascontiguousarray(curr_occupied, np.uint8)
        val = None if last_valid is None else np.ascontiguousarray(last_valid, np.uint8)
        assigned = np.full(max(len(lk), 1), -1, np.int32)
        n = C.c_int32()
        _lib.check(self._L.ovs_projection_match_current_and_last_frames(
            self._h, C.byref(cam), C.byref(gp), _p(ck), _p(cd), _p(xr), _p(occ), len(ck), _p(pc), _p(lk), _p(pw), _p(ld), _p(val), len(lk),
            _p(pl), _p(sf), len(sf), float(margin), int(self.check_orientation_), _p(assigned), C.byref(n)),
            "ovs_projection_match_current_and_last_frames")
        return assigned[:len(lk)].copy(), n.value


    def match_frame_and_keyframe(self, cam, gp, curr_keypts, curr_desc, pose_cw_curr, kf_keypts, kf_pos_w, kf_dist_min_max, kf_lm_desc,
                                 scale_factors, log_scale_factor, margin, hamm_dist_thr, curr_occupied=None, kf_valid=None):
        """projection::match_frame_and_keyframe(curr_frm, keyfrm, already_matched_lms, margin, hamm_dist_thr): returns
        (assigned, num_matches); assigned[i] is the current keypoint that receives the keyframe's landmark i, or -1.
        curr_keypts may be a frame_dev (the current frame resident; curr_desc and gp are then ignored)."""
        resident = isinstance(curr_keypts, frame_dev)
        if not resident:
            ck = np.ascontiguousarray(curr_keypts, KP_DTYPE)
            cd = np.ascontiguousarray(curr_desc, np.uint8).reshape(-1, 32)
        kk = np.ascontiguousarray(kf_keypts, KP_DTYPE)
        pw = np.ascontiguousarray(kf_pos_w, np.float64).reshape(-1, 3)
        dm = np.ascontiguousarray(kf_dist_min_max, np.float32).reshape(-1, 2)
        ld = np.ascontiguousarray(kf_lm_desc, np.uint8).reshape(-1, 32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        occ = None if curr_occupied is None else np.ascontiguousarray(curr_occupied, np.uint8)
        val = None if kf_valid is None else np.ascontiguousarray(kf_valid, np.uint8)
        assigned = np.full(max(len(kk), 1), -1, np.int32)
        n = C.c_int32()
        if resident:
            _lib.check(self._L.ovs_projection_match_frame_and_keyframe_f(
                self._h, C.byref(cam), curr_keypts._h, _p(occ), _p(_pose12(pose_cw_curr)), _p(kk), _p(pw), _p(dm), _p(ld), _p(val), len(kk), _p(sf),
                len(sf), float(log_scale_factor), float(margin), int(hamm_dist_thr), int(self.check_orientation_), _p(assigned), C.byref(n)),
                "ovs_projection_match_frame_and_keyframe_f")
            return assigned[:len(kk)].copy(), n.value
        _lib.check(self._L.ovs_projection_match_frame_and_keyframe(
            self._h, C.byref(cam), C.byref(gp), _p(ck), _p(cd), _p(occ), len(ck), _p(_pose12(pose_cw_curr)), _p(kk), _p(pw), _p(dm), _p(ld),
            _p(val), len(kk), _p(sf), len(sf), float(log_scale_factor), float(margin), int(hamm_dist_thr), int(self.check_orientation_),
            _p(assigned), C.byref(n)), "ovs_projection_match_frame_and_keyframe")
        return assigned[:len(kk)].copy(), n.value


    def match_by_Sim3_transform(self, cam, gp, keyfrm_keypts, keyfrm_desc, Sim3_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc,
                                scale_factors, log_scale_factor, margin, keyfrm_occupied=None, lm_valid=None):
        """projection::match_by_Sim3_transform(keyfrm, Sim3_cw, landmarks, matched_lms_in_keyfrm, margin): returns (assigned,
        num_matches); assigned[l] is the keyframe keypoint that receives landmark l, or -1. keyfrm_keypts may be a frame_dev."""
        resident = isinstance(keyfrm_keypts, frame_dev)
        if not resident:
            k = np.ascontiguousarray(keyfrm_keypts, KP_DTYPE)
            d = np.ascontiguousarray(keyfrm_desc, np.uint8).reshape(-1, 32)
        pw = np.ascontiguousarray(lm_pos_w, np.float64).reshape(-1, 3)
        dm = np.ascontiguousarray(lm_dist_min_max, np.float32).reshape(-1, 2)
        nr = np.ascontiguousarray(lm_normal, np.float64).reshape(-1, 3)
        ld = np.ascontiguousarray(lm_desc, np.uint8).reshape(-1, 32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        occ = None if keyfrm_occupied is None else np.ascontiguousarray(keyfrm_occupied, np.uint8)
        val = None if lm_valid is None else np.ascontiguousarray(lm_valid, np.uint8)
        assigned = np.full(max(len(pw), 1), -1, np.int32)
        n = C.c_int32()
        if resident:
            _lib.check(self._L.ovs_projection_match_by_sim3_transform_f(
                self._h, C.byref(cam), keyfrm_keypts._h, _p(occ), _p(_pose12(Sim3_cw)), _p(pw), _p(dm), _p(nr), _p(ld), _p(val), len(pw), _p(sf),
                len(sf), float(log_scale_factor), float(margin), _p(assigned), C.byref(n)), "ovs_projection_match_by_sim3_transform_f")
            return assigned[:len(pw)].copy(), n.value
        _lib.check(self._L.ovs_projection_match_by_sim3_transform(
            self._h, C.byref(cam), C.byref(gp), _p(k), _p(d), _p(occ), len(k), _p(_pose12(Sim3_cw)), _p(pw), _p(dm), _p(nr), _p(ld), _p(val),
            len(pw), _p(sf), len(sf), float(log_scale_factor), float(margin), _p(assigned), C.byref(n)),
            "ovs_projection_match_by_sim3_transform")
        return assigned[:len(pw)].copy(), n.value

    def match_keyframes_mutually(self, cam, gp, keypts_1, desc_1, pose_cw_1, lm_pos_w_1, lm_dist_1, lm_desc_1, lm_valid_1, keypts_2, desc_2,
                                 pose_cw_2, lm_pos_w_2, lm_dist_2, lm_desc_2, lm_valid_2, s_12, rot_12, trans_12, scale_factors,
                                 log_scale_factor, margin, cam_2=None, gp_2=None):
        """projection::match_keyframes_mutually(keyfrm_1, keyfrm_2, matched_lms_in_keyfrm_1, s_12, rot_12, trans_12, margin): returns
        (num_matches, matched_2_in_1); per-keypoint landmark arrays, lm_valid_k marks the keypoints whose landmark takes part.
        keypts_1 and keypts_2 may both be frame_dev handles (both keyframes resident)."""
        resident = isinstance(keypts_1, frame_dev) and isinstance(keypts_2, frame_dev)
        if not resident:
            k1 = np.ascontiguousarray(keypts_1, KP_DTYPE)
            k2 = np.ascontiguousarray(keypts_2, KP_DTYPE)
            d1 = np.ascontiguousarray(desc_1, np.uint8).reshape(-1, 32)
            d2 = np.ascontiguousarray(desc_2, np.uint8).reshape(-1, 32)
        n1 = keypts_1.num_keypts if resident else len(k1)
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
        cam_2 = cam if cam_2 is None else cam_2
        gp_2 = gp if gp_2 is None else gp_2
        out = np.full(max(n1, 1), -1, np.int32)
        n = C.c_int32()
        if resident:
            _lib.check(self._L.ovs_projection_match_keyframes_mutually_f(
                self._h, C.byref(cam), keypts_1._h, _p(_pose12(pose_cw_1)), _p(p1), _p(m1), _p(l1), _p(v1), C.byref(cam_2), keypts_2._h,
                _p(_pose12(pose_cw_2)), _p(p2), _p(m2), _p(l2), _p(v2), float(s_12), _p(R), _p(t), _p(sf), len(sf), float(log_scale_factor),
                float(margin), _p(out), C.byref(n)), "ovs_projection_match_keyframes_mutually_f")
            return n.value, out[:n1].copy()
        _lib.check(self._L.ovs_projection_match_keyframes_mutually(
            self._h, C.byref(cam), C.byref(gp), _p(k1), _p(d1), len(k1), _p(_pose12(pose_cw_1)), _p(p1), _p(m1), _p(l1), _p(v1), C.byref(cam_2),
            C.byref(gp_2), _p(k2), _p(d2), len(k2), _p(_pose12(pose_cw_2)), _p(p2), _p(m2), _p(l2), _p(v2), float(s_12), _p(R), _p(t), _p(sf),
            len(sf), float(log_scale_factor), float(margin), _p(out), C.byref(n)), "ovs_projection_match_keyframes_mutually")
        return n.value, out[:len(k1)].copy()


def _pose12(pose_cw):
    """3x4 (or 4x4) [R|t] -> 12 doubles: rotation row-major, then translation."""
    T = np.asarray(pose_cw, np.float64)
    return np.ascontiguousarray(np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]))


class area(_window_ctx):
    """match::area(lowe_ratio, check_orientation)."""

    def __init__(self, lowe_ratio=0.9, check_orientation=True, **kw):
        super().__init__(**kw)
        self.lowe_ratio_ = float(lowe_ratio)
        self.check_orientation_ = bool(check_orientation)

    def match_in_consistent_area(self, gp, keypts_1, desc_1, keypts_2, desc_2, prev_matched_pts, margin=10):
        """area::match_in_consistent_area(frm_1, frm_2, prev_matched_pts, matched_indices_2_in_frm_1, margin):
        returns (num_matches, matched_indices_2_in_frm_1); prev_matched_pts (n1, 2) float32 is updated in place."""
        if isinstance(keypts_1, frame_dev) and isinstance(keypts_2, frame_dev):   # both frames resident (gp / desc_* are the handles')
            n1 = keypts_1.num_keypts
            if prev_matched_pts.dtype != np.float32 or not prev_matched_pts.flags.c_contiguous or prev_matched_pts.shape != (n1, 2):
                raise ValueError("prev_matched_pts must be a C-contiguous (n1, 2) float32 array (it is updated in place)")
            matched = np.full(max(n1, 1), -1, np.int32)
            n = C.c_int32()
            _lib.check(self._L.ovs_area_match_in_consistent_area_f(self._h, keypts_1._h, keypts_2._h, _p(prev_matched_pts), _p(matched), int(margin),
                                                                   self.lowe_ratio_, int(self.check_orientation_), C.byref(n)),
                       "ovs_area_match_in_consistent_area_f")
            return n.value, matched[:n1].copy()
        k1 = np.ascontiguousarray(keypts_1, KP_DTYPE)
        k2 = np.ascontiguousarray(keypts_2, KP_DTYPE)
        d1 = np.ascontiguousarray(desc_1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(desc_2, np.uint8).reshape(-1, 32)
        if prev_matched_pts.dtype != np.float32 or not prev_matched_pts.flags.c_contiguous or prev_matched_pts.shape != (len(k1), 2):
            raise ValueError("prev_matched_pts must be a C-contiguous (n1, 2) float32 array (it is updated in place)")
        matched = np.full(max(len(k1), 1), -1, np.int32)
        n = C.c_int32()
        _lib.check(self._L.ovs_area_match_in_consistent_area(self._h, C.byref(gp), _p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2),
                                                             _p(prev_matched_pts), _p(matched), int(margin), self.lowe_ratio_,
                                                             int(self.check_orientation_), C.byref(n)), "ovs_area_match_in_consistent_area")
        return n.value, matched[:len(k1)].copy()


class bow_tree(_window_ctx):
    """match::bow_tree(lowe_ratio, check_orientation)."""

    def __init__(self, lowe_ratio=0.6, check_orientation=True, **kw):
        super().__init__(**kw)
        self.lowe_ratio_ = float(lowe_ratio)
        self.check_orientation_ = bool(check_orientation)

    def match_frame_and_keyframe(self, kf_keypts, kf_desc, kf_bow_feat_vec, frm_keypts, frm_desc, frm_bow_feat_vec, kf_has_landmark=None):
        """bow_tree::match_frame_and_keyframe(keyfrm, frm, matched_lms_in_frm): returns (num_matches, matched_kf_in_frm) where
        matched_kf_in_frm[j] is the keyframe keypoint whose landmark frame keypoint j receives, or -1. kf_keypts and frm_keypts may
        both be frame_dev handles (keyframe and frame resident)."""
        v = None if kf_has_landmark is None else np.ascontiguousarray(kf_has_landmark, np.uint8)
        ki, ks, kit = flatten_bow(kf_bow_feat_vec)
        fi, fs, fit = flatten_bow(frm_bow_feat_vec)
        n = C.c_int32()
        if isinstance(kf_keypts, frame_dev) and isinstance(frm_keypts, frame_dev):
            out = np.full(max(frm_keypts.num_keypts, 1), -1, np.int32)
            _lib.check(self._L.ovs_bow_match_frame_and_keyframe_f(self._h, kf_keypts._h, _p(v), _p(ki), _p(ks), _p(kit), len(ki), frm_keypts._h, _p(fi),
                                                                  _p(fs), _p(fit), len(fi), self.lowe_ratio_, int(self.check_orientation_), _p(out),
                                                                  C.byref(n)), "ovs_bow_match_frame_and_keyframe_f")
            return n.value, out[:frm_keypts.num_keypts].copy()
        kk = np.ascontiguousarray(kf_keypts, KP_DTYPE)
        fk = np.ascontiguousarray(frm_keypts, KP_DTYPE)
        kd = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32)
        fd = np.ascontiguousarray(frm_desc, np.uint8).reshape(-1, 32)
        out = np.full(max(len(fk), 1), -1, np.int32)
        _lib.check(self._L.ovs_bow_match_frame_and_keyframe(self._h, _p(kk), _p(kd), _p(v), len(kk), _p(ki), _p(ks), _p(kit), len(ki), _p(fk),
                                                            _p(fd), len(fk), _p(fi), _p(fs), _p(fit), len(fi), self.lowe_ratio_,
                                                            int(self.check_orientation_), _p(out), C.byref(n)),
                   "ovs_bow_match_frame_and_keyframe")
        return n.value, out[:len(fk)].copy()


    def match_keyframes(self, kps_1, desc_1, bow_feat_vec_1, kps_2, desc_2, bow_feat_vec_2, has_landmark_1=None, has_landmark_2=None):
        """bow_tree::match_keyframes(keyfrm_1, keyfrm_2, matched_lms_in_keyfrm_1): returns (num_matches, matched_2_in_1) where
        matched_2_in_1[idx_1] is the keyframe-2 keypoint whose landmark keyframe-1 keypoint idx_1 receives, or -1. kps_1 and kps_2 may
        both be frame_dev handles."""
        v1 = None if has_landmark_1 is None else np.ascontiguousarray(has_landmark_1, np.uint8)
        v2 = None if has_landmark_2 is None else np.ascontiguousarray(has_landmark_2, np.uint8)
        i1, s1, t1 = flatten_bow(bow_feat_vec_1)
        i2, s2, t2 = flatten_bow(bow_feat_vec_2)
        n = C.c_int32()
        if isinstance(kps_1, frame_dev) and isinstance(kps_2, frame_dev):
            out = np.full(max(kps_1.num_keypts, 1), -1, np.int32)
            _lib.check(self._L.ovs_bow_match_keyframes_f(self._h, kps_1._h, _p(v1), _p(i1), _p(s1), _p(t1), len(i1), kps_2._h, _p(v2), _p(i2), _p(s2),
                                                         _p(t2), len(i2), self.lowe_ratio_, int(self.check_orientation_), _p(out), C.byref(n)),
                       "ovs_bow_match_keyframes_f")
            return n.value, out[:kps_1.num_keypts].copy()
        k1 = np.ascontiguousarray(kps_1, KP_DTYPE)
        k2 = np.ascontiguousarray(kps_2, KP_DTYPE)
        d1 = np.ascontiguousarray(desc_1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(desc_2, np.uint8).reshape(-1, 32)
        out = np.full(max(len(k1), 1), -1, np.int32)
        _lib.check(self._L.ovs_bow_match_keyframes(self._h, _p(k1), _p(d1), _p(v1), len(k1), _p(i1), _p(s1), _p(t1), len(i1), _p(k2), _p(d2),
                                                   _p(v2), len(k2), _p(i2), _p(s2), _p(t2), len(i2), self.lowe_ratio_,
                                                   int(self.check_orientation_), _p(out), C.byref(n)), "ovs_bow_match_keyframes")
        return n.value, out[:len(k1)].copy()


class fuse(_window_ctx):
    """match::fuse(lowe_ratio): the candidate search of replace_duplication (the landmark-graph surgery stays with the caller)."""

    def __init__(self, lowe_ratio=0.6, **kw):
        super().__init__(**kw)
        self.lowe_ratio_ = float(lowe_ratio)

    def replace_duplication(self, cam, gp, keyfrm_keypts, keyfrm_desc, pose_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc,
                            scale_factors, inv_level_sigma_sq, log_scale_factor, margin=3.0, keyfrm_stereo_x_right=None, lm_valid=None):
        """returns (best_idx, num_fused): best_idx[l] = keyframe keypoint landmark l fuses with, or -1. keyfrm_keypts may be a frame_dev
        (the keyframe resident, stereo_x_right included; keyfrm_desc, gp and keyfrm_stereo_x_right are then ignored)."""
        resident = isinstance(keyfrm_keypts, frame_dev)
        if not resident:
            k = np.ascontiguousarray(keyfrm_keypts, KP_DTYPE)
            d = np.ascontiguousarray(keyfrm_desc, np.uint8).reshape(-1, 32)
        pw = np.ascontiguousarray(lm_pos_w, np.float64).reshape(-1, 3)
        dm = np.ascontiguousarray(lm_dist_min_max, np.float32).reshape(-1, 2)
        nr = np.ascontiguousarray(lm_normal, np.float64).reshape(-1, 3)
        ld = np.ascontiguousarray(lm_desc, np.uint8).reshape(-1, 32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        ils = np.ascontiguousarray(inv_level_sigma_sq, np.float32)
        xr = None if keyfrm_stereo_x_right is None else np.ascontiguousarray(keyfrm_stereo_x_right, np.float32)
        val = None if lm_valid is None else np.ascontiguousarray(lm_valid, np.uint8)
        best = np.full(max(len(pw), 1), -1, np.int32)
        n = C.c_int32()
        if resident:
            _lib.check(self._L.ovs_fuse_replace_duplication_f(self._h, C.byref(cam), keyfrm_keypts._h, _p(_pose12(pose_cw)), _p(pw), _p(dm), _p(nr), _p(ld),
                                                              _p(val), len(pw), _p(sf), _p(ils), len(sf), float(log_scale_factor), float(margin),
                                                              _p(best), C.byref(n)), "ovs_fuse_replace_duplication_f")
            return best[:len(pw)].copy(), n.value
        _lib.check(self._L.ovs_fuse_replace_duplication(self._h, C.byref(cam), C.byref(gp), _p(k), _p(d), _p(xr), len(k), _p(_pose12(pose_cw)),
                                                        _p(pw), _p(dm), _p(nr), _p(ld), _p(val), len(pw), _p(sf), _p(ils), len(sf),
                                                        float(log_scale_factor), float(margin), _p(best), C.byref(n)),
                   "ovs_fuse_replace_duplication")
        return best[:len(pw)].copy(), n.value


    def detect_duplication(self, cam, gp, keyfrm_keypts, keyfrm_desc, Sim3_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc, scale_factors,
                           log_scale_factor, margin, lm_valid=None):
        """fuse::detect_duplication(keyfrm, Sim3_cw, landmarks_to_check, margin, duplicated_lms_in_keyfrm), candidate search: returns
        (best_idx, num_found). keyfrm_keypts may be a frame_dev."""
        resident = isinstance(keyfrm_keypts, frame_dev)
        if not resident:
            k = np.ascontiguousarray(keyfrm_keypts, KP_DTYPE)
            d = np.ascontiguousarray(keyfrm_desc, np.uint8).reshape(-1, 32)
        pw = np.ascontiguousarray(lm_pos_w, np.float64).reshape(-1, 3)
        dm = np.ascontiguousarray(lm_dist_min_max, np.float32).reshape(-1, 2)
        nr = np.ascontiguousarray(lm_normal, np.float64).reshape(-1, 3)
        ld = np.ascontiguousarray(lm_desc, np.uint8).reshape(-1, 32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        val = None if lm_valid is None else np.ascontiguousarray(lm_valid, np.uint8)
        best = np.full(max(len(pw), 1), -1, np.int32)
        n = C.c_int32()
        if resident:
            _lib.check(self._L.ovs_fuse_detect_duplication_f(self._h, C.byref(cam), keyfrm_keypts._h, _p(_pose12(Sim3_cw)), _p(pw), _p(dm), _p(nr), _p(ld),
                                                             _p(val), len(pw), _p(sf), len(sf), float(log_scale_factor), float(margin), _p(best),
                                                             C.byref(n)), "ovs_fuse_detect_duplication_f")
            return best[:len(pw)].copy(), n.value
        _lib.check(self._L.ovs_fuse_detect_duplication(self._h, C.byref(cam), C.byref(gp), _p(k), _p(d), len(k), _p(_pose12(Sim3_cw)), _p(pw),
                                                       _p(dm), _p(nr), _p(ld), _p(val), len(pw), _p(sf), len(sf), float(log_scale_factor),
                                                       float(margin), _p(best), C.byref(n)), "ovs_fuse_detect_duplication")
        return best[:len(pw)].copy(), n.value


class stereo:
    """match::stereo(left_image_pyramid, right_image_pyramid, keypts_left, keypts_right, descs_left, descs_right, scale_factors,
    inv_scale_factors, focal_x_baseline, true_baseline). The pyramids and scale tables are those of the two extractors' last extract
    (they stay in HBM), so the ctor takes the extractors."""

    def __init__(self, extractor_left, extractor_right, keypts_left, descs_left, keypts_right, descs_right, focal_x_baseline, true_baseline,
                 max_rows=2048, max_keypoints=8192, device=0):
        self._L = _lib.lib()
        _lib.require_device()
        h = C.c_void_p()
        _lib.check(self._L.ovs_stereo_create(max_rows, max_keypoints, device, C.byref(h)), "ovs_stereo_create")
        self._h = h
        self._el, self._er = extractor_left, extractor_right
        self._kl = np.ascontiguousarray(keypts_left, KP_DTYPE)
        self._kr = np.ascontiguousarray(keypts_right, KP_DTYPE)
        self._dl = np.ascontiguousarray(descs_left, np.uint8).reshape(-1, 32)
        self._dr = np.ascontiguousarray(descs_right, np.uint8).reshape(-1, 32)
        self.focal_x_baseline_, self.true_baseline_ = float(focal_x_baseline), float(true_baseline)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.ovs_stereo_destroy(self._h)
            self._h = None

    def set_variant(self, which, value):
        """ovs_stereo_set_variant: "outlier_factor" (0: 2.0 | 1: 2.1), "parabola" (0: float | 1: double) -- ORACLE_SPEC rule 20's L-tagged choices."""
        idx = {"outlier_factor": 0, "parabola": 1}[which]
        _lib.check(self._L.ovs_stereo_set_variant(self._h, idx, int(value)), "ovs_stereo_set_variant")

    def compute(self):
        """stereo::compute(stereo_x_right, depths): returns (stereo_x_right, depths), -1 where a keypoint has no stereo match."""
        n = len(self._kl)
        xr = np.full(max(n, 1), -1, np.float32)
        dp = np.full(max(n, 1), -1, np.float32)
        nv = C.c_int32()
        _lib.check(self._L.ovs_stereo_compute(self._h, self._el._h, self._er._h, _p(self._kl), _p(self._dl), n, _p(self._kr), _p(self._dr),
                                              len(self._kr), self.focal_x_baseline_, self.true_baseline_, _p(xr), _p(dp), C.byref(nv)),
                   "ovs_stereo_compute")
        self.num_valid_ = nv.value
        return xr[:n].copy(), dp[:n].copy()


class robust_triangulation(_window_ctx):
    """match::robust(lowe_ratio, check_orientation)::match_for_triangulation (the BoW + epipolar-constraint matcher of
    mapping_module::create_new_landmarks). Separate context class: it runs on the windowed-matcher kernels."""

    def __init__(self, lowe_ratio=0.6, check_orientation=True, **kw):
        super().__init__(**kw)
        self.lowe_ratio_ = float(lowe_ratio)
        self.check_orientation_ = bool(check_orientation)

    def match_for_triangulation(self, kps_1, desc_1, bow_feat_vec_1, bearings_1, kps_2, desc_2, bow_feat_vec_2, bearings_2, E_12, epipole_in_2,
                                scale_factors, has_lm_1=None, has_lm_2=None, x_right_1=None, x_right_2=None):
        """returns (num_matches, matched_idx_pairs) with matched_idx_pairs an (n, 2) array of (idx_1, idx_2), ascending idx_1.
        kps_1 and kps_2 may both be frame_dev handles with bearings attached (descriptors, stereo_x_right and bearings then come from them)."""
        if isinstance(kps_1, frame_dev) and isinstance(kps_2, frame_dev):
            h1 = None if has_lm_1 is None else np.ascontiguousarray(has_lm_1, np.uint8)
            h2 = None if has_lm_2 is None else np.ascontiguousarray(has_lm_2, np.uint8)
            E = np.ascontiguousarray(E_12, np.float64).reshape(9)
            ep = np.ascontiguousarray(epipole_in_2, np.float64).reshape(3)
            sf = np.ascontiguousarray(scale_factors, np.float32)
            i1, s1, t1 = flatten_bow(bow_feat_vec_1)
            i2, s2, t2 = flatten_bow(bow_feat_vec_2)
            out = np.full(max(kps_1.num_keypts, 1), -1, np.int32)
            n = C.c_int32()
            _lib.check(self._L.ovs_robust_match_for_triangulation_f(self._h, kps_1._h, _p(h1), _p(i1), _p(s1), _p(t1), len(i1), kps_2._h, _p(h2), _p(i2),
                                                                    _p(s2), _p(t2), len(i2), _p(E), _p(ep), _p(sf), len(sf),
                                                                    int(self.check_orientation_), _p(out), C.byref(n)),
                       "ovs_robust_match_for_triangulation_f")
            m = out[:kps_1.num_keypts]
            idx = np.nonzero(m >= 0)[0]
            return n.value, np.stack([idx, m[idx]], 1).astype(np.int32)
        k1 = np.ascontiguousarray(kps_1, KP_DTYPE)
        k2 = np.ascontiguousarray(kps_2, KP_DTYPE)
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
        i1, s1, t1 = flatten_bow(bow_feat_vec_1)
        i2, s2, t2 = flatten_bow(bow_feat_vec_2)
        out = np.full(max(len(k1), 1), -1, np.int32)
        n = C.c_int32()
        _lib.check(self._L.ovs_robust_match_for_triangulation(self._h, _p(k1), _p(d1), _p(h1), _p(x1), _p(b1), len(k1), _p(i1), _p(s1), _p(t1),
                                                              len(i1), _p(k2), _p(d2), _p(h2), _p(x2), _p(b2), len(k2), _p(i2), _p(s2), _p(t2),
                                                              len(i2), _p(E), _p(ep), _p(sf), len(sf), int(self.check_orientation_), _p(out),
                                                              C.byref(n)), "ovs_robust_match_for_triangulation")
        m = out[:len(k1)]
        idx = np.nonzero(m >= 0)[0]
        return n.value, np.stack([idx, m[idx]], 1).astype(np.int32)
