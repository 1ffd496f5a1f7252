"""Host-side mirror of feature::orb_extractor / feature::orb_params (expected: src/openvslam/feature/orb_extractor.h,
orb_params.h) over the C ABI. Same names, same argument meaning, same defaults; the computation is the HIP library's."""
import ctypes as C

import numpy as np

from . import _lib

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])   # cv::KeyPoint layout, 28 bytes


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class orb_params:
    """feature::orb_params: max_num_keypts=2000, scale_factor=1.2, num_levels=8, ini_fast_thr=20, min_fast_thr=7."""

    def __init__(self, max_num_keypts=2000, scale_factor=1.2, num_levels=8, ini_fast_thr=20, min_fast_thr=7,
                 mask_rects=()):
        self.max_num_keypts = int(max_num_keypts)
        self.scale_factor = float(scale_factor)
        self.num_levels = int(num_levels)
        self.ini_fast_thr = int(ini_fast_thr)
        self.min_fast_thr = int(min_fast_thr)
        self.mask_rects = [tuple(map(float, r)) for r in mask_rects]   # [x_min, x_max, y_min, y_max] ratios
        for r in self.mask_rects:
            if len(r) != 4 or r[0] >= r[1] or r[2] >= r[3]:
                raise ValueError("mask rectangle must be [x_min, x_max, y_min, y_max] with min < max")

    def _c(self):
        return _lib.OrbParams(self.max_num_keypts, self.scale_factor, self.num_levels, self.ini_fast_thr, self.min_fast_thr)


class orb_extractor:
    """feature::orb_extractor. `extract(image, mask)` returns (keypts, descriptors) instead of filling out-arguments.

    max_rows/max_cols/max_batch size the device buffers once (upstream allocates lazily per frame; HBM is plentiful)."""

    def __init__(self, params=None, max_rows=1080, max_cols=1920, max_batch=1, device=0):
        self.orb_params_ = params or orb_params()
        self._L = _lib.lib()
        _lib.require_device()
        self._h = None
        self._shape = (max_rows, max_cols, max_batch, device)
        self.max_batch = max_batch
        self._variants = {}
        self._fast_split = None
        self._pipeline = None
        self._initialize()

    def _initialize(self):
        """orb_extractor::initialize(): (re)build the tables and the device state from orb_params_ (upstream's setters call it too)."""
        if self._h:
            self._L.ovs_orb_destroy(self._h)
            self._h = None
        max_rows, max_cols, max_batch, device = self._shape
        h = C.c_void_p()
        cp = self.orb_params_._c()
        _lib.check(self._L.ovs_orb_create(C.byref(cp), max_rows, max_cols, max_batch, device, C.byref(h)), "ovs_orb_create")
        self._h = h
        for idx, value in self._variants.items():   # the rule switches and the schedule choices survive a re-initialisation
            _lib.check(self._L.ovs_orb_set_variant(self._h, idx, value), "ovs_orb_set_variant")
        if self._fast_split is not None:
            _lib.check(self._L.ovs_orb_set_fast_split(self._h, self._fast_split), "ovs_orb_set_fast_split")
        if self._pipeline is not None:
            _lib.check(self._L.ovs_orb_set_pipeline(self._h, self._pipeline), "ovs_orb_set_pipeline")
        if getattr(self, "_pyramid_chain", None) is not None:
            _lib.check(self._L.ovs_orb_set_pyramid_chain(self._h, self._pyramid_chain), "ovs_orb_set_pyramid_chain")
        self.max_keypoints = self._L.ovs_orb_max_keypoints(self._h)
        n = self.orb_params_.num_levels
        self.scale_factors_ = np.zeros(n, np.float32)
        self.inv_scale_factors_ = np.zeros(n, np.float32)
        self.level_sigma_sq_ = np.zeros(n, np.float32)
        self.inv_level_sigma_sq_ = np.zeros(n, np.float32)
        self.num_keypts_per_level_ = np.zeros(n, np.int32)
        _lib.check(self._L.ovs_orb_tables(self._h, _p(self.scale_factors_), _p(self.inv_scale_factors_), _p(self.level_sigma_sq_),
                                          _p(self.inv_level_sigma_sq_), _p(self.num_keypts_per_level_)), "ovs_orb_tables")
        self._rect_mask = None

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.ovs_orb_destroy(self._h)
            self._h = None

    # upstream getters / setters (every setter re-initialises, as upstream's do)
    def get_max_num_keypoints(self):
        return self.orb_params_.max_num_keypts

    def set_max_num_keypoints(self, max_num_keypts):
        self.orb_params_.max_num_keypts = int(max_num_keypts)
        self._initialize()

    def set_scale_factor(self, scale_factor):
        self.orb_params_.scale_factor = float(scale_factor)
        self._initialize()

    def set_num_scale_levels(self, num_levels):
        self.orb_params_.num_levels = int(num_levels)
        self._initialize()

    def set_initial_fast_threshold(self, initial_fast_threshold):
        self.orb_params_.ini_fast_thr = int(initial_fast_threshold)
        self._initialize()

    def set_minimum_fast_threshold(self, minimum_fast_threshold):
        self.orb_params_.min_fast_thr = int(minimum_fast_threshold)
        self._initialize()

    def get_scale_factor(self):
        return self.orb_params_.scale_factor

    def get_num_scale_levels(self):
        return self.orb_params_.num_levels

    def get_initial_fast_threshold(self):
        return self.orb_params_.ini_fast_thr

    def get_minimum_fast_threshold(self):
        return self.orb_params_.min_fast_thr

    def get_scale_factors(self):
        return self.scale_factors_.copy()

    def get_inv_scale_factors(self):
        return self.inv_scale_factors_.copy()

    def get_level_sigma_sq(self):
        return self.level_sigma_sq_.copy()

    def get_inv_level_sigma_sq(self):
        return self.inv_level_sigma_sq_.copy()

    def create_rectangle_mask(self, cols, rows):
        """orb_extractor::create_rectangle_mask: 255 everywhere, 0 inside every mask rectangle (ratios of the image size)."""
        if self._rect_mask is None or self._rect_mask.shape != (rows, cols):
            m = np.full((rows, cols), 255, np.uint8)
            for (x0, x1, y0, y1) in self.orb_params_.mask_rects:
                m[int(rows * y0):int(rows * y1), int(cols * x0):int(cols * x1)] = 0
            self._rect_mask = m
        return self._rect_mask

    def extract(self, image, mask=None):
        image = np.ascontiguousarray(image)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise TypeError("image must be CV_8UC1 (upstream asserts image.type() == CV_8UC1)")
        if image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        if mask is None and self.orb_params_.mask_rects:
            mask = self.create_rectangle_mask(image.shape[1], image.shape[0])
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
            if mask.shape != image.shape:
                raise ValueError("mask must have the image's size")
        cap = self.max_keypoints
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        _lib.check(self._L.ovs_orb_extract(self._h, _p(image), image.shape[0], image.shape[1], image.strides[0], _p(mask),
                                           mask.strides[0] if mask is not None else 0, _p(kps), _p(desc), cap, C.byref(n)),
                   "ovs_orb_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_pair(self, left, right, mask_left=None, mask_right=None):
        """A stereo rig's left and right image in one call (ovs_orb_extract_pair; the extractor needs max_batch >= 2): the results of two
        extract() calls, bit for bit, at the launch count of one. Returns ((kps, desc) left, (kps, desc) right)."""
        left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
        if left.dtype != np.uint8 or left.ndim != 2 or right.dtype != np.uint8 or right.shape != left.shape:
            raise TypeError("both images must be CV_8UC1 of one size")
        if (mask_left is None) != (mask_right is None):
            raise ValueError("masks: both or neither")
        if mask_left is not None:
            mask_left, mask_right = np.ascontiguousarray(mask_left, np.uint8), np.ascontiguousarray(mask_right, np.uint8)
            if mask_left.shape != left.shape or mask_right.shape != left.shape:
                raise ValueError("masks must have the images' size")
        cap = self.max_keypoints
        out = [(np.zeros(cap, KP_DTYPE), np.zeros((cap, 32), np.uint8), C.c_int32(0)) for _ in range(2)]
        _lib.check(self._L.ovs_orb_extract_pair(self._h, _p(left), _p(right), left.shape[0], left.shape[1], left.strides[0], _p(mask_left),
                                                _p(mask_right), mask_left.strides[0] if mask_left is not None else 0, _p(out[0][0]),
                                                _p(out[0][1]), C.byref(out[0][2]), _p(out[1][0]), _p(out[1][1]), C.byref(out[1][2]), cap),
                   "ovs_orb_extract_pair")
        return tuple((k[:n.value].copy(), d[:n.value].copy()) for k, d, n in out)

    def set_fast_split(self, enable):
        """Level-0 FAST beside the pyramid on an internal stream (default on); off = one FAST launch after the pyramid."""
        _lib.check(self._L.ovs_orb_set_fast_split(self._h, 1 if enable else 0), "ovs_orb_set_fast_split")
        self._fast_split = 1 if enable else 0

    def set_pyramid_chain(self, max_frames):
        """All pyramid levels in ONE launch for calls of at most max_frames frames (default 2: the tracker's single frame or stereo pair); 0 / False = level-by-level
        launches always."""
        _lib.check(self._L.ovs_orb_set_pyramid_chain(self._h, int(max_frames)), "ovs_orb_set_pyramid_chain")
        self._pyramid_chain = int(max_frames)

    def set_variant(self, which, value):
        """ovs_orb_set_variant: "tree_switch_factor" (3 | 1), "tree_tie_order" (0 later-created first | 1 earlier first), "blur_taps" (0 | 1) --
        the rules of oracle/ORACLE_SPEC.md (6, 7, 10) that cannot be pinned without upstream's sources, as run-time choices."""
        idx = {"tree_switch_factor": 0, "tree_tie_order": 1, "blur_taps": 2, "trig": 3}[which]
        _lib.check(self._L.ovs_orb_set_variant(self._h, idx, int(value)), "ovs_orb_set_variant")
        self._variants[idx] = int(value)
        self.max_keypoints = self._L.ovs_orb_max_keypoints(self._h)   # tree_switch_factor = 1 can return up to 2 N per level

    def set_pipeline(self, n_sub):
        """Issue the device-batch extract as n_sub overlapping sub-batches on internal streams (ovs_orb_set_pipeline)."""
        _lib.check(self._L.ovs_orb_set_pipeline(self._h, int(n_sub)), "ovs_orb_set_pipeline")
        self._pipeline = int(n_sub)

    def extract_batch_dev(self, d_images, d_kps, d_desc, d_counts, stream=None, d_masks=None):
        """Device-resident batched extract. d_images: torch uint8 CUDA tensor (B, rows, cols) contiguous (cols % 4 == 0);
        outputs: d_kps (B, cap, 7) float32/int32 raw 28-byte records, d_desc (B, cap, 32) uint8, d_counts (B,) int32."""
        B, rows, cols = d_images.shape
        cap = d_desc.shape[1]
        _lib.check(self._L.ovs_orb_extract_batch_dev(self._h, d_images.data_ptr(), B, rows, cols, d_images.stride(1),
                                                     d_images.stride(0), d_masks.data_ptr() if d_masks is not None else None,
                                                     d_kps.data_ptr(), d_desc.data_ptr(), d_counts.data_ptr(), cap,
                                                     stream), "ovs_orb_extract_batch_dev")

    # observable state upstream exposes as the public member image_pyramid_
    def image_pyramid(self, level, frame=0):
        r, c = C.c_int32(), C.c_int32()
        _lib.check(self._L.ovs_orb_pyramid_level(self._h, frame, level, None, C.byref(r), C.byref(c)), "ovs_orb_pyramid_level")
        out = np.zeros((r.value, c.value), np.uint8)
        _lib.check(self._L.ovs_orb_pyramid_level(self._h, frame, level, _p(out), C.byref(r), C.byref(c)), "ovs_orb_pyramid_level")
        return out

    def debug_candidates(self, level, frame=0, cap=1 << 20):
        xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int32()
        _lib.check(self._L.ovs_orb_debug_candidates(self._h, frame, level, _p(xs), _p(ys), _p(sc), cap, C.byref(n)),
                   "ovs_orb_debug_candidates")
        return xs[:n.value].copy(), ys[:n.value].copy(), sc[:n.value].copy()

    def debug_level_counts(self, frame=0):
        c = np.zeros(self.orb_params_.num_levels, np.int32)
        _lib.check(self._L.ovs_orb_debug_level_counts(self._h, frame, _p(c)), "ovs_orb_debug_level_counts")
        return c
