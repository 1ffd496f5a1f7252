"""ctypes loader for libovslam_hip.so (the C ABI declared in include/ovslam_hip.h).

There is deliberately NO fallback: if the library has not been built (python -c "import __graft_entry__ as g; g.build()"
or make -C openvslam_amd/csrc) or no HIP device is usable, importing / calling raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OVS_LIB_PATH") or os.path.join(_HERE, "libovslam_hip.so")   # (OVS_LIB_PATH: A/B builds of the library, tools/)

OVS_OK = 0
STATUS_NAMES = {0: "OVS_OK", -1: "OVS_ERR_INVALID", -2: "OVS_ERR_NO_DEVICE", -3: "OVS_ERR_HIP", -4: "OVS_ERR_CAPACITY",
                -5: "OVS_ERR_ALIGN"}


class OvsError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = _lib.ovs_last_error().decode() if _lib is not None else ""
        super().__init__("%s failed: %s%s" % (where, STATUS_NAMES.get(status, status), (" (" + msg + ")") if msg else ""))


class OrbParams(C.Structure):
    """feature::orb_params (expected: src/openvslam/feature/orb_params.h)."""
    _fields_ = [("max_num_keypts", C.c_int32), ("scale_factor", C.c_float), ("num_levels", C.c_int32),
                ("ini_fast_thr", C.c_int32), ("min_fast_thr", C.c_int32)]


_lib = None

# every symbol include/ovslam_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _sz, _f = C.c_void_p, C.c_int32, C.c_size_t, C.c_float
SYMBOLS = {
    "ovs_last_error": (C.c_char_p, []),
    "ovs_device_count": (_i32, []),
    "ovs_device_arch": (_i32, [_i32, C.c_char_p, _sz]),
    "ovs_orb_create": (_i32, [C.POINTER(OrbParams), _i32, _i32, _i32, _i32, C.POINTER(_vp)]),
    "ovs_orb_destroy": (_i32, [_vp]),
    "ovs_orb_device": (_i32, [_vp]),
    "ovs_orb_tables": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "ovs_orb_max_keypoints": (_i32, [_vp]),
    "ovs_debug_inject_hip_failures": (_i32, [_i32, _i32]),
    "ovs_orb_extract": (_i32, [_vp, _vp, _i32, _i32, _sz, _vp, _sz, _vp, _vp, _i32, C.POINTER(_i32)]),
    "ovs_orb_extract_pair": (_i32, [_vp, _vp, _vp, _i32, _i32, _sz, _vp, _vp, _sz, _vp, _vp, C.POINTER(_i32), _vp, _vp, C.POINTER(_i32), _i32]),
    "ovs_orb_extract_submit": (_i32, [_vp, _vp, _i32, _i32, _sz, _vp, _sz]),
    "ovs_orb_extract_collect": (_i32, [_vp, _vp, _vp, _i32, C.POINTER(_i32)]),
    "ovs_orb_set_host_mode": (_i32, [_vp, _i32]),
    "ovs_orb_host_profile_read": (_i32, [_vp, _vp]),
    "ovs_orb_set_host_pyramid": (_i32, [_vp, _i32]),
    "ovs_orb_host_pyramid_level": (_i32, [_vp, _i32, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "ovs_orb_extract_batch_dev": (_i32, [_vp, _vp, _i32, _i32, _i32, _sz, _sz, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ovs_orb_set_pipeline": (_i32, [_vp, _i32]),
    "ovs_orb_set_fast_split": (_i32, [_vp, _i32]),
    "ovs_orb_set_pyramid_chain": (_i32, [_vp, _i32]),
    "ovs_orb_set_variant": (_i32, [_vp, _i32, _i32]),
    "ovs_orb_profile_read_aux": (_i32, [_vp, C.POINTER(_f), C.POINTER(_i32)]),
    "ovs_orb_pyramid_level": (_i32, [_vp, _i32, _i32, _vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "ovs_orb_debug_candidates": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, C.POINTER(_i32)]),
    "ovs_orb_debug_level_counts": (_i32, [_vp, _i32, _vp]),
    "ovs_orb_profile_enable": (_i32, [_vp, _i32]),
    "ovs_orb_profile_read": (_i32, [_vp, _vp, C.POINTER(_i32)]),
    "ovs_matcher_set_near_path": (_i32, [_vp, _i32]),
    "ovs_matcher_profile_enable": (_i32, [_vp, _i32]),
    "ovs_matcher_profile_read": (_i32, [_vp, _vp, C.POINTER(_i32)]),
    "ovs_matcher_create": (_i32, [_i32, _i32, _i32, _i32, C.POINTER(_vp)]),
    "ovs_matcher_destroy": (_i32, [_vp]),
    "ovs_robust_brute_force_match": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _f, _vp, _i32, C.POINTER(_i32)]),
    "ovs_robust_brute_force_match_batch_dev": (_i32, [_vp, _vp, _sz, _vp, _vp, _vp, _sz, _vp, _vp, _i32, _f, _vp, _vp, _i32, _vp]),
    "ovs_ba_linearize": (_i32, [_i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ovs_ba_linearize_dev": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ovs_ba_linearize_equirect": (_i32, [_i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ovs_ba_linearize_equirect_dev": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ovs_ba_graph_create": (_i32, [_i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, C.c_double, C.POINTER(_vp)]),
    "ovs_ba_graph_create_equirect": (_i32, [_i32, _i32, _vp, _i32, _vp, _i32, _i32, _i32, C.POINTER(_vp)]),
    "ovs_ba_graph_destroy": (_i32, [_vp]),
    "ovs_ba_pool_trim": (_i32, []),
    "ovs_ba_graph_linearize_dev": (_i32, [_vp, _vp, _vp, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ovs_frame_dev_create": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, C.POINTER(_vp)]),
    "ovs_frame_dev_destroy": (_i32, [_vp]),
    "ovs_frame_dev_num_keypoints": (_i32, [_vp]),
    "ovs_projection_match_frame_and_landmarks_f": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _f, _f, _vp, C.POINTER(_i32)]),
    "ovs_area_match_in_consistent_area_f": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _f, _i32, C.POINTER(_i32)]),
    "ovs_projection_match_current_and_last_frames_f": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _f, _i32, _vp,
                                                               C.POINTER(_i32)]),
    "ovs_frame_dev_attach_bearings": (_i32, [_vp, _vp]),
    "ovs_frame_dev_device": (_i32, [_vp]),
    # keyframe-side twins (round 4): an ovs_frame_dev where the host form takes (gp, kps, desc, [x_right], n)
    "ovs_bow_match_frame_and_keyframe_f": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _f, _i32, _vp, C.POINTER(_i32)]),
    "ovs_bow_match_keyframes_f": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _f, _i32, _vp, C.POINTER(_i32)]),
    "ovs_robust_match_for_triangulation_f": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp,
                                                    C.POINTER(_i32)]),
    "ovs_fuse_replace_duplication_f": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _f, _f, _vp, C.POINTER(_i32)]),
    "ovs_fuse_detect_duplication_f": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _f, _f, _vp, C.POINTER(_i32)]),
    "ovs_projection_match_frame_and_keyframe_f": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _f, _f, C.c_uint32, _i32,
                                                         _vp, C.POINTER(_i32)]),
    "ovs_projection_match_by_sim3_transform_f": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _f, _f, _vp,
                                                        C.POINTER(_i32)]),
    "ovs_projection_match_keyframes_mutually_f": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _vp, _vp,
                                                         _vp, _i32, _f, _f, _vp, C.POINTER(_i32)]),
    "ovs_ba_multi_create": (_i32, [_i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, C.c_double, C.POINTER(_vp)]),
    "ovs_ba_multi_destroy": (_i32, [_vp]),
    "ovs_ba_multi_set_exchange": (_i32, [_vp, _i32]),
    "ovs_ba_multi_linearize": (_i32, [_vp, _vp, _vp, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ovs_vocab_create": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "ovs_vocab_destroy": (_i32, [_vp]),
    "ovs_vocab_tree_load": (_i32, [C.c_char_p, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "ovs_vocab_tree_arrays": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "ovs_vocab_tree_free": (_i32, [_vp]),
    "ovs_vocab_load_file": (_i32, [_i32, C.c_char_p, _i32, C.POINTER(_vp), C.POINTER(_i32)]),
    "ovs_bow_transform": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "ovs_bow_transform_dev": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ovs_local_ba_optimize": (_i32, [_i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, C.c_double, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ovs_local_ba_optimize_equirect": (_i32, [_i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ovs_ba_linearize_stereo": (_i32, [_i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ovs_ba_linearize_stereo_dev": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, C.c_double, C.c_double, _i32, _vp, _vp, _vp, _vp, _vp,
                                           _vp, _vp]),
    "ovs_pose_set_variant": (_i32, [_i32, _i32]),
    "ovs_local_ba_set_solver": (_i32, [_i32]),
    "ovs_local_ba_get_solver": (_i32, []),
    "ovs_ba_dense_solve": (_i32, [_i32, _vp, _vp, _i32, _vp]),
    "ovs_pose_optimize": (_i32, [_i32, _vp, _vp, _i32, _vp, C.c_double, _i32, _vp, _vp, C.POINTER(_i32)]),
    "ovs_pose_optimize_batch_dev": (_i32, [_vp, _vp, _vp, _i32, _vp, C.c_double, _i32, _vp, _vp, _vp, _vp]),
    "ovs_pose_optimize_equirect": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, C.POINTER(_i32)]),
    "ovs_pose_optimize_equirect_batch_dev": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ovs_hamming_best2": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "ovs_wmatcher_create": (_i32, [_i32, _i32, _i32, _i32, C.POINTER(_vp)]),
    "ovs_wmatcher_destroy": (_i32, [_vp]),
    "ovs_match_set_variant": (_i32, [_i32, _i32]),
    "ovs_match_get_variant": (_i32, [_i32]),
    "ovs_assign_keypoints_to_grid": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, C.POINTER(_i32)]),
    "ovs_grid_assign_dev": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "ovs_projection_match_frame_and_landmarks": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _f, _f,
                                                        _vp, C.POINTER(_i32)]),
    "ovs_projection_match_frame_and_landmarks_dev": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _f,
                                                            _f, _vp, _vp, _vp]),
    "ovs_area_match_in_consistent_area": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _f, _i32, C.POINTER(_i32)]),
    "ovs_area_match_in_consistent_area_dev": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _f, _i32, _vp, _vp]),
    "ovs_bow_match_frame_and_keyframe": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _f, _i32,
                                                _vp, C.POINTER(_i32)]),
    "ovs_projection_match_current_and_last_frames": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                                            _i32, _f, _i32, _vp, C.POINTER(_i32)]),
    "ovs_projection_match_frame_and_keyframe": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _f, _f,
                                                       C.c_uint32, _i32, _vp, C.POINTER(_i32)]),
    "ovs_robust_match_for_triangulation": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                                  _vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp, C.POINTER(_i32)]),
    "ovs_bow_match_keyframes": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _f, _i32, _vp,
                                       C.POINTER(_i32)]),
    "ovs_fuse_replace_duplication": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _f, _f, _vp,
                                            C.POINTER(_i32)]),
    "ovs_fuse_detect_duplication": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _f, _f, _vp,
                                           C.POINTER(_i32)]),
    "ovs_projection_match_by_sim3_transform": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _f, _f,
                                                      _vp, C.POINTER(_i32)]),
    "ovs_projection_match_keyframes_mutually": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                                       _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _i32, _f, _f, _vp, C.POINTER(_i32)]),
    "ovs_stereo_create": (_i32, [_i32, _i32, _i32, C.POINTER(_vp)]),
    "ovs_stereo_destroy": (_i32, [_vp]),
    "ovs_stereo_set_variant": (_i32, [_vp, _i32, _i32]),
    "ovs_stereo_compute": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _f, _f, _vp, _vp, C.POINTER(_i32)]),
    "ovs_detmath_eval": (_i32, [_i32, _i32, _vp, _vp, _vp, _i32]),
    "ovs_stereo_compute_dev": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _f, _f, _vp, _vp, _vp, _vp]),
}


class Camera(C.Structure):
    """camera::base subset (ovs_camera). model: 0 perspective, 1 equirectangular; setup: 0 mono, 1 stereo, 2 RGBD."""
    _fields_ = [("model", C.c_int32), ("setup", C.c_int32), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("focal_x_baseline", C.c_double), ("true_baseline", C.c_double), ("cols", C.c_int32), ("rows", C.c_int32)]


class GridParams(C.Structure):
    """camera::base::img_bounds_ + num_grid_cols_/num_grid_rows_ (ovs_grid_params)."""
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float), ("cols", C.c_int32),
                ("rows", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                              "There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7; importing it first makes the dynamic linker
        # bind this library to the same runtime (same soname), so torch tensors / streams and our kernels share one context.
        # (A C++ host without torch binds to /opt/rocm's runtime through the library's RUNPATH.)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, where):
    if status != OVS_OK:
        raise OvsError(status, where)


def require_device():
    n = lib().ovs_device_count()
    if n < 1:
        raise OvsError(-2, "ovs_device_count")
    return n
