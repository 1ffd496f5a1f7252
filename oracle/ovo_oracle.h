/*
 * ovo_oracle.h -- CPU ORACLE for the OpenVSLAM per-frame hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library, and
 * only as the checker / reported CPU baseline.  Nothing under openvslam_amd/ links or calls it.
 *
 * PARITY UNPINNED.  /root/reference holds nothing but upstream's takedown notice
 * (/root/reference/README.md:1-4): no source, no tests, no golden vectors, and none of the third-party
 * code whose arithmetic the path uses (OpenCV resize/FAST/GaussianBlur/fastAtan2/cvRound, g2o) exists in
 * the build container.  Every function below is therefore a FROM-SPEC scalar restatement that follows the
 * algorithm description in SURVEY.md section 8(a) (rows A0-A9, M1-M8, D1, B1-B4) plus the published
 * algorithm definitions (Rosten & Drummond FAST-9/16 as implemented by OpenCV fast.cpp/fast_score.cpp;
 * Rublee et al. ORB; Mur-Artal et al. ORB-SLAM2 DistributeOctTree).  Each function cites the *expected*
 * upstream path (no line numbers can exist).  Where upstream behaviour is implementation-defined
 * (std::sort tie order on node pointers, OpenCV-version-dependent blur arithmetic) the rule chosen here is
 * written next to the code and in ORACLE_SPEC.md.  All parity claims in this repo are
 * "bit-exact vs. this oracle", never "vs. OpenVSLAM".
 *
 * Build: make -C oracle   (g++ -O2 -ffp-contract=off -fopenmp) -> oracle/liboracle.so
 */
#ifndef OVO_ORACLE_H
#define OVO_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Layout-compatible with cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id): 28 bytes. */
typedef struct ovo_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} ovo_keypoint;

/* feature::orb_params (expected: src/openvslam/feature/orb_params.{h,cc}); defaults 2000/1.2/8/20/7. */
typedef struct ovo_orb_params {
    int32_t max_num_keypts;
    float scale_factor;
    int32_t num_levels;
    int32_t ini_fast_thr;
    int32_t min_fast_thr;
} ovo_orb_params;

#define OVO_MAX_LEVELS 16

/* A0: tables. Arrays must hold num_levels entries (u_max: 16). */
int ovo_orb_tables(const ovo_orb_params* p, float* scale_factors, float* inv_scale_factors,
                   float* level_sigma_sq, float* inv_level_sigma_sq, int32_t* num_keypts_per_level,
                   int32_t* u_max16);
/* A1: level sizes from the ORIGINAL image size. */
int ovo_pyramid_sizes(const ovo_orb_params* p, int rows, int cols, int32_t* level_rows, int32_t* level_cols);
/* A1: cv::resize(..., INTER_LINEAR) for CV_8UC1, 11-bit fixed-point coefficients. */
int ovo_resize_linear_u8(const uint8_t* src, int srows, int scols, size_t sstride, uint8_t* dst, int drows,
                         int dcols, size_t dstride);
/* A3: cv::FAST(img, kps, thr, nonmax=true, TYPE_9_16) on a (sub-)image. Returns count (<= cap written). */
int ovo_fast9_16(const uint8_t* img, int rows, int cols, size_t stride, int threshold, int nonmax,
                 int32_t* xs, int32_t* ys, int32_t* scores, int cap);
/* A4: quad-tree distribution. Inputs are candidate coordinates RELATIVE to (min_x,min_y) and responses, in
 * emission order. Writes the indices (into the candidate arrays) of the selected keypoints in node-list
 * order. Returns count. */
int ovo_distribute_via_tree(const float* xs, const float* ys, const float* responses, int n, int min_x,
                            int max_x, int min_y, int max_y, int num_keypts, int32_t* out_idx, int cap);
/* A5: ic_angle + cv::fastAtan2 (degrees). */
float ovo_fast_atan2(float y, float x);
/* include/ovs_detmath.h evaluated on the host: fn 0 logf(float a) 1 asin 2 acos 3 atan2(a, b) */
int ovo_detmath_eval(int fn, const double* a, const double* b, double* out, int n);
long long ovo_detmath_logf_vs_libm(uint32_t first_bits, uint32_t last_bits);
float ovo_ic_angle(const uint8_t* img, size_t stride, int x, int y, const int32_t* u_max16);
/* A6: 7x7 sigma=2 Gaussian, 8.8 fixed point, BORDER_REFLECT_101. */
int ovo_gaussian_blur_7x7(const uint8_t* src, int rows, int cols, size_t sstride, uint8_t* dst, size_t dstride);
int ovo_gaussian_blur_7x7_v(const uint8_t* src, int rows, int cols, size_t sstride, uint8_t* dst, size_t dstride, int taps_variant);
int ovo_distribute_via_tree_v(const float* xs, const float* ys, const float* responses, int n, int min_x, int max_x, int min_y, int max_y,
                              int num_keypts, int switch_factor, int tie_earlier_first, int32_t* out_idx, int cap);
/* A7: util::cos / util::sin polynomial and one steered-BRIEF descriptor (32 bytes). */
float ovo_util_cos(float v);
float ovo_util_sin(float v);
int ovo_orb_descriptor(const uint8_t* blurred, size_t stride, int x, int y, float angle_deg, uint8_t* desc32);
int ovo_orb_descriptor_v(const uint8_t* blurred, size_t stride, int x, int y, float angle_deg, uint8_t* desc32, int trig_variant);
long ovo_trig_mismatches_vs_libm(uint32_t lo_bits, uint32_t hi_bits);
long ovo_deg2rad_mismatches(uint32_t lo_bits, uint32_t hi_bits);
float ovo_det_sinf(float v);
float ovo_det_cosf(float v);
const int8_t* ovo_orb_pattern(void); /* 256*4 int8 */

/* A9: whole extractor with observable intermediates. */
typedef struct ovo_orb ovo_orb;
ovo_orb* ovo_orb_create(const ovo_orb_params* p);
void ovo_orb_destroy(ovo_orb* h);
void ovo_orb_set_threads(ovo_orb* h, int n); /* OpenMP threads over levels (upstream USE_OPENMP shape) */
/* ORACLE_SPEC rules 6, 7, 10 as run-time variants: which 0 = quad-tree switch factor (3 default | 1), 1 = equal-count tie order of the sorted
 * phase (0 later-created node first, default | 1 earlier-created first), 2 = blur taps (0 = 18,34,48,56 error-diffused, default | 1 = 18,34,49,55
 * independently rounded, saturating). Returns -1 for an unknown pair. */
int ovo_orb_set_variant(ovo_orb* h, int which, int value);
/* mask: NULL or rows x cols u8 (0 = masked out). Returns 0, writes *n_out (<= cap). */
int ovo_orb_extract(ovo_orb* h, const uint8_t* img, int rows, int cols, size_t stride, const uint8_t* mask,
                    size_t mask_stride, ovo_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* observables of the last extract */
int ovo_orb_level_size(const ovo_orb* h, int level, int* rows, int* cols);
const uint8_t* ovo_orb_level_image(const ovo_orb* h, int level);   /* contiguous rows*cols */
const uint8_t* ovo_orb_level_blurred(const ovo_orb* h, int level); /* contiguous rows*cols (empty level: NULL) */
int ovo_orb_level_num_candidates(const ovo_orb* h, int level);
/* candidates in emission order: level-image coordinates (x,y) and FAST score */
int ovo_orb_level_candidates(const ovo_orb* h, int level, int32_t* xs, int32_t* ys, int32_t* scores, int cap);
int ovo_orb_level_num_keypts(const ovo_orb* h, int level);

/* ---- matchers (ovo_match.cc) ---- */
#define OVO_HAMMING_DIST_THR_LOW 50
#define OVO_HAMMING_DIST_THR_HIGH 100
#define OVO_MAX_HAMMING_DIST 256
/* M1: match::base::compute_descriptor_distance_32 (8 x u32 parallel bit count). */
uint32_t ovo_descriptor_distance_32(const uint8_t* a, const uint8_t* b);
/* M2: match::robust::brute_force_match(frame, keyframe, matches).
 *   desc_frm: n_frm x 32 (frame = idx_1 side), desc_kf: n_kf x 32 (keyframe = idx_2 side),
 *   kf_valid: NULL or n_kf bytes (landmark present and not will_be_erased); frm_valid: NULL or n_frm bytes, 0 = the frame
 *   keypoint is skipped by the inner loop (optional frame-side mask, ORACLE_SPEC rule 14).
 *   Output pairs (idx_1, idx_2) in emission order. Returns number of matches. */
int ovo_robust_brute_force_match(const uint8_t* desc_frm, int n_frm, const uint8_t* frm_valid, const uint8_t* desc_kf, int n_kf,
                                 const uint8_t* kf_valid, float lowe_ratio, int32_t* pairs, int cap);
/* Unconstrained per-query best / second best (first-seen tie rule) over all valid targets. */
int ovo_hamming_best2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* t_valid,
                      int32_t* best_idx, uint16_t* best, uint16_t* second);

/* ---- windowed matchers and their candidate generator (ovo_match2.cc) ---- */
/* camera::base::img_bounds_ + num_grid_cols_/num_grid_rows_ (64 x 48 upstream). */
typedef struct ovo_grid_params {
    float min_x, min_y, max_x, max_y;
    int32_t cols, rows;
} ovo_grid_params;
/* D1 assign_keypoints_to_grid as CSR: cell id = cx*rows + cy, cell_start[cols*rows+1], items = keypoint indices. Returns #items. */
int ovo_assign_keypoints_to_grid(const ovo_grid_params* p, const float* xs, const float* ys, int n, int32_t* cell_start,
                                 int32_t* items);
/* D1 get_keypoints_in_cell, indices in upstream's order. Returns the count (<= cap written). */
int ovo_get_keypoints_in_cell(const ovo_grid_params* p, const float* xs, const float* ys, const int32_t* octaves, int n, float ref_x,
                              float ref_y, float margin, int min_level, int max_level, int32_t* out, int cap);
/* match::angle_checker<int>: invalid[i] = 1 iff match i falls outside the 3 fullest of the 30 bins. */
int ovo_match_set_variant(int which, int value);   /* which 0: angle_checker drops bins below 0.1 x the fullest (0 | 1), process-wide */
void ovo_angle_checker_invalid(const float* delta_angles, int n, uint8_t* invalid);
/* M3 projection::match_frame_and_landmarks. assigned[l] = frame keypoint index or -1. Returns num_matches. */
int ovo_projection_match_frame_and_landmarks(const ovo_grid_params* gp, const float* xs, const float* ys, const int32_t* octaves,
                                             const float* stereo_x_right, const uint8_t* desc, const uint8_t* occupied, int n,
                                             const float* lm_x, const float* lm_y, const float* lm_x_right, const int32_t* lm_level,
                                             const uint8_t* lm_desc, const uint8_t* lm_valid, int m, const float* scale_factors,
                                             float margin, float lowe_ratio, int32_t* assigned);
/* M5 area::match_in_consistent_area. prev_matched_xy (n1 x 2) is updated in place; matched_2_in_1[n1]. Returns num_matches. */
int ovo_area_match_in_consistent_area(const ovo_grid_params* gp, const int32_t* octaves_1, const float* angles_1,
                                      const uint8_t* desc_1, int n1, const float* xs_2, const float* ys_2, const int32_t* octaves_2,
                                      const float* angles_2, const uint8_t* desc_2, int n2, float* prev_matched_xy,
                                      int32_t* matched_2_in_1, int margin, float lowe_ratio, int check_orientation);
/* M7 bow_tree::match_frame_and_keyframe. Feature vectors as CSR over ascending node ids. matched_kf_in_frm[n_frm]. */
int ovo_bow_match_frame_and_keyframe(const uint8_t* kf_desc, const float* kf_angles, const uint8_t* kf_valid, int n_kf,
                                     const int32_t* kf_node_ids, const int32_t* kf_node_start, const int32_t* kf_items, int kf_nodes,
                                     const uint8_t* frm_desc, const float* frm_angles, int n_frm, const int32_t* frm_node_ids,
                                     const int32_t* frm_node_start, const int32_t* frm_items, int frm_nodes, float lowe_ratio,
                                     int check_orientation, int32_t* matched_kf_in_frm);
/* camera::base subset used by the matchers that reproject inside the call (M4). model: 0 = perspective, 1 = equirectangular;
 * setup: 0 = monocular, 1 = stereo, 2 = RGBD. */
typedef struct ovo_camera {
    int32_t model, setup;
    double fx, fy, cx, cy;
    double focal_x_baseline, true_baseline;
    int32_t cols, rows;
} ovo_camera;
/* camera::{perspective,equirectangular}::reproject_to_image(rot_cw, trans_cw, pos_w, reproj, x_right). pose_cw = 12 doubles
 * (rotation row-major, then translation). Returns 1 if in the image. */
int ovo_reproject_to_image(const ovo_camera* cam, const ovo_grid_params* bounds, const double* pose_cw, const double* pos_w,
                           double* reproj_xy, float* x_right);
/* M4 projection::match_current_and_last_frames(curr_frm, last_frm, margin). last_valid[i] != 0 iff last_frm.landmarks_[i] &&
 * !last_frm.outlier_flags_[i]. assigned[i] = current-frame keypoint that receives last_frm.landmarks_[i], or -1. */
int ovo_projection_match_current_and_last_frames(const ovo_camera* cam, const ovo_grid_params* gp, const float* xs, const float* ys,
                                                 const int32_t* octaves, const float* angles, const float* stereo_x_right,
                                                 const uint8_t* desc, const uint8_t* occupied, int n_curr, const double* pose_cw_curr,
                                                 const int32_t* last_octaves, const float* last_angles, const double* last_pos_w,
                                                 const uint8_t* last_lm_desc, const uint8_t* last_valid, int n_last,
                                                 const double* pose_cw_last, const float* scale_factors, int num_scale_levels,
                                                 float margin, int check_orientation, int32_t* assigned);
/* M8 fuse::replace_duplication(keyfrm, landmarks_to_check, margin) -- the candidate search (the landmark-graph surgery that follows
 * stays on the host). lm_valid[l] != 0 iff lm && !will_be_erased() && !is_observed_in_keyframe(keyfrm). lm_dist = (min, max) valid
 * distances; lm_normal = obs_mean_normal. best_idx[l] = keyframe keypoint the landmark fuses with, or -1. Returns num_fused. */
int ovo_fuse_replace_duplication(const ovo_camera* cam, const ovo_grid_params* gp, const float* xs, const float* ys,
                                 const int32_t* octaves, const float* stereo_x_right, const uint8_t* desc, int n, const double* pose_cw,
                                 const double* lm_pos_w, const float* lm_dist_min_max, const double* lm_normal, const uint8_t* lm_desc,
                                 const uint8_t* lm_valid, int m, const float* scale_factors, const float* inv_level_sigma_sq,
                                 int num_scale_levels, float log_scale_factor, float margin, int32_t* best_idx);
/* M7 bow_tree::match_keyframes. matched_2_in_1[n1] = keyframe-2 keypoint matched to keyframe-1 keypoint, or -1. */
int ovo_bow_match_keyframes(const uint8_t* desc_1, const float* angles_1, const uint8_t* valid_1, int n1, const int32_t* node_ids_1,
                            const int32_t* node_start_1, const int32_t* items_1, int nodes_1, const uint8_t* desc_2, const float* angles_2,
                            const uint8_t* valid_2, int n2, const int32_t* node_ids_2, const int32_t* node_start_2, const int32_t* items_2,
                            int nodes_2, float lowe_ratio, int check_orientation, int32_t* matched_2_in_1);
/* M4 projection::match_frame_and_keyframe(curr_frm, keyfrm, already_matched_lms, margin, hamm_dist_thr). kf_valid[i] != 0 iff
 * landmarks[i] && !will_be_erased() && !already_matched_lms.count(landmarks[i]); occupied[j] != 0 iff curr_frm.landmarks_[j]. */
int ovo_projection_match_frame_and_keyframe(const ovo_camera* cam, const ovo_grid_params* gp, const float* xs, const float* ys,
                                            const int32_t* octaves, const float* angles, const uint8_t* desc, const uint8_t* occupied,
                                            int n_curr, const double* pose_cw_curr, const float* kf_angles, const double* kf_pos_w,
                                            const float* kf_dist_min_max, const uint8_t* kf_lm_desc, const uint8_t* kf_valid, int n_kf,
                                            const float* scale_factors, int num_scale_levels, float log_scale_factor, float margin,
                                            unsigned hamm_dist_thr, int check_orientation, int32_t* assigned);
/* M2 robust::match_for_triangulation. has_lm_i[k] != 0 iff keyframe i's keypoint k already holds a landmark; x_right_i = stereo_x_right_
 * (NULL = monocular); bearings_i = n x 3 doubles; epipole_in_2 = keyfrm_1's centre as a bearing in keyfrm_2. matched_2_in_1[n1]. */
int ovo_fuse_detect_duplication(const ovo_camera* cam, const ovo_grid_params* gp, const float* xs, const float* ys, const int32_t* octaves,
                                const uint8_t* desc, int n, const double* sim3_cw, const double* lm_pos_w, const float* lm_dist_min_max,
                                const double* lm_normal, const uint8_t* lm_desc, const uint8_t* lm_valid, int m, const float* scale_factors,
                                int num_scale_levels, float log_scale_factor, float margin, int32_t* best_idx_out);
int ovo_projection_match_by_sim3_transform(const ovo_camera* cam, const ovo_grid_params* gp, const float* xs, const float* ys,
                                           const int32_t* octaves, const uint8_t* desc, const uint8_t* occupied, int n, const double* sim3_cw,
                                           const double* lm_pos_w, const float* lm_dist_min_max, const double* lm_normal, const uint8_t* lm_desc,
                                           const uint8_t* lm_valid, int m, const float* scale_factors, int num_scale_levels,
                                           float log_scale_factor, float margin, int32_t* assigned);
int ovo_projection_match_keyframes_mutually(const ovo_camera* cam_1, const ovo_grid_params* gp_1, const float* xs_1, const float* ys_1,
                                            const int32_t* octaves_1, const uint8_t* desc_1, int n1, const double* pose_cw_1,
                                            const double* lm_pos_w_1, const float* lm_dist_1, const uint8_t* lm_desc_1, const uint8_t* lm_valid_1,
                                            const ovo_camera* cam_2, const ovo_grid_params* gp_2, const float* xs_2, const float* ys_2,
                                            const int32_t* octaves_2, const uint8_t* desc_2, int n2, const double* pose_cw_2,
                                            const double* lm_pos_w_2, const float* lm_dist_2, const uint8_t* lm_desc_2, const uint8_t* lm_valid_2,
                                            double s_12, const double* rot_12, const double* trans_12, const float* scale_factors,
                                            int num_scale_levels, float log_scale_factor, float margin, int32_t* matched_2_in_1);
int ovo_bow_transform(int n_nodes, const int32_t* child_start, const int32_t* children, const uint8_t* node_desc, const double* node_weight,
                      const int32_t* node_word_id, int depth, const uint8_t* desc, int n, int levelsup, int32_t* word_id, double* weight,
                      int32_t* node_id);
int ovo_robust_match_for_triangulation(const uint8_t* desc_1, const float* angles_1, const int32_t* octaves_1, const uint8_t* has_lm_1,
                                       const float* x_right_1, const double* bearings_1, int n1, const int32_t* node_ids_1,
                                       const int32_t* node_start_1, const int32_t* items_1, int nodes_1, const uint8_t* desc_2,
                                       const float* angles_2, const uint8_t* has_lm_2, const float* x_right_2, const double* bearings_2,
                                       int n2, const int32_t* node_ids_2, const int32_t* node_start_2, const int32_t* items_2, int nodes_2,
                                       const double* E_12, const double* epipole_in_2, const float* scale_factors, int check_orientation,
                                       int32_t* matched_2_in_1);
/* M6 stereo::compute. Pyramids = the two extractors' image_pyramid_ (unblurred). Returns the number of valid depths. */
int ovo_stereo_compute(const uint8_t* const* pyr_left, const uint8_t* const* pyr_right, const int32_t* level_rows,
                       const int32_t* level_cols, const size_t* stride_left, const size_t* stride_right, int num_levels,
                       const ovo_keypoint* kps_left, const uint8_t* desc_left, int n_left, const ovo_keypoint* kps_right,
                       const uint8_t* desc_right, int n_right, const float* scale_factors, const float* inv_scale_factors,
                       float focal_x_baseline, float true_baseline, float* stereo_x_right, float* depths);
int ovo_stereo_compute_v(const uint8_t* const* pyr_left, const uint8_t* const* pyr_right, const int32_t* level_rows,
                       const int32_t* level_cols, const size_t* stride_left, const size_t* stride_right, int num_levels,
                       const ovo_keypoint* kps_left, const uint8_t* desc_left, int n_left, const ovo_keypoint* kps_right,
                       const uint8_t* desc_right, int n_right, const float* scale_factors, const float* inv_scale_factors,
                       float focal_x_baseline, float true_baseline, float* stereo_x_right, float* depths, int variant);

/* ---- optimize::pose_optimizer (ovo_pose.cc) ---- */
typedef struct ovo_pose_obs {   /* one observed landmark of the frame: pose_opt_edge_wrapper */
    double pos_w[3];
    double obs_x, obs_y, obs_x_right;   /* undistorted keypoint; obs_x_right only for stereo keypoints */
    double inv_sigma_sq;                /* inv_level_sigma_sq[octave] */
    int32_t is_stereo, pad;
} ovo_pose_obs;
/* pose_cw: 12 doubles (rotation row-major, translation). cam4 = fx, fy, cx, cy. outlier[n] = frm.outlier_flags_. */
int ovo_pose_set_variant(int which, int value);   /* which 0: reset the frame vertex every round (0 | 1), process-wide */
int ovo_pose_optimize(const double* pose_cw_in, const ovo_pose_obs* obs, int n, const double* cam4, double focal_x_baseline, int setup_type,
                      double* pose_cw_out, uint8_t* outlier, int* num_valid);
/* equirectangular frames (equirectangular_pose_opt_edge): monocular edges only, Monocular rig; see ovo_pose.cc */
int ovo_pose_optimize_equirect(const double* pose_cw_in, const ovo_pose_obs* obs, int n, int cols, int rows, double* pose_cw_out,
                               uint8_t* outlier, int* num_valid);

#ifdef __cplusplus
}
#endif
#endif
