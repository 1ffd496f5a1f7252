/*
 * ovslam_hip.h -- C ABI of libovslam_hip.so: the MI355X (gfx950) implementation of OpenVSLAM's per-frame hot path.
 *
 * This is the drop-in boundary.  Upstream has NO plugin / FFI interface for this path: the
 * boundary is the public surface of concrete C++ classes.  Each entry point below names the upstream class method whose
 * body it replaces ("replaces:"), by EXPECTED path inside an OpenVSLAM checkout -- line numbers cannot be given because
 * /root/reference contains only upstream's takedown notice (/root/reference/README.md:1-4).  The C++ classes with the
 * upstream signatures that call this ABI live in openvslam_amd/cpp/ (see INTEGRATION.md).
 *
 * Conventions
 *   - plain C types only; no cv::, Eigen:: or torch types; all functions return ovs_status (0 = OK, <0 = error);
 *   - "host" entry points take caller-owned host pointers and do H2D / kernels / D2H on the handle's own stream;
 *   - "_dev" entry points take DEVICE pointers (inputs already resident in HBM) plus a hipStream_t passed as void*, used
 *     verbatim (NULL = HIP's default stream, which is also what torch's default stream handle is) and never synchronise:
 *     the caller orders and syncs the stream;
 *   - handles are independent (own stream, own buffers, no global mutable state), so two extractors may run
 *     concurrently from two threads (upstream: stereo left/right std::threads); one handle = one caller at a time;
 *   - there is NO CPU fallback inside this library: without a usable HIP device every call fails with
 *     OVS_ERR_NO_DEVICE / OVS_ERR_HIP.
 */
#ifndef OVSLAM_HIP_H
#define OVSLAM_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t ovs_status;
#define OVS_OK 0
#define OVS_ERR_INVALID (-1)     /* bad argument */
#define OVS_ERR_NO_DEVICE (-2)   /* no HIP device visible */
#define OVS_ERR_HIP (-3)         /* a HIP runtime call failed; see ovs_last_error() */
#define OVS_ERR_CAPACITY (-4)    /* image / batch / output larger than the handle or buffer was created for */
#define OVS_ERR_ALIGN (-5)       /* device image base or row stride not 4-byte aligned */

#define OVS_MAX_LEVELS 16

/* Thread-local text of the last HIP error seen by the calling thread ("" if none). */
const char* ovs_last_error(void);
/* Number of visible HIP devices (0 if none / runtime unusable). */
int32_t ovs_device_count(void);
/* gfx arch name of device `dev` into buf (e.g. "gfx950"). */
ovs_status ovs_device_arch(int32_t dev, char* buf, size_t buflen);

/* Layout-compatible with cv::KeyPoint: pt.x, pt.y, size, angle, response, octave, class_id (28 bytes). */
typedef struct ovs_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} ovs_keypoint;

/* replaces: feature::orb_params (src/openvslam/feature/orb_params.h): same five fields, same defaults
 * (2000, 1.2, 8, 20, 7). Feature.mask_rectangles is applied by the C++ class (it rasterises rect_mask_). */
typedef struct ovs_orb_params {
    int32_t max_num_keypts;
    float scale_factor;
    int32_t num_levels;
    int32_t ini_fast_thr;
    int32_t min_fast_thr;
} ovs_orb_params;

/* ------------------------------------------------------------------------------------------------------------------
 * ORB extractor.  replaces: feature::orb_extractor (src/openvslam/feature/orb_extractor.{h,cc}):
 *   ctor/initialize/calc_scale_factors -> ovs_orb_create (+ ovs_orb_tables)
 *   extract(image, mask, keypts, descriptors) -> ovs_orb_extract (host) / ovs_orb_extract_batch_dev (device, batched)
 *   public member image_pyramid_ -> ovs_orb_pyramid_level
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ovs_orb ovs_orb;

/* Creates an extractor able to process up to max_batch frames of up to max_rows x max_cols per call on HIP device
 * `device`. All device memory is allocated here; extract calls never allocate. */
ovs_status ovs_orb_create(const ovs_orb_params* params, int32_t max_rows, int32_t max_cols, int32_t max_batch, int32_t device,
                          ovs_orb** out);
ovs_status ovs_orb_destroy(ovs_orb* h);
/* the HIP device the handle was created on (-1 for NULL): matcher / stereo contexts of the class shims are created on the same one */
int32_t ovs_orb_device(const ovs_orb* h);

/* replaces: orb_extractor::get_scale_factors / get_inv_scale_factors / get_level_sigma_sq / get_inv_level_sigma_sq and
 * num_keypts_per_level_. Any output pointer may be NULL. Arrays hold num_levels entries. */
ovs_status ovs_orb_tables(const ovs_orb* h, float* scale_factors, float* inv_scale_factors, float* level_sigma_sq,
                          float* inv_level_sigma_sq, int32_t* num_keypts_per_level);

/* Test hook for failure handling above the ABI (openvslam_amd/cpp/openvslam/util/device_policy.h, tests/test_cpp_shim.py): after
 * `skip_calls` further HIP runtime calls that the library checks on a run-time path, the next `n_calls` of them still execute but are
 * REPORTED as failed (hipErrorLaunchFailure -> OVS_ERR_HIP), any thread. (0, 0) disarms. Nothing else in the library reads it. */
ovs_status ovs_debug_inject_hip_failures(int32_t skip_calls, int32_t n_calls);

/* Upper bound on the keypoints one frame can produce (sum over levels of N_level + 3; 2 N_level + 3 after
 * ovs_orb_set_variant(OVS_VARIANT_TREE_SWITCH_FACTOR, 1) -- ask again after changing that variant): size outputs with this. */
int32_t ovs_orb_max_keypoints(const ovs_orb* h);

/* replaces: orb_extractor::extract(const cv::_InputArray& image, const cv::_InputArray& mask,
 *                                  std::vector<cv::KeyPoint>& keypts, const cv::_OutputArray& descriptors).
 * image: rows x cols CV_8UC1 with `stride` bytes per row. mask: NULL or rows x cols u8, 0 = masked out (upstream:
 * in_mask or rect_mask_). kps / desc: caller-owned, capacity `cap` keypoints (desc = cap x 32 bytes, row i of desc
 * belongs to kps[i]; byte j bit i = test 8j+i). *n_out = number written. Keypoints are ordered level-major, within a
 * level in quad-tree node-list order. Empty image (rows or cols 0): OK with *n_out = 0 (upstream: early return). */
ovs_status ovs_orb_extract(ovs_orb* h, const uint8_t* image, int32_t rows, int32_t cols, size_t stride, const uint8_t* mask,
                           size_t mask_stride, ovs_keypoint* kps, uint8_t* desc, int32_t cap, int32_t* n_out);

/* Stereo rig in ONE call: upstream's frame constructor runs  extractor_left_->extract(...)  and  extractor_right_->extract(...)  on two
 * std::threads (src/openvslam/data/frame.cc, stereo constructor); two extractors on two threads cost 0.36 ms each here against 0.23 ms
 * alone (the ~25 runtime calls of a frame queue on the runtime's locks), while a batch of two costs no more launches than one frame.
 * Results are bit-identical to two ovs_orb_extract calls. The handle must have been created with max_batch >= 2 (OVS_ERR_CAPACITY
 * otherwise); masks: both or neither; cap applies to each side. */
ovs_status ovs_orb_extract_pair(ovs_orb* h, const uint8_t* left, const uint8_t* right, int32_t rows, int32_t cols, size_t stride,
                                const uint8_t* mask_left, const uint8_t* mask_right, size_t mask_stride, ovs_keypoint* kps_left,
                                uint8_t* desc_left, int32_t* n_left, ovs_keypoint* kps_right, uint8_t* desc_right, int32_t* n_right, int32_t cap);

/* Asynchronous pair behind ovs_orb_extract (which is submit + collect): the handle owns TWO slots of pinned staging memory, uploads on
 * its own copy stream and runs kernels + the single result D2H on its compute stream, so the upload of frame k+1 overlaps the kernels
 * of frame k ("copies overlapped on a second stream, double-buffered pinned host memory"). At most two frames may be in
 * flight; collect returns them in submission order and is the only point that waits (one event per frame). The image / mask buffers may
 * be reused as soon as submit returns. */
ovs_status ovs_orb_extract_submit(ovs_orb* h, const uint8_t* image, int32_t rows, int32_t cols, size_t stride, const uint8_t* mask,
                                  size_t mask_stride);
ovs_status ovs_orb_extract_collect(ovs_orb* h, ovs_keypoint* kps, uint8_t* desc, int32_t cap, int32_t* n_out);
/* Upload strategy of the host path: 0 (default) = hipMemcpy2DAsync straight from the caller's pageable rows (0.054 ms per 1080p frame on
 * the MI355X box); 1 = row bands copied through the slot's pinned buffer, each band's DMA overlapping the CPU copy of the next (0.12 ms:
 * kept for hosts whose runtime stages pageable copies badly). Same results; both are measured by openvslam_amd/cpp/bench_shim. */
ovs_status ovs_orb_set_host_mode(ovs_orb* h, int32_t mode);
/* h2d | kernels | d2h milliseconds of the last collected frame (HIP events; valid after ovs_orb_profile_enable(h, 1)). */
ovs_status ovs_orb_host_profile_read(const ovs_orb* h, float* h2d_kernels_d2h_ms);

/* replaces: the public member orb_extractor::image_pyramid_ on the host path WITHOUT a copy per level: with enable != 0 every submitted
 * frame also brings its pyramid block (levels 1 .. L-1, one D2H) into pinned memory owned by the handle; ovs_orb_host_pyramid_level
 * returns a pointer into that block for the last COLLECTED frame (valid until the slot is reused two submits later), with its pitch.
 * Level 0 reports base = NULL: it is the caller's own image (upstream aliases it as well). Default: disabled -- a monocular tracker
 * never reads image_pyramid_, and match::stereo reads the device copy (ovs_stereo_compute). */
ovs_status ovs_orb_set_host_pyramid(ovs_orb* h, int32_t enable);
ovs_status ovs_orb_host_pyramid_level(const ovs_orb* h, int32_t level, const uint8_t** base, int32_t* rows, int32_t* cols, int32_t* pitch);

/* Device-resident batched form of the same computation: `batch` frames of rows x cols u8 already in HBM, frame b at
 * d_images + b*frame_stride, row stride `stride` (both multiples of 4, base 4-byte aligned). d_masks: NULL or same layout.
 * Outputs stay in HBM: d_kps[b*cap + i], d_desc[(b*cap + i)*32], d_counts[b] (counts are clamped to cap).
 * Asynchronous on `stream`. */
ovs_status ovs_orb_extract_batch_dev(ovs_orb* h, const uint8_t* d_images, int32_t batch, int32_t rows, int32_t cols, size_t stride,
                                     size_t frame_stride, const uint8_t* d_masks, ovs_keypoint* d_kps, uint8_t* d_desc,
                                     int32_t* d_counts, int32_t cap, void* stream);

/* Sub-batch pipelining of ovs_orb_extract_batch_dev (no upstream counterpart: upstream extracts one frame per call). With
 * n_sub > 1 the batch is cut into n_sub contiguous sub-batches whose pyramid -> FAST -> quad-tree -> describe chains are issued on
 * n_sub internal streams, forked from / joined to the caller's stream with events: the latency-bound stages of one sub-batch
 * overlap the VALU-bound FAST pass of another. Results are identical (frames are independent). n_sub in [1, 4]; 1 = off. */
ovs_status ovs_orb_set_pipeline(ovs_orb* h, int32_t n_sub);

/* Level-0 FAST beside the pyramid (no upstream counterpart; default ON): FAST on level 0 reads the caller's image and does not depend on
 * the seven resize launches, so it is issued on an internal stream next to them (integer-VALU-bound beside latency / bandwidth-bound) and the
 * remaining levels follow the pyramid. Identical results (cells are independent). With it ON the four stage times of ovs_orb_profile_read are
 * {pyramid window (level-0 FAST running beside it), FAST levels >= 1, quad-tree, describe}; ovs_orb_profile_read_aux returns the level-0 FAST
 * launch's own duration. OFF = one FAST launch over all levels after the pyramid (what per-kernel rooflines are quoted on). */
ovs_status ovs_orb_set_fast_split(ovs_orb* h, int32_t enable);

/* One-launch pyramid for single frames (no upstream counterpart; default max_frames = 2): when a call extracts at most max_frames frames (the
 * tracker's orb_extractor::extract, expected src/openvslam/feature/orb_extractor.cc compute_image_pyramid), all levels are computed by ONE kernel
 * launch (each workgroup chains its tile through every level in LDS) instead of num_levels - 1 dependent launches: lower latency, lower throughput
 * (measured: one frame 24 vs 37 us, two frames 33.5 vs 35.7, eight 73 vs 46). Identical bytes in every plane. 0 = level-by-level launches always. */
ovs_status ovs_orb_set_pyramid_chain(ovs_orb* h, int32_t max_frames);
/* Variants of the three rules of the extraction that upstream's (absent) sources and OpenCV version decide and that cannot be pinned in this
 * repository (oracle/ORACLE_SPEC.md rules 6, 7, 10): if a checkout or an OpenCV build shows the other choice, matching it
 * is this call, not a kernel rewrite. Defaults unchanged; the CPU checker has the same switches and every setting is parity-tested.
 *   OVS_VARIANT_TREE_SWITCH_FACTOR  3 (default: ORB-SLAM2's `N < nodes + 3 * splittable`) | 1
 *   OVS_VARIANT_TREE_TIE_ORDER      0 (default: equal counts -> later-created node first) | 1 (earlier-created first)
 *   OVS_VARIANT_BLUR_TAPS           0 (default: 18 34 48 56 48 34 18, OpenCV >= 3.4.7 error-diffused fixed point) | 1 (18 34 49 55 49 34 18,
 *                                   every tap rounded on its own: older OpenCV; sum 257, result saturates)
 *   OVS_VARIANT_TRIG                0 (default: util::cos / util::sin, the polynomial of src/openvslam/util/trigonometric.h) | 1 (std::cos /
 *                                   std::sin on the float angle as glibc >= 2.28 computes them: ORB-SLAM2's form; rule 11, round 4)
 * Takes effect from the next extract. OVS_ERR_INVALID for an unknown (which, value). */
#define OVS_VARIANT_TREE_SWITCH_FACTOR 0
#define OVS_VARIANT_TREE_TIE_ORDER 1
#define OVS_VARIANT_BLUR_TAPS 2
#define OVS_VARIANT_TRIG 3
ovs_status ovs_orb_set_variant(ovs_orb* h, int32_t which, int32_t value);
ovs_status ovs_orb_profile_read_aux(ovs_orb* h, float* fast_level0_ms, int32_t* ncalls);

/* replaces: the public member orb_extractor::image_pyramid_ (read by match::stereo). Copies level `level` of frame
 * `frame` (0 for the host API) of the LAST extract into host_dst (rows*cols bytes, contiguous) and reports its size.
 * host_dst may be NULL to query the size only. Waits for the stream the last extract ran on (not for the device). */
ovs_status ovs_orb_pyramid_level(ovs_orb* h, int32_t frame, int32_t level, uint8_t* host_dst, int32_t* rows, int32_t* cols);

/* Test / profiling observables of the LAST extract (synchronise the handle's stream):
 * FAST candidates of (frame, level) after the per-cell two-threshold rule, as level-image x, y and FAST score, sorted by
 * upstream's emission order (cell row, cell column, then row-major inside the cell). Returns the count in *n_out. */
ovs_status ovs_orb_debug_candidates(ovs_orb* h, int32_t frame, int32_t level, int32_t* xs, int32_t* ys, int32_t* scores,
                                    int32_t cap, int32_t* n_out);
/* Keypoints per level of (frame) of the last extract. */
ovs_status ovs_orb_debug_level_counts(ovs_orb* h, int32_t frame, int32_t* counts /* num_levels */);

/* Measurement hooks (bench.py): when enabled, every extract call records HIP events on the stream it launches on, at the
 * stage boundaries pyramid | FAST | quad-tree | describe. ovs_orb_profile_read waits for the recorded events, writes the
 * accumulated milliseconds per stage (4 floats) and the number of calls they cover, and resets the accumulators. */
ovs_status ovs_orb_profile_enable(ovs_orb* h, int32_t enable);
ovs_status ovs_orb_profile_read(ovs_orb* h, float* stage_ms /* 4 */, int32_t* ncalls);

/* ------------------------------------------------------------------------------------------------------------------
 * 256-bit Hamming matching.  replaces: match::base::compute_descriptor_distance_32 and match::robust::brute_force_match
 * (src/openvslam/match/base.h, robust.{h,cc}).
 * ------------------------------------------------------------------------------------------------------------------ */
#define OVS_HAMMING_DIST_THR_LOW 50
#define OVS_HAMMING_DIST_THR_HIGH 100
#define OVS_MAX_HAMMING_DIST 256

typedef struct ovs_matcher ovs_matcher;
/* A matcher context sized for `max_batch` problems of up to max_n1 frame descriptors x max_n2 keyframe descriptors. */
ovs_status ovs_matcher_create(int32_t max_n1, int32_t max_n2, int32_t max_batch, int32_t device, ovs_matcher** out);
ovs_status ovs_matcher_destroy(ovs_matcher* m);

/* replaces: unsigned int robust::brute_force_match(data::frame& frm, data::keyframe* keyfrm,
 *                                                  std::vector<std::pair<int,int>>& matches) const.
 * desc_1: n1 x 32 frame descriptors (idx_1 side); desc_2: n2 x 32 keyframe descriptors (idx_2 side); valid_2: NULL or n2
 * bytes, non-zero where the keyframe keypoint holds a landmark that !will_be_erased(). valid_1: NULL or n1 bytes, zero where the
 * FRAME keypoint must not take part (treated exactly like an already matched idx_1: skipped in the inner loop); the shim passes NULL
 * because upstream's inner loop, as recalled, only tests already_matched_indices_1 -- a maintainer whose checkout also skips frame
 * keypoints that hold a landmark (`if (frm.landmarks_.at(idx_1)) continue;`) fills it from frm.landmarks_. pairs: capacity cap pairs of
 * (idx_1, idx_2) in upstream's emission order (ascending idx_2). *n_out = number of matches. */
ovs_status ovs_robust_brute_force_match(ovs_matcher* m, const uint8_t* desc_1, int32_t n1, const uint8_t* valid_1, const uint8_t* desc_2,
                                        int32_t n2, const uint8_t* valid_2, float lowe_ratio, int32_t* pairs, int32_t cap, int32_t* n_out);

/* Device-resident batched form: problem p uses d_desc_1 + p*stride_1 (n1[p] rows) and d_desc_2 + p*stride_2 (n2[p] rows);
 * d_n1 / d_n2 are device int32[batch] (so extractor counts can be consumed without a host round trip); d_valid_2: NULL or
 * batch x stride_2/32 bytes; d_valid_1: NULL or batch x stride_1/32 bytes. Outputs: d_pairs[(p*cap + i)*2 + {0,1}], d_counts[p]. Asynchronous on `stream`. */
ovs_status ovs_robust_brute_force_match_batch_dev(ovs_matcher* m, const uint8_t* d_desc_1, size_t stride_1, const int32_t* d_n1,
                                                  const uint8_t* d_valid_1, const uint8_t* d_desc_2, size_t stride_2, const int32_t* d_n2,
                                                  const uint8_t* d_valid_2, int32_t batch, float lowe_ratio, int32_t* d_pairs,
                                                  int32_t* d_counts, int32_t cap, void* stream);

/* The all-pairs stage of brute_force_match has two bit-identical implementations: OVS_NEAR_PATH_MATRIX (default) evaluates the
 * 256-bit Hamming distances as exact i8 dot products on the matrix cores, OVS_NEAR_PATH_POPCOUNT is the vector-ALU xor / popcount
 * form (BASELINE north star). Same near lists, same pairs; the switch exists for measurement and for parity tests of both. */
#define OVS_NEAR_PATH_MATRIX 0
#define OVS_NEAR_PATH_POPCOUNT 1
ovs_status ovs_matcher_set_near_path(ovs_matcher* m, int32_t path);

/* Measurement hooks: stages = all-pairs near-list kernel | resolve kernel (2 floats). */
ovs_status ovs_matcher_profile_enable(ovs_matcher* m, int32_t enable);
ovs_status ovs_matcher_profile_read(ovs_matcher* m, float* stage_ms /* 2 */, int32_t* ncalls);

/* Unconstrained best / second-best Hamming distance per query over all valid targets (first-seen = lowest index wins
 * ties), the primitive under bow_tree / area / projection style matchers when the candidate set is "all". Host pointers. */
ovs_status ovs_hamming_best2(ovs_matcher* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, const uint8_t* t_valid,
                             int32_t* best_idx, uint16_t* best, uint16_t* second);

/* ------------------------------------------------------------------------------------------------------------------
 * Grid candidate generator and the windowed matchers built on it.
 * replaces: data::assign_keypoints_to_grid / get_keypoints_in_cell (src/openvslam/data/common.{h,cc}),
 *           match::projection::match_frame_and_landmarks (src/openvslam/match/projection.{h,cc}),
 *           match::area::match_in_consistent_area (src/openvslam/match/area.{h,cc}),
 *           match::bow_tree::match_frame_and_keyframe (src/openvslam/match/bow_tree.{h,cc}),
 *           match::angle_checker (src/openvslam/match/angle_checker.h).
 * Keypoints cross the boundary as ovs_keypoint arrays (= std::vector<cv::KeyPoint>::data(), e.g. frm.undist_keypts_).
 * ------------------------------------------------------------------------------------------------------------------ */
/* camera::base::img_bounds_ and num_grid_cols_ / num_grid_rows_ (64 x 48 upstream). */
typedef struct ovs_grid_params {
    float min_x, min_y, max_x, max_y;
    int32_t cols, rows;
} ovs_grid_params;

typedef struct ovs_wmatcher ovs_wmatcher;
/* A context for problems of up to max_targets grid-side keypoints, max_queries queries and max_entries candidate pairs
 * (a call that would produce more returns OVS_ERR_CAPACITY; nothing is truncated silently). */
ovs_status ovs_wmatcher_create(int32_t max_targets, int32_t max_queries, int32_t max_entries, int32_t device, ovs_wmatcher** out);
ovs_status ovs_wmatcher_destroy(ovs_wmatcher* w);
/* oracle/ORACLE_SPEC.md rule 17 (match::angle_checker), tagged M / L for the keep rule, as a run-time variant (process-wide; oracle:
 * ovo_match_set_variant): ANGLE_KEEP_RULE 0 (default: the three fullest of the 30 bins are kept whatever they hold) | 1 (ORB-SLAM2's
 * ComputeThreeMaxima: a second bin with fewer than 0.1 x the fullest bin's entries is dropped together with the third, a third bin below that
 * alone). Applies to every matcher with check_orientation (the device resolvers and, through ovs_match_get_variant, the host-side
 * match::angle_checker of the class shims). */
#define OVS_MATCH_VARIANT_ANGLE_KEEP_RULE 0
/* rule 17's tie order: ANGLE_TIE_ORDER 0 (default: of two equally full bins the LOWER one ranks first) | 1 (the higher one). Upstream ranks the
 * bins with std::sort on their sizes (expected src/openvslam/match/angle_checker.h), which leaves equal sizes to the library. Same scope as above. */
#define OVS_MATCH_VARIANT_ANGLE_TIE_ORDER 1
/* rule 14's frame-side test in robust::brute_force_match (expected src/openvslam/match/robust.cc): BF_FRAME_MASK 0 (default: a frame keypoint is
 * skipped only when this call has already matched it) | 1 (also when it already owns a landmark: `if (frm.landmarks_.at(idx_1)) continue;`, as one
 * recollection of upstream has it). The ABI takes the mask either way (valid_1 of ovs_robust_brute_force_match); this switch is what the class shim
 * robust::brute_force_match passes. */
#define OVS_MATCH_VARIANT_BF_FRAME_MASK 2
ovs_status ovs_match_set_variant(int32_t which, int32_t value);
int32_t ovs_match_get_variant(int32_t which);   /* -1 for an unknown variant */

/* replaces: data::assign_keypoints_to_grid(camera, undist_keypts, keypt_indices_in_cells).
 * CSR result: cell id = cx*rows + cy (upstream keypt_indices_in_cells[cx][cy]), cell_start[cols*rows + 1], items = keypoint
 * indices, ascending inside a cell (upstream's push_back order). items may be NULL. */
ovs_status ovs_assign_keypoints_to_grid(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* kps, int32_t n,
                                        int32_t* cell_start, int32_t* items, int32_t* n_items);
/* Device form: builds the context's grid from keypoints resident in HBM (used by the matchers below). */
ovs_status ovs_grid_assign_dev(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* d_kps, int32_t n, void* stream);

/* replaces: unsigned int projection::match_frame_and_landmarks(data::frame& frm, const std::vector<data::landmark*>&
 *                                                              local_landmarks, const float margin) const.
 * Frame side: kps = frm.undist_keypts_, desc = frm.descriptors_ (n x 32), stereo_x_right = frm.stereo_x_right_ or NULL,
 * occupied[i] != 0 iff frm.landmarks_[i] && frm.landmarks_[i]->has_observation() (NULL = none).
 * Landmark side (flattened by the shim, in the order of local_landmarks): lm_xy = reproj_in_tracking_, lm_x_right =
 * x_right_in_tracking_ (required iff stereo_x_right), lm_level = scale_level_in_tracking_, lm_desc = get_descriptor(),
 * lm_valid[l] != 0 iff is_observable_in_tracking_ && !will_be_erased() (NULL = all).
 * scale_factors = frm.scale_factors_. Output: assigned[l] = index of the frame keypoint landmark l is written to
 * (frm.landmarks_[assigned[l]] = local_landmarks[l]) or -1; *num_matches = the return value. */
ovs_status ovs_projection_match_frame_and_landmarks(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* kps,
                                                    const uint8_t* desc, const float* stereo_x_right, const uint8_t* occupied, int32_t n,
                                                    const float* lm_xy, const float* lm_x_right, const int32_t* lm_level,
                                                    const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m,
                                                    const float* scale_factors, int32_t num_levels, float margin, float lowe_ratio,
                                                    int32_t* assigned, int32_t* num_matches);
/* Device-pointer form (scale_factors stays a host pointer); asynchronous on `stream`. */
ovs_status ovs_projection_match_frame_and_landmarks_dev(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* d_kps,
                                                        const uint8_t* d_desc, const float* d_stereo_x_right, const uint8_t* d_occupied,
                                                        int32_t n, const float* d_lm_xy, const float* d_lm_x_right,
                                                        const int32_t* d_lm_level, const uint8_t* d_lm_desc, const uint8_t* d_lm_valid,
                                                        int32_t m, const float* scale_factors, int32_t num_levels, float margin,
                                                        float lowe_ratio, int32_t* d_assigned, int32_t* d_num_matches, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Frame residency.  A frame's matcher-side data resident in HBM: what data::frame's constructor produces once per
 * frame -- undist_keypts_, descriptors_, stereo_x_right_ (NULL = monocular) and the keypoint grid of data::assign_keypoints_to_grid
 * (keypt_indices_in_cells_) -- uploaded with one copy and indexed once. tracking_module calls two to four matchers on the same frame;
 * the *_f entry points below take the handle instead of re-uploading and re-indexing the frame in every call, stage their per-call host
 * arrays through one pinned buffer (one copy up) and fetch the result block with one copy down. Device arenas of destroyed handles are
 * pooled: creating one handle per tracked frame costs no hipMalloc in steady state. The handle is read-only after creation and may be
 * shared by matcher contexts / threads on the same device.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ovs_frame_dev ovs_frame_dev;
ovs_status ovs_frame_dev_create(int32_t device, const ovs_grid_params* gp, const ovs_keypoint* undist_kps, const uint8_t* desc,
                                const float* stereo_x_right, int32_t n, ovs_frame_dev** out);
ovs_status ovs_frame_dev_destroy(ovs_frame_dev* f);
int32_t ovs_frame_dev_num_keypoints(const ovs_frame_dev* f);
/* projection::match_frame_and_landmarks with the frame side resident (arguments as ovs_projection_match_frame_and_landmarks). */
ovs_status ovs_projection_match_frame_and_landmarks_f(ovs_wmatcher* w, const ovs_frame_dev* frm, const uint8_t* occupied, const float* lm_xy,
                                                      const float* lm_x_right, const int32_t* lm_level, const uint8_t* lm_desc,
                                                      const uint8_t* lm_valid, int32_t m, const float* scale_factors, int32_t num_levels,
                                                      float margin, float lowe_ratio, int32_t* assigned, int32_t* num_matches);
/* area::match_in_consistent_area with both frames resident (initializer: frm_1 is matched against every new frame until it succeeds). */
ovs_status ovs_area_match_in_consistent_area_f(ovs_wmatcher* w, const ovs_frame_dev* frm_1, const ovs_frame_dev* frm_2, float* prev_matched_xy,
                                               int32_t* matched_2_in_1, int32_t margin, float lowe_ratio, int32_t check_orientation,
                                               int32_t* num_matches);

/* camera::base subset needed by the matchers that reproject inside the call. model: 0 = perspective, 1 = equirectangular
 * (camera::model_type_t); setup: 0 = Monocular, 1 = Stereo, 2 = RGBD (camera::setup_type_t). */
typedef struct ovs_camera {
    int32_t model, setup;
    double fx, fy, cx, cy;
    double focal_x_baseline, true_baseline;
    int32_t cols, rows;
} ovs_camera;

/* replaces: unsigned int projection::match_current_and_last_frames(data::frame& curr_frm, const data::frame& last_frm,
 *                                                                  const float margin) const.
 * pose_cw_* = frm.cam_pose_cw_ as 12 doubles (rotation row-major, then translation). curr_*: undist_keypts_, descriptors_,
 * stereo_x_right_ (or NULL), occupied as in match_frame_and_landmarks. last_kps = last_frm.undist_keypts_; last_pos_w[i] =
 * landmarks_[i]->get_pos_in_world(), last_lm_desc[i] = landmarks_[i]->get_descriptor(), last_valid[i] != 0 iff landmarks_[i] &&
 * !outlier_flags_[i] (rows of invalid entries are ignored). assigned[i] = current keypoint that receives last_frm.landmarks_[i]
 * (curr_frm.landmarks_[assigned[i]] = lm) or -1. camera::reproject_to_image runs on the device in double precision
 * (perspective: exact parity with the CPU oracle; equirectangular: asin/atan2 are the device library's). */
ovs_status ovs_projection_match_current_and_last_frames(ovs_wmatcher* w, const ovs_camera* cam, const ovs_grid_params* gp,
                                                        const ovs_keypoint* curr_kps, const uint8_t* curr_desc,
                                                        const float* curr_stereo_x_right, const uint8_t* curr_occupied, int32_t n_curr,
                                                        const double* pose_cw_curr, const ovs_keypoint* last_kps, const double* last_pos_w,
                                                        const uint8_t* last_lm_desc, const uint8_t* last_valid, int32_t n_last,
                                                        const double* pose_cw_last, const float* scale_factors, int32_t num_levels,
                                                        float margin, int32_t check_orientation, int32_t* assigned, int32_t* num_matches);
/* The same with the current frame resident (ovs_frame_dev; its grid parameters are the handle's). */
ovs_status ovs_projection_match_current_and_last_frames_f(ovs_wmatcher* w, const ovs_camera* cam, const ovs_frame_dev* curr,
                                                          const uint8_t* curr_occupied, const double* pose_cw_curr, const ovs_keypoint* last_kps,
                                                          const double* last_pos_w, const uint8_t* last_lm_desc, const uint8_t* last_valid,
                                                          int32_t n_last, const double* pose_cw_last, const float* scale_factors,
                                                          int32_t num_levels, float margin, int32_t check_orientation, int32_t* assigned,
                                                          int32_t* num_matches);

/* replaces: unsigned int projection::match_frame_and_keyframe(data::frame& curr_frm, data::keyframe* keyfrm,
 *               const std::set<data::landmark*>& already_matched_lms, const float margin, const unsigned int hamm_dist_thr) const.
 * curr_occupied[j] != 0 iff curr_frm.landmarks_[j]; kf_kps = keyfrm->undist_keypts_; per keyframe keypoint i: kf_pos_w, kf_dist_min_max
 * (the RAW min_valid_dist_ / max_valid_dist_ members, see ovs_fuse_replace_duplication), kf_lm_desc of landmarks[i]; kf_valid[i] != 0 iff landmarks[i] && !will_be_erased() &&
 * !already_matched_lms.count(landmarks[i]). assigned[i] = current keypoint that receives landmarks[i], or -1. */
ovs_status ovs_projection_match_frame_and_keyframe(ovs_wmatcher* w, const ovs_camera* cam, const ovs_grid_params* gp,
                                                   const ovs_keypoint* curr_kps, const uint8_t* curr_desc, const uint8_t* curr_occupied,
                                                   int32_t n_curr, const double* pose_cw_curr, const ovs_keypoint* kf_kps,
                                                   const double* kf_pos_w, const float* kf_dist_min_max, const uint8_t* kf_lm_desc,
                                                   const uint8_t* kf_valid, int32_t n_kf, const float* scale_factors, int32_t num_levels,
                                                   float log_scale_factor, float margin, uint32_t hamm_dist_thr, int32_t check_orientation,
                                                   int32_t* assigned, int32_t* num_matches);

/* replaces: the candidate search of  template<typename T> unsigned int fuse::replace_duplication(data::keyframe* keyfrm,
 *               const T& landmarks_to_check, const float margin)  (src/openvslam/match/fuse.{h,cc}); landmarks are independent there.
 * kps / desc / stereo_x_right = keyfrm->undist_keypts_ / descriptors_ / stereo_x_right_ (NULL = monocular); pose_cw = keyfrm pose.
 * Per landmark (in the order of landmarks_to_check): lm_pos_w = get_pos_in_world(), lm_dist_min_max = the RAW (min_valid_dist_,
 * max_valid_dist_) members (the kernel applies the getters' 0.7 / 1.3 widening to the range gate and, like landmark::predict_scale_level,
 * the raw maximum to the level prediction), lm_normal = get_obs_mean_normal(), lm_desc = get_descriptor(), lm_valid != 0 iff lm &&
 * !will_be_erased() && !is_observed_in_keyframe(keyfrm). inv_level_sigma_sq / log_scale_factor = the keyframe's tables.
 * best_idx[l] = keypoint the landmark fuses with or -1; the shim then performs upstream's replace / add_observation in order. */
ovs_status ovs_fuse_replace_duplication(ovs_wmatcher* w, const ovs_camera* cam, const ovs_grid_params* gp, const ovs_keypoint* kps,
                                        const uint8_t* desc, const float* stereo_x_right, int32_t n, const double* pose_cw,
                                        const double* lm_pos_w, const float* lm_dist_min_max, const double* lm_normal,
                                        const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m, const float* scale_factors,
                                        const float* inv_level_sigma_sq, int32_t num_levels, float log_scale_factor, float margin,
                                        int32_t* best_idx, int32_t* num_fused);

/* replaces: the candidate search of  template<typename T> unsigned int fuse::detect_duplication(data::keyframe* keyfrm,
 *               const Mat44_t& Sim3_cw, const T& landmarks_to_check, const float margin, std::vector<data::landmark*>&
 *               duplicated_lms_in_keyfrm)  (src/openvslam/match/fuse.{h,cc}; loop closing). sim3_cw = the top 3 rows of Sim3_cw as 12 doubles
 * (sR row-major, then the translation column); it is decomposed here as upstream does (scale from the first row). No chi-square gate.
 * lm_valid != 0 iff lm && !will_be_erased() && lm is not already a landmark of keyfrm. best_idx as for ovs_fuse_replace_duplication. */
ovs_status ovs_fuse_detect_duplication(ovs_wmatcher* w, const ovs_camera* cam, const ovs_grid_params* gp, const ovs_keypoint* kps,
                                       const uint8_t* desc, int32_t n, const double* sim3_cw, const double* lm_pos_w,
                                       const float* lm_dist_min_max, const double* lm_normal, const uint8_t* lm_desc, const uint8_t* lm_valid,
                                       int32_t m, const float* scale_factors, int32_t num_levels, float log_scale_factor, float margin,
                                       int32_t* best_idx, int32_t* num_found);

/* replaces: unsigned int projection::match_by_Sim3_transform(data::keyframe* keyfrm, const Mat44_t& Sim3_cw,
 *               const std::vector<data::landmark*>& landmarks, std::vector<data::landmark*>& matched_lms_in_keyfrm, float margin)
 * (src/openvslam/match/projection.{h,cc}; loop closing). occupied[k] != 0 iff matched_lms_in_keyfrm[k] != nullptr on entry; lm_valid != 0
 * iff the landmark is live and not already in matched_lms_in_keyfrm. Landmarks claim keypoints in order (sequential claim, replayed
 * exactly). assigned[l] = keypoint that receives landmark l, or -1. */
ovs_status ovs_projection_match_by_sim3_transform(ovs_wmatcher* w, const ovs_camera* cam, const ovs_grid_params* gp, const ovs_keypoint* kps,
                                                  const uint8_t* desc, const uint8_t* occupied, int32_t n, const double* sim3_cw,
                                                  const double* lm_pos_w, const float* lm_dist_min_max, const double* lm_normal,
                                                  const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m, const float* scale_factors,
                                                  int32_t num_levels, float log_scale_factor, float margin, int32_t* assigned,
                                                  int32_t* num_matches);

/* replaces: unsigned int projection::match_keyframes_mutually(data::keyframe* keyfrm_1, data::keyframe* keyfrm_2,
 *               std::vector<data::landmark*>& matched_lms_in_keyfrm_1, float s_12, const Mat33_t& rot_12, const Vec3_t& trans_12,
 *               float margin)  (src/openvslam/match/projection.{h,cc}; Sim3 refinement in loop detection). Per keyframe k: keypoints,
 * descriptors, pose_cw (12 doubles) and, indexed by keypoint, the landmark's world position / (min, max) valid distance / descriptor;
 * lm_valid_k[i] != 0 iff keypoint i has a live landmark that takes part (side 1: not already in matched_lms_in_keyfrm_1; side 2: its
 * keypoint is not the partner of an already matched landmark). Both keyframes share the ORB scale tables.
 * matched_2_in_1[idx_1] = idx_2 for the pairs both directions agree on, else -1. */
ovs_status ovs_projection_match_keyframes_mutually(ovs_wmatcher* w, const ovs_camera* cam_1, const ovs_grid_params* gp_1,
                                                   const ovs_keypoint* kps_1, const uint8_t* desc_1, int32_t n1, const double* pose_cw_1,
                                                   const double* lm_pos_w_1, const float* lm_dist_1, const uint8_t* lm_desc_1,
                                                   const uint8_t* lm_valid_1, const ovs_camera* cam_2, const ovs_grid_params* gp_2,
                                                   const ovs_keypoint* kps_2, const uint8_t* desc_2, int32_t n2, const double* pose_cw_2,
                                                   const double* lm_pos_w_2, const float* lm_dist_2, const uint8_t* lm_desc_2,
                                                   const uint8_t* lm_valid_2, double s_12, const double* rot_12, const double* trans_12,
                                                   const float* scale_factors, int32_t num_levels, float log_scale_factor, float margin,
                                                   int32_t* matched_2_in_1, int32_t* num_matches);

/* replaces: unsigned int area::match_in_consistent_area(data::frame& frm_1, data::frame& frm_2,
 *               std::vector<cv::Point2f>& prev_matched_pts, std::vector<int>& matched_indices_2_in_frm_1, int margin).
 * kps_i / desc_i = frm_i.undist_keypts_ / descriptors_; gp = frm_2's camera grid. prev_matched_xy (n1 x 2) is updated in
 * place for the final matches; matched_2_in_1[n1]; lowe_ratio / check_orientation = the matcher's ctor arguments. */
ovs_status ovs_area_match_in_consistent_area(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* kps_1, const uint8_t* desc_1,
                                             int32_t n1, const ovs_keypoint* kps_2, const uint8_t* desc_2, int32_t n2,
                                             float* prev_matched_xy, int32_t* matched_2_in_1, int32_t margin, float lowe_ratio,
                                             int32_t check_orientation, int32_t* num_matches);
ovs_status ovs_area_match_in_consistent_area_dev(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* d_kps_1,
                                                 const uint8_t* d_desc_1, int32_t n1, const ovs_keypoint* d_kps_2,
                                                 const uint8_t* d_desc_2, int32_t n2, float* d_prev_matched_xy,
                                                 int32_t* d_matched_2_in_1, int32_t margin, float lowe_ratio, int32_t check_orientation,
                                                 int32_t* d_num_matches, void* stream);

/* replaces: unsigned int bow_tree::match_frame_and_keyframe(data::keyframe* keyfrm, data::frame& frm,
 *                                                           std::vector<data::landmark*>& matched_lms_in_frm) const.
 * The two BoW feature vectors (std::map<node id, std::vector<unsigned>>) are flattened to CSR over ascending node ids.
 * kf_valid[i] != 0 iff the keyframe keypoint holds a landmark that !will_be_erased() (NULL = all).
 * matched_kf_in_frm[j] = keyframe keypoint index whose landmark frame keypoint j receives, or -1. */
ovs_status ovs_bow_match_frame_and_keyframe(ovs_wmatcher* w, const ovs_keypoint* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_valid,
                                            int32_t n_kf, const int32_t* kf_node_ids, const int32_t* kf_node_start,
                                            const int32_t* kf_items, int32_t kf_nodes, const ovs_keypoint* frm_kps,
                                            const uint8_t* frm_desc, int32_t n_frm, const int32_t* frm_node_ids,
                                            const int32_t* frm_node_start, const int32_t* frm_items, int32_t frm_nodes, float lowe_ratio,
                                            int32_t check_orientation, int32_t* matched_kf_in_frm, int32_t* num_matches);

/* replaces: unsigned int bow_tree::match_keyframes(data::keyframe* keyfrm_1, data::keyframe* keyfrm_2,
 *                                                  std::vector<data::landmark*>& matched_lms_in_keyfrm_1) const.
 * valid_i[k] != 0 iff keyframe i's keypoint k holds a landmark that !will_be_erased(). matched_2_in_1[idx_1] = idx_2 (the shim
 * writes keyfrm_2's landmark there) or -1. */
ovs_status ovs_bow_match_keyframes(ovs_wmatcher* w, const ovs_keypoint* kps_1, const uint8_t* desc_1, const uint8_t* valid_1, int32_t n1,
                                   const int32_t* node_ids_1, const int32_t* node_start_1, const int32_t* items_1, int32_t nodes_1,
                                   const ovs_keypoint* kps_2, const uint8_t* desc_2, const uint8_t* valid_2, int32_t n2,
                                   const int32_t* node_ids_2, const int32_t* node_start_2, const int32_t* items_2, int32_t nodes_2,
                                   float lowe_ratio, int32_t check_orientation, int32_t* matched_2_in_1, int32_t* num_matches);

/* replaces: unsigned int robust::match_for_triangulation(data::keyframe* keyfrm_1, data::keyframe* keyfrm_2, const Mat33_t& E_12,
 *               std::vector<std::pair<unsigned int, unsigned int>>& matched_idx_pairs)  and  robust::check_epipolar_constraint
 * (src/openvslam/match/robust.{h,cc}). kps_i = undist_keypts_, has_lm_i[k] != 0 iff the keypoint already holds a landmark,
 * x_right_i = stereo_x_right_ (NULL = monocular), bearings_i = bearings_ (n x 3 doubles), E_12 row-major, epipole_in_2 = keyfrm_1's
 * camera centre as a bearing in keyfrm_2 (camera->reproject_to_bearing). matched_2_in_1[idx_1] = idx_2 or -1: the shim emits the
 * pairs (idx_1, matched_2_in_1[idx_1]) in ascending idx_1 as upstream does. */
ovs_status ovs_robust_match_for_triangulation(ovs_wmatcher* w, const ovs_keypoint* kps_1, const uint8_t* desc_1, const uint8_t* has_lm_1,
                                              const float* x_right_1, const double* bearings_1, int32_t n1, const int32_t* node_ids_1,
                                              const int32_t* node_start_1, const int32_t* items_1, int32_t nodes_1,
                                              const ovs_keypoint* kps_2, const uint8_t* desc_2, const uint8_t* has_lm_2,
                                              const float* x_right_2, const double* bearings_2, int32_t n2, const int32_t* node_ids_2,
                                              const int32_t* node_start_2, const int32_t* items_2, int32_t nodes_2, const double* E_12,
                                              const double* epipole_in_2, const float* scale_factors, int32_t num_levels,
                                              int32_t check_orientation, int32_t* matched_2_in_1, int32_t* num_matches);

/* ------------------------------------------------------------------------------------------------------------------
 * Stereo matcher.  replaces: match::stereo (src/openvslam/match/stereo.{h,cc}): the ctor's image pyramids are the two
 * extractors' image_pyramid_ members, which here never leave HBM -- the context reads the pyramids of the LAST extract of the
 * two ovs_orb handles (same device, same image size).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ovs_stereo ovs_stereo;
ovs_status ovs_stereo_create(int32_t max_rows, int32_t max_keypoints, int32_t device, ovs_stereo** out);
ovs_status ovs_stereo_destroy(ovs_stereo* s);
/* The two choices of stereo::compute that upstream's (absent) source decides (oracle/ORACLE_SPEC.md rule 20, tagged L), as run-time variants on
 * both sides (oracle: ovo_stereo_compute_v): OUTLIER_FACTOR 0 (default: matches with L1 distance > 2.0 x median dropped) | 1 (2.1 = ORB-SLAM2's
 * 1.5f * 1.4f); PARABOLA 0 (default: the sub-pixel parabola in float arithmetic) | 1 (in double, rounded to float once). */
#define OVS_STEREO_VARIANT_OUTLIER_FACTOR 0
#define OVS_STEREO_VARIANT_PARABOLA 1
ovs_status ovs_stereo_set_variant(ovs_stereo* s, int32_t which, int32_t value);
/* replaces: void stereo::compute(std::vector<float>& stereo_x_right, std::vector<float>& depths) const.
 * kps_* / desc_* = the keypoints and descriptors the two handles extracted (frame 0 of their last extract);
 * focal_x_baseline / true_baseline = camera->focal_x_baseline_ / true_baseline_. stereo_x_right / depths: n_left floats, -1 where
 * no match. *n_valid (may be NULL) = number of keypoints that received a depth. When the keypoints / descriptors are byte for byte what the
 * two handles' last ovs_orb_extract calls returned (the frame constructor's case), they are used where they already are -- in the extractors'
 * device output blocks -- and nothing is uploaded. */
ovs_status ovs_stereo_compute(ovs_stereo* s, const ovs_orb* left, const ovs_orb* right, const ovs_keypoint* kps_left,
                              const uint8_t* desc_left, int32_t n_left, const ovs_keypoint* kps_right, const uint8_t* desc_right,
                              int32_t n_right, float focal_x_baseline, float true_baseline, float* stereo_x_right, float* depths,
                              int32_t* n_valid);
/* Device-resident form: keypoints / descriptors are the extractors' device outputs (frame_left / frame_right of their last batched
 * extract), d_n_left / d_n_right their device counts (NULL: use cap_left / cap_right as the counts). Asynchronous on `stream`. */
ovs_status ovs_stereo_compute_dev(ovs_stereo* s, const ovs_orb* left, int32_t frame_left, const ovs_orb* right, int32_t frame_right,
                                  const ovs_keypoint* d_kps_left, const uint8_t* d_desc_left, const int32_t* d_n_left, int32_t cap_left,
                                  const ovs_keypoint* d_kps_right, const uint8_t* d_desc_right, const int32_t* d_n_right, int32_t cap_right,
                                  float focal_x_baseline, float true_baseline, float* d_stereo_x_right, float* d_depths,
                                  int32_t* d_n_valid, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Local bundle adjustment: residual + Jacobian + normal-equation blocks (one Levenberg-Marquardt linearisation).
 * replaces: the computeError / linearizeOplus / constructQuadraticForm loop g2o runs inside
 * optimize::local_bundle_adjuster::optimize (src/openvslam/optimize/local_bundle_adjuster.cc; edge math in
 * src/openvslam/optimize/g2o/se3/perspective_reproj_edge.{h,cc}). The sparse Schur solve stays on the host.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ovs_ba_cam {
    double fx, fy, cx, cy;   /* perspective, monocular observation (2 residuals) */
} ovs_ba_cam;

typedef struct ovs_ba_edge {   /* one observation: reproj_edge_wrapper */
    int32_t pose_idx, point_idx;
    double obs_x, obs_y;
    double inv_sigma_sq;       /* information = inv_level_sigma_sq[octave] * I2 */
} ovs_ba_edge;

typedef struct ovs_ba_edge_stereo {   /* one stereo observation: stereo_perspective_reproj_edge (3 residuals: u, v, u_right) */
    int32_t pose_idx, point_idx;
    double obs_x, obs_y, obs_x_right;
    double inv_sigma_sq;              /* information = inv_level_sigma_sq[octave] * I3 */
} ovs_ba_edge_stereo;

/* poses: n_pose x 7 = (tx,ty,tz,qx,qy,qz,qw), world->camera (g2o SE3Quat::toVector order); pose_fixed: NULL or n_pose bytes;
 * points: n_pt x 3. huber_delta <= 0 disables the robust kernel (the second optimisation round).
 * Outputs: Hpp n_pose x 36 (row-major 6x6, pose order omega then upsilon), bp n_pose x 6, Hll n_pt x 9, bl n_pt x 3,
 * Hpl n_edge x 18 (row-major 6x3 = Jp^T W Jl, zero for fixed poses), chi2[2] = {sum e^T Omega e, sum rho(.)}; b = -J^T W e.
 * For an edge SHARD (multi-GPU: edges partitioned by keyframe) the same call produces the shard's partial sums; only
 * Hll|bl need an all-reduce across shards (see openvslam_amd/ba.py). */
ovs_status ovs_ba_linearize(int32_t device, const double* poses, const uint8_t* pose_fixed, int32_t n_pose, const double* points,
                            int32_t n_pt, const ovs_ba_edge* edges, int32_t n_edge, const ovs_ba_cam* cam, double huber_delta,
                            double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* chi2);
/* Device-pointer form (cam stays a host pointer: it is passed by value to the kernel). Zeroes the outputs, asynchronous. */
ovs_status ovs_ba_linearize_dev(const double* d_poses, const uint8_t* d_pose_fixed, int32_t n_pose, const double* d_points,
                                int32_t n_pt, const ovs_ba_edge* d_edges, int32_t n_edge, const ovs_ba_cam* cam, double huber_delta,
                                double* d_Hpp, double* d_bp, double* d_Hll, double* d_bl, double* d_Hpl, double* d_chi2, void* stream);

/* Equirectangular edges (replaces: optimize::g2o::se3::equirectangular_reproj_edge::computeError / linearizeOplus,
 * src/openvslam/optimize/g2o/se3/equirectangular_reproj_edge.{h,cc}): e = z - (cols (1/2 + atan2(x, z) / 2 pi), rows (1/2 + asin(y / |p|) / pi)),
 * no wrap-around correction at the +-180 degree seam (oracle/ORACLE_SPEC.md rule 26). Same edge records and outputs as ovs_ba_linearize. */
ovs_status ovs_ba_linearize_equirect(int32_t device, const double* poses, const uint8_t* pose_fixed, int32_t n_pose, const double* points,
                                     int32_t n_pt, const ovs_ba_edge* edges, int32_t n_edge, int32_t cols, int32_t rows, double huber_delta,
                                     double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* chi2);
ovs_status ovs_ba_linearize_equirect_dev(const double* d_poses, const uint8_t* d_pose_fixed, int32_t n_pose, const double* d_points,
                                         int32_t n_pt, const ovs_ba_edge* d_edges, int32_t n_edge, int32_t cols, int32_t rows,
                                         double huber_delta, double* d_Hpp, double* d_bp, double* d_Hll, double* d_bl, double* d_Hpl,
                                         double* d_chi2, void* stream);
/* Stereo edges (replaces: optimize::g2o::se3::stereo_perspective_reproj_edge::computeError / linearizeOplus,
 * src/openvslam/optimize/g2o/se3/perspective_reproj_edge.{h,cc}): e = (u, v, u_r) - pi(RX + t), u_r = u - focal_x_baseline / z;
 * Huber delta sqrt(7.815) in local BA. Same outputs as ovs_ba_linearize (Hpl: n_edge x 18 for THESE edges). The device form can
 * accumulate (accumulate != 0) into blocks a preceding mono call produced, which is how a stereo / RGBD local BA mixes both. */
ovs_status ovs_ba_linearize_stereo(int32_t device, const double* poses, const uint8_t* pose_fixed, int32_t n_pose, const double* points,
                                   int32_t n_pt, const ovs_ba_edge_stereo* edges, int32_t n_edge, const ovs_ba_cam* cam,
                                   double focal_x_baseline, double huber_delta, double* Hpp, double* bp, double* Hll, double* bl,
                                   double* Hpl, double* chi2);
ovs_status ovs_ba_linearize_stereo_dev(const double* d_poses, const uint8_t* d_pose_fixed, int32_t n_pose, const double* d_points,
                                       int32_t n_pt, const ovs_ba_edge_stereo* d_edges, int32_t n_edge, const ovs_ba_cam* cam,
                                       double focal_x_baseline, double huber_delta, int32_t accumulate, double* d_Hpp, double* d_bp,
                                       double* d_Hll, double* d_bl, double* d_Hpl, double* d_chi2, void* stream);

/* Atomics-free form of the same linearisation (round 2): the edge set is indexed ONCE (by landmark and by keyframe) into a graph handle
 * resident in HBM; every call then evaluates Hll | bl by one lane per landmark over its edges in ascending index (exactly the order a
 * sequential loop adds them: bit-identical to the CPU oracle), Hpl per edge, Hpp | bp by one workgroup per keyframe with a fixed-shape
 * tree, chi2 by a fixed tree -- identical bits from run to run, no fp64 atomics. mono / stereo edges as for ovs_ba_linearize(_stereo);
 * Hpl rows: the mono edges in input order, then the stereo edges. d_chi2 receives THREE doubles: chi2, robustified chi2, and
 * max |diagonal| over the free pose blocks and the landmarks that have edges (g2o's computeLambdaInit input). huber_mono / huber_stereo:
 * the Huber deltas of the two edge kinds (0 = no kernel). For an edge shard (multi-GPU: edges partitioned by keyframe) build the graph
 * from the shard; Hll | bl | chi2 are then partial sums to be all-reduced, Hpp | bp are complete on the shard that owns the keyframe. */
typedef struct ovs_ba_graph ovs_ba_graph;
ovs_status ovs_ba_graph_create(int32_t device, int32_t n_pose, const uint8_t* pose_fixed, int32_t n_pt, const ovs_ba_edge* mono, int32_t n_mono,
                               const ovs_ba_edge_stereo* stereo, int32_t n_stereo, const ovs_ba_cam* cam, double focal_x_baseline,
                               ovs_ba_graph** out);
/* Equirectangular graph (replaces: the equirectangular_reproj_edge branch of local_bundle_adjuster::optimize's graph build): every edge is a
 * monocular equirectangular edge as in ovs_ba_linearize_equirect (cols / rows = camera->cols_ / rows_). Same handle, same linearize call
 * (huber_stereo is unused). */
ovs_status ovs_ba_graph_create_equirect(int32_t device, int32_t n_pose, const uint8_t* pose_fixed, int32_t n_pt, const ovs_ba_edge* mono,
                                        int32_t n_mono, int32_t cols, int32_t rows, ovs_ba_graph** out);
ovs_status ovs_ba_graph_destroy(ovs_ba_graph* g);
/* The graphs' device arenas are recycled through a small per-device pool (mapping_module builds a graph per keyframe; upstream's
 * local_bundle_adjuster::optimize builds and drops a g2o::SparseOptimizer per call). A long-lived process that is done with local BA
 * returns the kept blocks (at most four per device) with this call; graphs that still exist are not touched. */
ovs_status ovs_ba_pool_trim(void);
ovs_status ovs_ba_graph_linearize_dev(ovs_ba_graph* g, const double* d_poses, const double* d_points, double huber_mono, double huber_stereo,
                                      double* d_Hpp, double* d_bp, double* d_Hll, double* d_bl, double* d_Hpl, double* d_chi2, void* stream);

/* Natively sharded linearisation (BASELINE config 5): ONE process drives devices 0 .. n_gpus-1 -- the shape in which
 * mapping_module would use it. Edges are partitioned by keyframe into n_gpus contiguous keyframe blocks (one ovs_ba_graph per device); a
 * call uploads the state to every device, linearises the shards concurrently and sums the landmark blocks with ONE packed RCCL all-reduce
 * of Hll | bl | chi2 over xGMI; Hpp | bp and Hpl are complete on the shard that owns the keyframe and are only collected. Host pointers
 * in / out, same layouts as ovs_ba_linearize (Hpl: mono edges in input order, then stereo; chi2: 2 doubles). RCCL is loaded lazily
 * (dlopen) and only for n_gpus > 1; n_gpus = 1 runs the same code without a communicator. Results equal the single-device graph path
 * within 1e-10 relative (the all-reduce changes the association of the landmark sums); Hpl is bit-identical. */
typedef struct ovs_ba_multi ovs_ba_multi;
ovs_status ovs_ba_multi_create(int32_t n_gpus, int32_t n_pose, const uint8_t* pose_fixed, int32_t n_pt, const ovs_ba_edge* mono, int32_t n_mono,
                               const ovs_ba_edge_stereo* stereo, int32_t n_stereo, const ovs_ba_cam* cam, double focal_x_baseline,
                               ovs_ba_multi** out);
ovs_status ovs_ba_multi_destroy(ovs_ba_multi* m);
/* How the landmark blocks are summed across the devices (SURVEY 8(e) asks for both to be measured): one packed ncclAllReduce (default), or
 * a direct exchange -- every device reads the peers' packed blocks over xGMI through peer-mapped pointers and sums them itself in a fixed
 * device order (bit-identical on every device and from run to run). OVS_ERR_NO_DEVICE if some pair of devices has no peer access. */
#define OVS_BA_EXCHANGE_RCCL 0
#define OVS_BA_EXCHANGE_PEER 1
ovs_status ovs_ba_multi_set_exchange(ovs_ba_multi* m, int32_t exchange);
ovs_status ovs_ba_multi_linearize(ovs_ba_multi* m, const double* poses, const double* points, double huber_mono, double huber_stereo, double* Hpp,
                                  double* bp, double* Hll, double* bl, double* Hpl, double* chi2);

/* ------------------------------------------------------------------------------------------------------------------
 * Bag-of-words transform.  replaces: the per-descriptor tree descent of DBoW2::TemplatedVocabulary::transform(
 *   const std::vector<TDescriptor>& features, BowVector& v, FeatureVector& fv, int levelsup) as called by data::frame::compute_bow /
 *   data::keyframe::compute_bow (src/openvslam/data/frame.cc, keyframe.cc; levelsup = 4). DBoW2 is a third-party dependency.
 * The vocabulary is handed over as a tree: node 0 = root, child_start[n_nodes + 1] / children = CSR of child node ids (DBoW2 order),
 * node_desc n_nodes x 32 bytes, node_weight (word weights; inner nodes ignored), node_word_id (-1 for inner nodes), depth = L.
 * Per feature: word_id, weight (the leaf's) and node_id (the ancestor at level L - levelsup, 0 if levelsup >= L). The BowVector
 * (v[word] += weight in feature order, then L1-normalised) and the FeatureVector (fv[node].push_back(i)) are std::maps the shim fills
 * from these arrays exactly as DBoW2 does. The _dev form takes the extractor's device outputs (descriptors B x cap x 32, counts B). */
typedef struct ovs_vocab ovs_vocab;
ovs_status ovs_vocab_create(int32_t device, int32_t n_nodes, const int32_t* child_start, const int32_t* children, const uint8_t* node_desc,
                            const double* node_weight, const int32_t* node_word_id, int32_t depth, int32_t max_features, ovs_vocab** out);
ovs_status ovs_vocab_destroy(ovs_vocab* v);
/* On-disk vocabularies (replaces: DBoW2::TemplatedVocabulary::loadFromBinaryFile / loadFromTextFile and fbow::Vocabulary::readFromFile as
 * called by system.cc for data::bow_vocabulary). The format is detected from the content: FBoW (.fbow), the DBoW2 fork's binary (.dbow2)
 * or DBoW2 / ORB-SLAM2 text. ovs_vocab_tree_* parse on the host only (no device needed) into the arrays ovs_vocab_create takes;
 * ovs_vocab_load_file = parse + create. Layouts: csrc/bow_vocab_io.hip; FBoW inner nodes get ids in block order (ORACLE_SPEC rule 30). */
#define OVS_VOCAB_DBOW2_TEXT 1
#define OVS_VOCAB_DBOW2_BINARY 2
#define OVS_VOCAB_FBOW 3
typedef struct ovs_vocab_tree ovs_vocab_tree;
ovs_status ovs_vocab_tree_load(const char* path, ovs_vocab_tree** out, int32_t* format, int32_t* n_nodes, int32_t* depth);
/* any pointer may be NULL; sizes: child_start n_nodes + 1, children n_nodes - 1, node_desc n_nodes x 32, node_weight / node_word_id n_nodes */
ovs_status ovs_vocab_tree_arrays(const ovs_vocab_tree* t, int32_t* child_start, int32_t* children, uint8_t* node_desc, double* node_weight,
                                 int32_t* node_word_id);
ovs_status ovs_vocab_tree_free(ovs_vocab_tree* t);
ovs_status ovs_vocab_load_file(int32_t device, const char* path, int32_t max_features, ovs_vocab** out, int32_t* format);
ovs_status ovs_bow_transform(ovs_vocab* v, const uint8_t* desc, int32_t n, int32_t levelsup, int32_t* word_id, double* weight,
                             int32_t* node_id);
ovs_status ovs_bow_transform_dev(ovs_vocab* v, const uint8_t* d_desc, const int32_t* d_counts, int32_t batch, int32_t cap, int32_t levelsup,
                                 int32_t* d_word_id, double* d_weight, int32_t* d_node_id, void* stream);

/* replaces: the optimisation inside  void optimize::local_bundle_adjuster::optimize(data::keyframe* curr_keyfrm, bool* const
 *               force_stop_flag) const  (src/openvslam/optimize/local_bundle_adjuster.{h,cc}): everything between the graph build and the
 * write-back, i.e. optimizer.optimize(num_first_iter) with Huber kernels (ONE delta per rig: setup_type 0 = Monocular -> sqrtf(5.99146f),
 * otherwise sqrtf(7.81473f), as upstream picks it from keyfrm->camera_->setup_type_), the chi-square (5.99146f mono / 7.81473f stereo edge) /
 * depth-positive outlier test that moves edges to level 1 and drops the kernels, optimizer.optimize(num_second_iter), and the final
 * outlier test. g2o's Levenberg-Marquardt schedule and BlockSolver_6_3's landmark elimination are restated (oracle/ORACLE_SPEC.md
 * rules 25, 28); linearisations run on the device (ba_linearize kernels), and since round 4 so does the solve of the reduced camera
 * system (csrc/ba_solve.hip: one-workgroup blocked Cholesky on the f64 matrix cores; an LM trial reads back three scalars and a flag).
 * ovs_local_ba_set_solver(1) selects the host Cholesky of rounds 1-3 instead (what BASELINE's north star describes; also the path of
 * systems beyond 1024 unknowns): same schedule, results equal to ~1e-10 relative (tests/test_gpu_ba.py).
 * poses (n_pose x 7, in/out; fixed ones untouched), points (n_pt x 3, in/out), mono / stereo edges as for ovs_ba_linearize(_stereo).
 * force_stop_flag: NULL or the caller's flag, polled between iterations. mono_outlier / stereo_outlier: 1 for the observations the
 * caller must erase. info: NULL or 6 doubles {robust chi2 before / after round 1, chi2 before / after round 2, iterations 1, 2}. */
ovs_status ovs_local_ba_optimize(int32_t device, double* poses, const uint8_t* pose_fixed, int32_t n_pose, double* points, int32_t n_pt,
                                 const ovs_ba_edge* mono, int32_t n_mono, const ovs_ba_edge_stereo* stereo, int32_t n_stereo,
                                 const ovs_ba_cam* cam, double focal_x_baseline, int32_t setup_type, int32_t num_first_iter, int32_t num_second_iter,
                                 const volatile uint8_t* force_stop_flag, uint8_t* mono_outlier, uint8_t* stereo_outlier, double* info);

/* Equirectangular local map (camera::model_type_t::Equirectangular keyframes: optimize::g2o::se3::equirectangular_reproj_edge, every edge
 * monocular, Monocular rig -> Huber sqrtf(5.99146f), gate 5.99146f; depth_is_positive() is always true for this model). Same schedule,
 * outputs and info as ovs_local_ba_optimize. */
ovs_status ovs_local_ba_optimize_equirect(int32_t device, double* poses, const uint8_t* pose_fixed, int32_t n_pose, double* points, int32_t n_pt,
                                          const ovs_ba_edge* mono, int32_t n_mono, int32_t cols, int32_t rows, int32_t num_first_iter,
                                          int32_t num_second_iter, const volatile uint8_t* force_stop_flag, uint8_t* mono_outlier, double* info);
/* where ovs_local_ba_optimize(_equirect) solves the reduced camera system: 0 = device (default), 1 = host. Process-wide. */
ovs_status ovs_local_ba_set_solver(int32_t where);
int32_t ovs_local_ba_get_solver(void);
/* The device solver alone on host arrays (test entry): S (n x n, row-major, symmetric positive definite; the lower triangle is read),
 * rhs (n) -> x (n); n <= 1024. OVS_ERR_INVALID when a pivot is not positive. */
ovs_status ovs_ba_dense_solve(int32_t device, const double* S, const double* rhs, int32_t n, double* x);

/* ------------------------------------------------------------------------------------------------------------------
 * Pose-only optimisation of one frame.  replaces: unsigned int optimize::pose_optimizer::optimize(data::frame& frm) const
 * (src/openvslam/optimize/pose_optimizer.{h,cc}; perspective mono / stereo pose_opt edges): 4 rounds x 10 Levenberg-Marquardt
 * iterations with Huber kernels in the first three rounds and chi2 outlier re-classification (5.99146f / 7.81473f) after every round, in ONE
 * kernel launch. The shim flattens the frame's landmarks into ovs_pose_obs records, writes pose_cw_out back with
 * frm.set_cam_pose(...), outlier_flags into frm.outlier_flags_ and returns *num_valid.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ovs_pose_obs {   /* one observed landmark: pose_opt_edge_wrapper */
    double pos_w[3];
    double obs_x, obs_y, obs_x_right;   /* undistorted keypoint; obs_x_right is read only when is_stereo != 0 */
    double inv_sigma_sq;                /* frm.inv_level_sigma_sq_[octave] */
    int32_t is_stereo, pad;
} ovs_pose_obs;
/* pose_cw_*: 12 doubles (rotation row-major, translation), world -> camera. n_obs <= 8192 (OVS_ERR_CAPACITY above; the _dev form
 * reports a frame with more observations as d_num_valid = -1). setup_type = frm.camera_->setup_type_ (0 Monocular, 1 Stereo, 2 RGBD):
 * upstream picks ONE Huber delta per frame from it (Monocular: sqrtf(5.99146f), otherwise sqrtf(7.81473f)); the chi-square outlier
 * gates (5.99146f / 7.81473f) stay per observation (is_stereo). */
ovs_status ovs_pose_optimize(int32_t device, const double* pose_cw_in, const ovs_pose_obs* obs, int32_t n_obs, const ovs_ba_cam* cam,
                             double focal_x_baseline, int32_t setup_type, double* pose_cw_out, uint8_t* outlier_flags, int32_t* num_valid);
/* Device-resident batch: frame p owns observations [d_obs_offsets[p], d_obs_offsets[p + 1]); one workgroup per frame. */
/* oracle/ORACLE_SPEC.md rule 25 (iv), tagged L, as a run-time variant (process-wide; the oracle has ovo_pose_set_variant): RESET_EACH_ROUND
 * 0 (default: the frame vertex is initialised once, every round continues from the previous round's estimate -- OpenVSLAM as recalled) | 1 (the
 * estimate is re-set to pose_cw_in at the start of each of the four rounds -- ORB-SLAM2's Optimizer::PoseOptimization). */
#define OVS_POSE_VARIANT_RESET_EACH_ROUND 0
ovs_status ovs_pose_set_variant(int32_t which, int32_t value);
ovs_status ovs_pose_optimize_batch_dev(const double* d_poses_in, const ovs_pose_obs* d_obs, const int32_t* d_obs_offsets, int32_t batch,
                                       const ovs_ba_cam* cam, double focal_x_baseline, int32_t setup_type, double* d_poses_out,
                                       uint8_t* d_outlier, int32_t* d_num_valid, void* stream);
/* Equirectangular frames.  replaces: the `camera::model_type_t::Equirectangular` branch of pose_optimizer::optimize, i.e.
 * optimize::g2o::se3::equirectangular_pose_opt_edge::computeError / linearizeOplus (src/openvslam/optimize/g2o/se3/
 * equirectangular_pose_opt_edge.{h,cc}): e = z - (cols (1/2 + atan2(x, z) / 2 pi), rows (1/2 + asin(y / |p|) / pi)), no wrap-around at the
 * +-180 degree seam (oracle/ORACLE_SPEC.md rule 26). Every edge is monocular and the rig is Monocular (Huber sqrtf(5.99146f), gate
 * 5.99146f); obs[i].is_stereo / obs_x_right are ignored. cols / rows = camera->cols_ / rows_. Same schedule, outputs and limits. */
ovs_status ovs_pose_optimize_equirect(int32_t device, const double* pose_cw_in, const ovs_pose_obs* obs, int32_t n_obs, int32_t cols, int32_t rows,
                                      double* pose_cw_out, uint8_t* outlier_flags, int32_t* num_valid);
ovs_status ovs_pose_optimize_equirect_batch_dev(const double* d_poses_in, const ovs_pose_obs* d_obs, const int32_t* d_obs_offsets, int32_t batch,
                                                int32_t cols, int32_t rows, double* d_poses_out, uint8_t* d_outlier, int32_t* d_num_valid,
                                                void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Keyframe residency (round 4).  data::keyframe copies undist_keypts_, descriptors_, stereo_x_right_, bearings_ and the keypoint grid from
 * the frame it is made of and never changes them (expected: src/openvslam/data/keyframe.{h,cc}), while mapping_module and the loop closer
 * hand the same keyframe to a matcher again and again: fuse::replace_duplication over ~20 covisible keyframes per new keyframe
 * (expected: src/openvslam/mapping_module.cc::fuse_landmark_duplication), bow_tree / robust::match_for_triangulation per neighbour
 * (create_new_landmarks), match_keyframes_mutually and match_by_Sim3_transform per loop candidate. The *_f twins below take an
 * ovs_frame_dev for every (grid parameters, keypoints, descriptors, stereo_x_right, n) group of their host-array forms -- the handle
 * is created once per keyframe (same constructor as for a frame) and shared by every thread on that device; nothing else about the
 * calls changes, results are bit-identical to the host-array forms. replaces: the same upstream functions as the forms they twin.
 * ------------------------------------------------------------------------------------------------------------------ */
/* 3 doubles per keypoint (data::keyframe::bearings_) for ovs_robust_match_for_triangulation_f: uploaded once. Call it before the handle is
 * shared between threads (the class shims do it inside the keyframe cache's creation lock). */
ovs_status ovs_frame_dev_attach_bearings(ovs_frame_dev* f, const double* bearings);
int32_t ovs_frame_dev_device(const ovs_frame_dev* f);   /* the device the handle lives on (-1 for NULL) */
/* bow_tree::match_frame_and_keyframe (src/openvslam/match/bow_tree.cc): keyframe and frame resident; BoW feature vectors as in the host form */
ovs_status ovs_bow_match_frame_and_keyframe_f(ovs_wmatcher* w, const ovs_frame_dev* keyfrm, const uint8_t* kf_valid, const int32_t* kf_node_ids,
                                              const int32_t* kf_node_start, const int32_t* kf_items, int32_t kf_nodes, const ovs_frame_dev* frm,
                                              const int32_t* frm_node_ids, const int32_t* frm_node_start, const int32_t* frm_items, int32_t frm_nodes,
                                              float lowe_ratio, int32_t check_orientation, int32_t* matched_kf_in_frm, int32_t* num_matches);
/* bow_tree::match_keyframes */
ovs_status ovs_bow_match_keyframes_f(ovs_wmatcher* w, const ovs_frame_dev* keyfrm_1, const uint8_t* valid_1, const int32_t* node_ids_1,
                                     const int32_t* node_start_1, const int32_t* items_1, int32_t nodes_1, const ovs_frame_dev* keyfrm_2,
                                     const uint8_t* valid_2, const int32_t* node_ids_2, const int32_t* node_start_2, const int32_t* items_2, int32_t nodes_2,
                                     float lowe_ratio, int32_t check_orientation, int32_t* matched_2_in_1, int32_t* num_matches);
/* robust::match_for_triangulation (src/openvslam/match/robust.cc): both handles must carry bearings (ovs_frame_dev_attach_bearings) */
ovs_status ovs_robust_match_for_triangulation_f(ovs_wmatcher* w, const ovs_frame_dev* keyfrm_1, const uint8_t* has_lm_1, const int32_t* node_ids_1,
                                                const int32_t* node_start_1, const int32_t* items_1, int32_t nodes_1, const ovs_frame_dev* keyfrm_2,
                                                const uint8_t* has_lm_2, const int32_t* node_ids_2, const int32_t* node_start_2, const int32_t* items_2,
                                                int32_t nodes_2, const double* E_12, const double* epipole_in_2, const float* scale_factors,
                                                int32_t num_levels, int32_t check_orientation, int32_t* matched_2_in_1, int32_t* num_matches);
/* fuse::replace_duplication (src/openvslam/match/fuse.cc) */
ovs_status ovs_fuse_replace_duplication_f(ovs_wmatcher* w, const ovs_camera* cam, const ovs_frame_dev* keyfrm, const double* pose_cw, const double* lm_pos_w,
                                          const float* lm_dist_min_max, const double* lm_normal, const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m,
                                          const float* scale_factors, const float* inv_level_sigma_sq, int32_t num_levels, float log_scale_factor,
                                          float margin, int32_t* best_idx, int32_t* num_fused);
/* fuse::detect_duplication */
ovs_status ovs_fuse_detect_duplication_f(ovs_wmatcher* w, const ovs_camera* cam, const ovs_frame_dev* keyfrm, const double* sim3_cw, const double* lm_pos_w,
                                         const float* lm_dist_min_max, const double* lm_normal, const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m,
                                         const float* scale_factors, int32_t num_levels, float log_scale_factor, float margin, int32_t* best_idx,
                                         int32_t* num_found);
/* projection::match_frame_and_keyframe (src/openvslam/match/projection.cc): the CURRENT FRAME resident (the keyframe side is its landmarks) */
ovs_status ovs_projection_match_frame_and_keyframe_f(ovs_wmatcher* w, const ovs_camera* cam, const ovs_frame_dev* curr, const uint8_t* curr_occupied,
                                                     const double* pose_cw_curr, const ovs_keypoint* kf_kps, const double* kf_pos_w,
                                                     const float* kf_dist_min_max, const uint8_t* kf_lm_desc, const uint8_t* kf_valid, int32_t n_kf,
                                                     const float* scale_factors, int32_t num_levels, float log_scale_factor, float margin,
                                                     uint32_t hamm_dist_thr, int32_t check_orientation, int32_t* assigned, int32_t* num_matches);
/* projection::match_by_Sim3_transform */
ovs_status ovs_projection_match_by_sim3_transform_f(ovs_wmatcher* w, const ovs_camera* cam, const ovs_frame_dev* keyfrm, const uint8_t* occupied,
                                                    const double* sim3_cw, const double* lm_pos_w, const float* lm_dist_min_max, const double* lm_normal,
                                                    const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m, const float* scale_factors,
                                                    int32_t num_levels, float log_scale_factor, float margin, int32_t* assigned, int32_t* num_matches);
/* projection::match_keyframes_mutually */
ovs_status ovs_projection_match_keyframes_mutually_f(ovs_wmatcher* w, const ovs_camera* cam_1, const ovs_frame_dev* keyfrm_1, const double* pose_cw_1,
                                                     const double* lm_pos_w_1, const float* lm_dist_1, const uint8_t* lm_desc_1, const uint8_t* lm_valid_1,
                                                     const ovs_camera* cam_2, const ovs_frame_dev* keyfrm_2, const double* pose_cw_2,
                                                     const double* lm_pos_w_2, const float* lm_dist_2, const uint8_t* lm_desc_2, const uint8_t* lm_valid_2,
                                                     double s_12, const double* rot_12, const double* trans_12, const float* scale_factors,
                                                     int32_t num_levels, float log_scale_factor, float margin, int32_t* matched_2_in_1,
                                                     int32_t* num_matches);

/* ------------------------------------------------------------------------------------------------------------------
 * Self-test of include/ovs_detmath.h on the device: out[i] = fn(a[i] (, b[i])) evaluated by a gfx950 kernel. The four functions
 * replace libm calls upstream makes where a float decides a match pair (landmark::predict_scale_level's std::log(float),
 * camera::equirectangular::reproject_to_image's asin / atan2, match::robust::check_epipolar_constraint's acos); the same header is
 * compiled into the CPU oracle, and tests require identical bits from both. logf takes / returns floats widened to double.
 * ------------------------------------------------------------------------------------------------------------------ */
#define OVS_DETMATH_LOGF 0
#define OVS_DETMATH_ASIN 1
#define OVS_DETMATH_ACOS 2
#define OVS_DETMATH_ATAN2 3   /* atan2(a, b) */
#define OVS_DETMATH_SINF 4    /* float in, float out (widened): the OVS_VARIANT_TRIG = 1 steering */
#define OVS_DETMATH_COSF 5
ovs_status ovs_detmath_eval(int32_t device, int32_t fn, const double* a, const double* b, double* out, int32_t n);

#ifdef __cplusplus
}
#endif
#endif
