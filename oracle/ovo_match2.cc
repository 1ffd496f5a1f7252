// ovo_match2.cc -- CPU ORACLE (test infrastructure, see ovo_oracle.h): the windowed matchers and their candidate generator,
// restated from spec. PARITY UNPINNED (upstream absent). Expected upstream paths:
//   D1  src/openvslam/data/common.{h,cc}            assign_keypoints_to_grid, get_cell_indices, get_keypoints_in_cell
//   --  src/openvslam/match/angle_checker.h         30-bin rotation histogram, keep the 3 fullest bins
//   M3  src/openvslam/match/projection.{h,cc}       match_frame_and_landmarks
//   M4  src/openvslam/match/projection.{h,cc}       match_current_and_last_frames (perspective camera)
//   M5  src/openvslam/match/area.{h,cc}             match_in_consistent_area
//   M6  src/openvslam/match/stereo.{h,cc}           compute (+ get_right_keypoint_indices_in_each_row,
//                                                   find_closest_keypoints_in_stereo, compute_subpixel_disparity)
//   M7  src/openvslam/match/bow_tree.{h,cc}         match_frame_and_keyframe
// Object graphs (data::frame, data::landmark*, camera::base) are flattened into SoA arrays exactly as the C ABI of
// include/ovslam_hip.h flattens them; every rule that had to be chosen is listed in ORACLE_SPEC.md (rules 16+).
#include "ovo_oracle.h"
#include "../include/ovs_detmath.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <numeric>
#include <utility>
#include <vector>

namespace {

inline uint32_t distance_32(const uint8_t* a, const uint8_t* b) { return ovo_descriptor_distance_32(a, b); }

inline int cv_round(float v) { return (int)std::nearbyintf(v); }   // cvRound: round-half-to-even (SSE cvtss2si)
inline int cv_floor(float v) { return (int)std::floor(v); }
inline int cv_ceil(float v) { return (int)std::ceil(v); }

struct Grid {
    ovo_grid_params p;
    float inv_w, inv_h;
    std::vector<int32_t> start, items;   // cell id = cx * rows + cy (upstream: keypt_indices_in_cells[cx][cy])
};

// camera::base: inv_cell_width_ = num_grid_cols_ / (img_bounds_.max_x_ - img_bounds_.min_x_) (double division, float member)
inline void grid_scales(const ovo_grid_params& p, float& inv_w, float& inv_h) {
    inv_w = (float)((double)p.cols / (double)(p.max_x - p.min_x));
    inv_h = (float)((double)p.rows / (double)(p.max_y - p.min_y));
}

// D1 get_cell_indices: cvRound of the scaled coordinate; keypoints falling outside the grid are in no cell.
inline bool cell_of(const Grid& g, float x, float y, int& cx, int& cy) {
    cx = cv_round((x - g.p.min_x) * g.inv_w);
    cy = cv_round((y - g.p.min_y) * g.inv_h);
    return 0 <= cx && cx < g.p.cols && 0 <= cy && cy < g.p.rows;
}

// D1 assign_keypoints_to_grid: push_back in keypoint order => ascending indices inside a cell.
void build_grid(Grid& g, const ovo_grid_params& p, const float* xs, const float* ys, int n) {
    g.p = p;
    grid_scales(p, g.inv_w, g.inv_h);
    const int nc = p.cols * p.rows;
    g.start.assign(nc + 1, 0);
    std::vector<int32_t> cell(n, -1);
    for (int i = 0; i < n; ++i) {
        int cx, cy;
        if (cell_of(g, xs[i], ys[i], cx, cy)) {
            cell[i] = cx * p.rows + cy;
            ++g.start[cell[i] + 1];
        }
    }
    for (int c = 0; c < nc; ++c) g.start[c + 1] += g.start[c];
    g.items.assign(g.start[nc], 0);
    std::vector<int32_t> fill(g.start.begin(), g.start.end() - 1);
    for (int i = 0; i < n; ++i)
        if (cell[i] >= 0) g.items[fill[cell[i]]++] = i;
}

// D1 get_keypoints_in_cell: cells [floor((ref - min - margin) * inv), ceil((ref - min + margin) * inv)] clamped to the grid,
// x-major then y, members in cell order, filtered by level range and |dx| < margin && |dy| < margin.
template <typename F>
void for_keypoints_in_cell(const Grid& g, const float* xs, const float* ys, const int32_t* octaves, float ref_x, float ref_y,
                           float margin, int min_level, int max_level, F&& f) {
    const int min_cx = std::max(0, cv_floor((ref_x - g.p.min_x - margin) * g.inv_w));
    if (g.p.cols <= min_cx) return;
    const int max_cx = std::min(g.p.cols - 1, cv_ceil((ref_x - g.p.min_x + margin) * g.inv_w));
    if (max_cx < 0) return;
    const int min_cy = std::max(0, cv_floor((ref_y - g.p.min_y - margin) * g.inv_h));
    if (g.p.rows <= min_cy) return;
    const int max_cy = std::min(g.p.rows - 1, cv_ceil((ref_y - g.p.min_y + margin) * g.inv_h));
    if (max_cy < 0) return;
    const bool check_level = (0 < min_level) || (0 <= max_level);
    for (int cx = min_cx; cx <= max_cx; ++cx) {
        for (int cy = min_cy; cy <= max_cy; ++cy) {
            const int c = cx * g.p.rows + cy;
            for (int k = g.start[c]; k < g.start[c + 1]; ++k) {
                const int idx = g.items[k];
                if (check_level) {
                    if (octaves[idx] < min_level || (0 <= max_level && max_level < octaves[idx])) continue;
                }
                const float dist_x = xs[idx] - ref_x, dist_y = ys[idx] - ref_y;
                if (std::fabs(dist_x) < margin && std::fabs(dist_y) < margin) f(idx);
            }
        }
    }
}

// match::angle_checker<int>: histogram_length 30, inv_histogram_length = 1/30 (so bins are 30 degrees wide and only
// bins 0..12 are ever hit -- upstream's inherited ORB-SLAM2 quirk), keep the num_bins_to_keep = 3 fullest bins.
// Tie rule (upstream: std::sort on sizes, implementation-defined): equal sizes -> lower bin index first.
// ORACLE_SPEC rule 17 as a run-time variant (process-wide, ovo_match_set_variant): 0 (default) the three fullest bins are kept whatever their
// sizes; 1 ORB-SLAM2's ComputeThreeMaxima: the second (and with it the third) is dropped when it holds fewer than 0.1 x the fullest bin's
// entries, the third alone when only it does
int g_angle_keep_rule = 0;
// rule 17's tie order as a variant: 0 (default) of two equally full bins the lower one ranks first, 1 the higher one (upstream: std::sort on the
// sizes, implementation-defined for equal sizes)
int g_angle_tie_order = 0;
struct AngleChecker {
    static constexpr int kLen = 30, kKeep = 3;
    std::vector<int> bins[kLen];
    void append(float delta_angle, int match) {
        if (delta_angle < 0.0f) delta_angle += 360.0f;
        if (360.0f <= delta_angle) delta_angle -= 360.0f;
        int bin = cv_round(delta_angle * (1.0f / kLen));
        if (bin == kLen) bin = 0;
        bins[bin].push_back(match);
    }
    std::vector<int> invalid() const {
        int order[kLen];
        std::iota(order, order + kLen, 0);
        std::stable_sort(order, order + kLen, [&](int a, int b) {
            return bins[a].size() > bins[b].size() || (g_angle_tie_order && bins[a].size() == bins[b].size() && a > b);
        });
        int n_keep = kKeep;
        if (g_angle_keep_rule) {
            const float max1 = (float)bins[order[0]].size();
            if ((float)bins[order[1]].size() < 0.1f * max1) n_keep = 1;
            else if ((float)bins[order[2]].size() < 0.1f * max1) n_keep = 2;
        }
        std::vector<int> out;
        for (int b = 0; b < kLen; ++b) {
            bool keep = false;
            for (int k = 0; k < n_keep; ++k) keep |= order[k] == b;
            if (!keep) out.insert(out.end(), bins[b].begin(), bins[b].end());
        }
        return out;
    }
};

}   // namespace

extern "C" {

int ovo_assign_keypoints_to_grid(const ovo_grid_params* p, const float* xs, const float* ys, int n, int32_t* cell_start,
                                 int32_t* items) {
    Grid g;
    build_grid(g, *p, xs, ys, n);
    std::memcpy(cell_start, g.start.data(), sizeof(int32_t) * g.start.size());
    if (!g.items.empty()) std::memcpy(items, g.items.data(), sizeof(int32_t) * g.items.size());
    return (int)g.items.size();
}

int ovo_get_keypoints_in_cell(const ovo_grid_params* p, const float* xs, const float* ys, const int32_t* octaves, int n, float ref_x,
                              float ref_y, float margin, int min_level, int max_level, int32_t* out, int cap) {
    Grid g;
    build_grid(g, *p, xs, ys, n);
    int m = 0;
    for_keypoints_in_cell(g, xs, ys, octaves, ref_x, ref_y, margin, min_level, max_level, [&](int idx) {
        if (m < cap) out[m] = idx;
        ++m;
    });
    return m;
}

int ovo_match_set_variant(int which, int value) {
    if ((which != 0 && which != 1) || (value != 0 && value != 1)) return -1;
    (which == 0 ? g_angle_keep_rule : g_angle_tie_order) = value;
    return 0;
}

void ovo_angle_checker_invalid(const float* delta_angles, int n, uint8_t* invalid) {
    AngleChecker ac;
    for (int i = 0; i < n; ++i) ac.append(delta_angles[i], i);
    std::memset(invalid, 0, (size_t)n);
    for (int i : ac.invalid()) invalid[i] = 1;
}

// M3  projection::match_frame_and_landmarks(frm, local_landmarks, margin): landmarks in order; candidates from the grid at
// the reprojection with radius margin * scale_factors[pred_level], levels [pred_level-1, pred_level]; keypoints that
// already hold a landmark with observations are skipped (including those assigned earlier in this very loop); stereo
// keypoints must agree on x_right; strict `<` best/second WITH their levels; accept iff best <= THR_HIGH and not
// (best_level == second_level && best > lowe_ratio * second); then frm.landmarks_[best_idx] = lm.
// assigned[l] = frame keypoint index given to landmark l, or -1. Returns num_matches.
int ovo_projection_match_frame_and_landmarks(const ovo_grid_params* gp, const float* xs, const float* ys, const int32_t* octaves,
                                             const float* stereo_x_right, const uint8_t* desc, const uint8_t* occupied, int n,
                                             const float* lm_x, const float* lm_y, const float* lm_x_right, const int32_t* lm_level,
                                             const uint8_t* lm_desc, const uint8_t* lm_valid, int m, const float* scale_factors,
                                             float margin, float lowe_ratio, int32_t* assigned) {
    Grid g;
    build_grid(g, *gp, xs, ys, n);
    std::vector<uint8_t> occ(occupied, occupied + n);
    int num_matches = 0;
    for (int l = 0; l < m; ++l) {
        assigned[l] = -1;
        if (lm_valid && !lm_valid[l]) continue;   // !is_observable_in_tracking_ || will_be_erased()
        const int pred = lm_level[l];
        const float r = margin * scale_factors[pred];
        unsigned best = OVO_MAX_HAMMING_DIST, second = OVO_MAX_HAMMING_DIST;
        int best_level = -1, second_level = -1, best_idx = -1;
        for_keypoints_in_cell(g, xs, ys, octaves, lm_x[l], lm_y[l], r, pred - 1, pred, [&](int idx) {
            if (occ[idx]) return;
            if (stereo_x_right && 0 < stereo_x_right[idx]) {
                const float reproj_error = std::fabs(lm_x_right[l] - stereo_x_right[idx]);
                if (r < reproj_error) return;
            }
            const unsigned d = distance_32(lm_desc + (size_t)l * 32, desc + (size_t)idx * 32);
            if (d < best) {
                second = best;
                best = d;
                second_level = best_level;
                best_level = octaves[idx];
                best_idx = idx;
            } else if (d < second) {
                second_level = octaves[idx];
                second = d;
            }
        });
        if (best <= OVO_HAMMING_DIST_THR_HIGH) {
            if (best_level == second_level && (float)best > lowe_ratio * (float)second) continue;
            assigned[l] = best_idx;
            occ[best_idx] = 1;
            ++num_matches;
        }
    }
    return num_matches;
}

// M5  area::match_in_consistent_area(frm_1, frm_2, prev_matched_pts, matched_indices_2_in_frm_1, margin): level-0 keypoints
// of frame 1 only; candidates of frame 2 around the previous matched position, level [0, 0]; a candidate already matched
// at a distance <= ours is skipped; accept iff best <= THR_LOW and !(second * ratio < best); an earlier owner of the
// target is unmatched (stolen); optional orientation histogram; finally prev_matched_pts is updated for the matches.
int ovo_area_match_in_consistent_area(const ovo_grid_params* gp, const int32_t* octaves_1, const float* angles_1,
                                      const uint8_t* desc_1, int n1, const float* xs_2, const float* ys_2, const int32_t* octaves_2,
                                      const float* angles_2, const uint8_t* desc_2, int n2, float* prev_matched_xy,
                                      int32_t* matched_2_in_1, int margin, float lowe_ratio, int check_orientation) {
    Grid g;
    build_grid(g, *gp, xs_2, ys_2, n2);
    int num_matches = 0;
    AngleChecker ac;
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
    std::vector<unsigned> matched_dists_2((size_t)n2, OVO_MAX_HAMMING_DIST);
    std::vector<int> matched_1_in_2((size_t)n2, -1);
    for (int idx_1 = 0; idx_1 < n1; ++idx_1) {
        const int level_1 = octaves_1[idx_1];
        if (0 < level_1) continue;
        unsigned best = OVO_MAX_HAMMING_DIST, second = OVO_MAX_HAMMING_DIST;
        int best_idx_2 = -1;
        for_keypoints_in_cell(g, xs_2, ys_2, octaves_2, prev_matched_xy[2 * idx_1], prev_matched_xy[2 * idx_1 + 1], (float)margin,
                              level_1, level_1, [&](int idx_2) {
                                  const unsigned d = distance_32(desc_1 + (size_t)idx_1 * 32, desc_2 + (size_t)idx_2 * 32);
                                  if (matched_dists_2[idx_2] <= d) return;
                                  if (d < best) {
                                      second = best;
                                      best = d;
                                      best_idx_2 = idx_2;
                                  } else if (d < second) {
                                      second = d;
                                  }
                              });
        if (OVO_HAMMING_DIST_THR_LOW < best) continue;
        if ((float)second * lowe_ratio < (float)best) continue;
        const int prev_idx_1 = matched_1_in_2[best_idx_2];
        if (0 <= prev_idx_1) {
            matched_2_in_1[prev_idx_1] = -1;
            --num_matches;
        }
        matched_2_in_1[idx_1] = best_idx_2;
        matched_1_in_2[best_idx_2] = idx_1;
        matched_dists_2[best_idx_2] = best;
        ++num_matches;
        if (check_orientation) ac.append(angles_1[idx_1] - angles_2[best_idx_2], idx_1);
    }
    if (check_orientation) {
        for (int invalid_idx_1 : ac.invalid()) {
            if (0 <= matched_2_in_1[invalid_idx_1]) {
                matched_2_in_1[invalid_idx_1] = -1;
                --num_matches;
            }
        }
    }
    for (int idx_1 = 0; idx_1 < n1; ++idx_1) {
        if (0 <= matched_2_in_1[idx_1]) {
            prev_matched_xy[2 * idx_1] = xs_2[matched_2_in_1[idx_1]];
            prev_matched_xy[2 * idx_1 + 1] = ys_2[matched_2_in_1[idx_1]];
        }
    }
    return num_matches;
}

// M7  bow_tree::match_frame_and_keyframe(keyfrm, frm, matched_lms_in_frm): walk the two BoW feature vectors (node id ->
// keypoint indices, ascending node ids); inside a common node every keyframe keypoint with a live landmark scans the
// frame keypoints of that node that are still unmatched; accept iff best <= THR_LOW and !(ratio * second < best);
// optional orientation histogram. matched_kf_in_frm[frame idx] = keyframe keypoint index (its landmark) or -1.
int ovo_bow_match_frame_and_keyframe(const uint8_t* kf_desc, const float* kf_angles, const uint8_t* kf_valid, int n_kf,
                                     const int32_t* kf_node_ids, const int32_t* kf_node_start, const int32_t* kf_items, int kf_nodes,
                                     const uint8_t* frm_desc, const float* frm_angles, int n_frm, const int32_t* frm_node_ids,
                                     const int32_t* frm_node_start, const int32_t* frm_items, int frm_nodes, float lowe_ratio,
                                     int check_orientation, int32_t* matched_kf_in_frm) {
    (void)n_kf;
    int num_matches = 0;
    AngleChecker ac;
    for (int i = 0; i < n_frm; ++i) matched_kf_in_frm[i] = -1;
    int a = 0, b = 0;
    while (a < kf_nodes && b < frm_nodes) {
        if (kf_node_ids[a] == frm_node_ids[b]) {
            for (int ka = kf_node_start[a]; ka < kf_node_start[a + 1]; ++ka) {
                const int kf_idx = kf_items[ka];
                if (kf_valid && !kf_valid[kf_idx]) continue;
                unsigned best = OVO_MAX_HAMMING_DIST, second = OVO_MAX_HAMMING_DIST;
                int best_frm_idx = -1;
                for (int kb = frm_node_start[b]; kb < frm_node_start[b + 1]; ++kb) {
                    const int frm_idx = frm_items[kb];
                    if (matched_kf_in_frm[frm_idx] >= 0) continue;
                    const unsigned d = distance_32(kf_desc + (size_t)kf_idx * 32, frm_desc + (size_t)frm_idx * 32);
                    if (d < best) {
                        second = best;
                        best = d;
                        best_frm_idx = frm_idx;
                    } else if (d < second) {
                        second = d;
                    }
                }
                if (OVO_HAMMING_DIST_THR_LOW < best) continue;
                if (lowe_ratio * (float)second < (float)best) continue;
                matched_kf_in_frm[best_frm_idx] = kf_idx;
                if (check_orientation) ac.append(kf_angles[kf_idx] - frm_angles[best_frm_idx], best_frm_idx);
                ++num_matches;
            }
            ++a;
            ++b;
        } else if (kf_node_ids[a] < frm_node_ids[b]) {
            a = (int)(std::lower_bound(kf_node_ids + a, kf_node_ids + kf_nodes, frm_node_ids[b]) - kf_node_ids);
        } else {
            b = (int)(std::lower_bound(frm_node_ids + b, frm_node_ids + frm_nodes, kf_node_ids[a]) - frm_node_ids);
        }
    }
    if (check_orientation) {
        for (int invalid_idx : ac.invalid()) {
            matched_kf_in_frm[invalid_idx] = -1;
            --num_matches;
        }
    }
    return num_matches;
}

// landmark::get_min_valid_distance() / get_max_valid_distance() (expected: src/openvslam/data/landmark.cc): the stored range widened by
// 30 %, `return 0.7 * min_valid_dist_;` / `return 1.3 * max_valid_dist_;` -- a double product returned as float. The range gate uses
// these; landmark::predict_scale_level uses the RAW max_valid_dist_. The *_dist_min_max arrays below carry the raw members.
static inline double valid_min(float raw) { return (double)(float)(0.7 * (double)raw); }
static inline double valid_max(float raw) { return (double)(float)(1.3 * (double)raw); }

// camera::perspective / camera::equirectangular ::reproject_to_image (expected: src/openvslam/camera/{perspective,equirectangular}.cc).
// Double precision, one rounding per operation (no FMA contraction: the library is built with -ffp-contract=off).
static bool reproject_to_image(const ovo_camera& cam, const ovo_grid_params& b, const double* P, const double* X, double* reproj,
                               float* x_right) {
    const double pc[3] = {(P[0] * X[0] + P[1] * X[1]) + P[2] * X[2] + P[9], (P[3] * X[0] + P[4] * X[1]) + P[5] * X[2] + P[10],
                          (P[6] * X[0] + P[7] * X[1]) + P[8] * X[2] + P[11]};
    if (cam.model == 0) {
        if (pc[2] <= 0.0) return false;
        const double z_inv = 1.0 / pc[2];
        reproj[0] = cam.fx * pc[0] * z_inv + cam.cx;
        reproj[1] = cam.fy * pc[1] * z_inv + cam.cy;
        *x_right = (float)(reproj[0] - cam.focal_x_baseline * z_inv);
        if (reproj[0] < b.min_x || reproj[0] > b.max_x) return false;
        if (reproj[1] < b.min_y || reproj[1] > b.max_y) return false;
        return true;
    }
    const double norm = std::sqrt((pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2]);
    const double bx = pc[0] / norm, by = pc[1] / norm, bz = pc[2] / norm;
    const double latitude = -ovs_det_asin(by);
    const double longitude = ovs_det_atan2(bx, bz);
    reproj[0] = cam.cols * (0.5 + longitude / (2.0 * M_PI));
    reproj[1] = cam.rows * (0.5 - latitude / M_PI);
    *x_right = -1.0f;
    return true;
}

int ovo_detmath_eval(int fn, const double* a, const double* b, double* out, int n) {
    for (int i = 0; i < n; ++i) {
        switch (fn) {
            case 0: out[i] = (double)ovs_det_logf((float)a[i]); break;
            case 1: out[i] = ovs_det_asin(a[i]); break;
            case 2: out[i] = ovs_det_acos(a[i]); break;
            case 3: out[i] = ovs_det_atan2(a[i], b[i]); break;
            case 4: out[i] = (double)ovs_det_sinf((float)a[i]); break;
            case 5: out[i] = (double)ovs_det_cosf((float)a[i]); break;
            default: return -1;
        }
    }
    return 0;
}

// ovs_det_logf against THIS machine's libm logf over every positive finite float bit pattern in [first, last]: number of mismatches
long long ovo_detmath_logf_vs_libm(uint32_t first, uint32_t last) {
    long long mism = 0;
#pragma omp parallel for reduction(+ : mism) schedule(static)
    for (int64_t u = first; u <= (int64_t)last; ++u) {
        const uint32_t b = (uint32_t)u;
        float x;
        std::memcpy(&x, &b, 4);
        const float mine = ovs_det_logf(x), ref = ::logf(x);
        uint32_t um, ur;
        std::memcpy(&um, &mine, 4);
        std::memcpy(&ur, &ref, 4);
        if (um != ur) ++mism;
    }
    return mism;
}

int ovo_reproject_to_image(const ovo_camera* cam, const ovo_grid_params* bounds, const double* pose_cw, const double* pos_w,
                           double* reproj_xy, float* x_right) {
    return reproject_to_image(*cam, *bounds, pose_cw, pos_w, reproj_xy, x_right) ? 1 : 0;
}

// M4  projection::match_current_and_last_frames(curr_frm, last_frm, margin): every last-frame keypoint with a live, inlier landmark
// is reprojected with the CURRENT pose; the level window follows the motion (non-monocular only: forward if the z-translation
// current->last exceeds the baseline -> [level, top]; backward -> [0, level]; else [level-1, level+1]); keypoints of the current
// frame that already hold an observed landmark are skipped (sequential claim); stereo keypoints must agree on x_right; best
// Hamming only (no ratio test), accept iff best <= THR_HIGH; orientation histogram keyed by the current keypoint.
int ovo_projection_match_current_and_last_frames(const ovo_camera* cam, const ovo_grid_params* gp, const float* xs, const float* ys,
                                                 const int32_t* octaves, const float* angles, const float* stereo_x_right,
                                                 const uint8_t* desc, const uint8_t* occupied, int n_curr, const double* pose_cw_curr,
                                                 const int32_t* last_octaves, const float* last_angles, const double* last_pos_w,
                                                 const uint8_t* last_lm_desc, const uint8_t* last_valid, int n_last,
                                                 const double* pose_cw_last, const float* scale_factors, int num_scale_levels,
                                                 float margin, int check_orientation, int32_t* assigned) {
    Grid g;
    build_grid(g, *gp, xs, ys, n_curr);
    std::vector<uint8_t> occ((size_t)n_curr, 0);
    if (occupied) occ.assign(occupied, occupied + n_curr);
    int num_matches = 0;
    AngleChecker ac;
    const double* Rc = pose_cw_curr;
    const double* tc = pose_cw_curr + 9;
    // trans_wc = -rot_cw^T * trans_cw; trans_lc = rot_lw * trans_wc + trans_lw
    const double twc[3] = {-((Rc[0] * tc[0] + Rc[3] * tc[1]) + Rc[6] * tc[2]), -((Rc[1] * tc[0] + Rc[4] * tc[1]) + Rc[7] * tc[2]),
                           -((Rc[2] * tc[0] + Rc[5] * tc[1]) + Rc[8] * tc[2])};
    const double* Rl = pose_cw_last;
    const double tlc_z = ((Rl[6] * twc[0] + Rl[7] * twc[1]) + Rl[8] * twc[2]) + pose_cw_last[11];
    const bool assume_forward = cam->setup == 0 ? false : tlc_z > cam->true_baseline;
    const bool assume_backward = cam->setup == 0 ? false : -tlc_z > cam->true_baseline;
    std::vector<int> target_of((size_t)n_curr, -1);   // current keypoint -> last index that owns it (for the histogram removal)
    for (int il = 0; il < n_last; ++il) {
        assigned[il] = -1;
        if (last_valid && !last_valid[il]) continue;
        double reproj[2];
        float x_right;
        if (!reproject_to_image(*cam, *gp, pose_cw_curr, last_pos_w + 3 * (size_t)il, reproj, &x_right)) continue;
        const int lvl = last_octaves[il];
        const float r = margin * scale_factors[lvl];
        int minl, maxl;
        if (assume_forward) {
            minl = lvl;
            maxl = num_scale_levels - 1;
        } else if (assume_backward) {
            minl = 0;
            maxl = lvl;
        } else {
            minl = lvl - 1;
            maxl = lvl + 1;
        }
        unsigned best = OVO_MAX_HAMMING_DIST;
        int best_idx = -1;
        for_keypoints_in_cell(g, xs, ys, octaves, (float)reproj[0], (float)reproj[1], r, minl, maxl, [&](int idx) {
            if (occ[idx]) return;
            if (stereo_x_right && stereo_x_right[idx] > 0) {
                const float reproj_error = std::fabs(x_right - stereo_x_right[idx]);
                if (r < reproj_error) return;
            }
            const unsigned d = distance_32(last_lm_desc + (size_t)il * 32, desc + (size_t)idx * 32);
            if (d < best) {
                best = d;
                best_idx = idx;
            }
        });
        if (OVO_HAMMING_DIST_THR_HIGH < best) continue;
        assigned[il] = best_idx;
        occ[best_idx] = 1;
        target_of[best_idx] = il;
        ++num_matches;
        if (check_orientation) ac.append(last_angles[il] - angles[best_idx], best_idx);
    }
    if (check_orientation) {
        for (int invalid_idx : ac.invalid()) {
            assigned[target_of[invalid_idx]] = -1;
            --num_matches;
        }
    }
    return num_matches;
}

// M8  fuse::replace_duplication(keyfrm, landmarks_to_check, margin), candidate search (expected: src/openvslam/match/fuse.{h,cc}):
// per landmark, independently: reproject with the keyframe pose; distance inside [min_valid, max_valid]; viewing angle
// cam_to_lm . mean_normal >= 0.5 |cam_to_lm|; predicted level = clamp(ceil(log(max_valid / dist) / log_scale_factor), 0, L-1)
// (landmark::predict_scale_level, float); ALL grid candidates within margin * scale_factors[pred]; keypoint level in
// [pred-1, pred]; chi-square gate on the reprojection error (5.99146 mono, 7.81473 with a stereo x_right) scaled by
// inv_level_sigma_sq[level]; best Hamming (strict <), accept iff best <= THR_LOW.
int ovo_fuse_replace_duplication(const ovo_camera* cam, const ovo_grid_params* gp, const float* xs, const float* ys,
                                 const int32_t* octaves, const float* stereo_x_right, const uint8_t* desc, int n, const double* pose_cw,
                                 const double* lm_pos_w, const float* lm_dist_min_max, const double* lm_normal, const uint8_t* lm_desc,
                                 const uint8_t* lm_valid, int m, const float* scale_factors, const float* inv_level_sigma_sq,
                                 int num_scale_levels, float log_scale_factor, float margin, int32_t* best_idx_out) {
    Grid g;
    build_grid(g, *gp, xs, ys, n);
    const double* R = pose_cw;
    const double* t = pose_cw + 9;
    // cam_center = -R^T t
    const double cc[3] = {-((R[0] * t[0] + R[3] * t[1]) + R[6] * t[2]), -((R[1] * t[0] + R[4] * t[1]) + R[7] * t[2]),
                          -((R[2] * t[0] + R[5] * t[1]) + R[8] * t[2])};
    int num_fused = 0;
    for (int l = 0; l < m; ++l) {
        best_idx_out[l] = -1;
        if (lm_valid && !lm_valid[l]) continue;
        const double* X = lm_pos_w + 3 * (size_t)l;
        double reproj[2];
        float x_right;
        if (!reproject_to_image(*cam, *gp, pose_cw, X, reproj, &x_right)) continue;
        const double v[3] = {X[0] - cc[0], X[1] - cc[1], X[2] - cc[2]};
        const double dist = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        const float dmin = lm_dist_min_max[2 * l], dmax = lm_dist_min_max[2 * l + 1];
        if (dist < valid_min(dmin) || valid_max(dmax) < dist) continue;
        const double* nrm = lm_normal + 3 * (size_t)l;
        if ((v[0] * nrm[0] + v[1] * nrm[1]) + v[2] * nrm[2] < 0.5 * dist) continue;
        // landmark::predict_scale_level(cam_to_lm_dist, keyfrm)
        const float ratio = dmax / (float)dist;
        int pred = (int)std::ceil(ovs_det_logf(ratio) / log_scale_factor);
        if (pred < 0) pred = 0;
        else if (num_scale_levels <= pred) pred = num_scale_levels - 1;
        const float r = margin * scale_factors[pred];
        unsigned best = OVO_MAX_HAMMING_DIST;
        int best_idx = -1;
        for_keypoints_in_cell(g, xs, ys, octaves, (float)reproj[0], (float)reproj[1], r, -1, -1, [&](int idx) {
            const int level = octaves[idx];
            if (level < pred - 1 || pred < level) return;
            const double ex = reproj[0] - xs[idx], ey = reproj[1] - ys[idx];
            if (stereo_x_right && stereo_x_right[idx] >= 0) {
                const double exr = (double)x_right - (double)stereo_x_right[idx];
                const double e2 = (ex * ex + ey * ey) + exr * exr;
                if (7.81473f < e2 * inv_level_sigma_sq[level]) return;
            } else {
                const double e2 = ex * ex + ey * ey;
                if (5.99146f < e2 * inv_level_sigma_sq[level]) return;
            }
            const unsigned d = distance_32(lm_desc + (size_t)l * 32, desc + (size_t)idx * 32);
            if (d < best) {
                best = d;
                best_idx = idx;
            }
        });
        if (OVO_HAMMING_DIST_THR_LOW < best) continue;
        best_idx_out[l] = best_idx;
        ++num_fused;
    }
    return num_fused;
}

// M7  bow_tree::match_keyframes(keyfrm_1, keyfrm_2, matched_lms_in_keyfrm_1): as match_frame_and_keyframe, but both sides must
// hold a live landmark, and the result is indexed by keyframe-1 keypoints. matched_2_in_1[idx_1] = idx_2 or -1.
int ovo_bow_match_keyframes(const uint8_t* desc_1, const float* angles_1, const uint8_t* valid_1, int n1, const int32_t* node_ids_1,
                            const int32_t* node_start_1, const int32_t* items_1, int nodes_1, const uint8_t* desc_2, const float* angles_2,
                            const uint8_t* valid_2, int n2, const int32_t* node_ids_2, const int32_t* node_start_2, const int32_t* items_2,
                            int nodes_2, float lowe_ratio, int check_orientation, int32_t* matched_2_in_1) {
    int num_matches = 0;
    AngleChecker ac;
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
    std::vector<uint8_t> already_2((size_t)n2, 0);
    int a = 0, b = 0;
    while (a < nodes_1 && b < nodes_2) {
        if (node_ids_1[a] == node_ids_2[b]) {
            for (int ka = node_start_1[a]; ka < node_start_1[a + 1]; ++ka) {
                const int idx_1 = items_1[ka];
                if (valid_1 && !valid_1[idx_1]) continue;
                unsigned best = OVO_MAX_HAMMING_DIST, second = OVO_MAX_HAMMING_DIST;
                int best_idx_2 = -1;
                for (int kb = node_start_2[b]; kb < node_start_2[b + 1]; ++kb) {
                    const int idx_2 = items_2[kb];
                    if (valid_2 && !valid_2[idx_2]) continue;
                    if (already_2[idx_2]) continue;
                    const unsigned d = distance_32(desc_1 + (size_t)idx_1 * 32, desc_2 + (size_t)idx_2 * 32);
                    if (d < best) {
                        second = best;
                        best = d;
                        best_idx_2 = idx_2;
                    } else if (d < second) {
                        second = d;
                    }
                }
                if (OVO_HAMMING_DIST_THR_LOW < best) continue;
                if (lowe_ratio * (float)second < (float)best) continue;
                matched_2_in_1[idx_1] = best_idx_2;
                already_2[best_idx_2] = 1;
                ++num_matches;
                if (check_orientation) ac.append(angles_1[idx_1] - angles_2[best_idx_2], idx_1);
            }
            ++a;
            ++b;
        } else if (node_ids_1[a] < node_ids_2[b]) {
            a = (int)(std::lower_bound(node_ids_1 + a, node_ids_1 + nodes_1, node_ids_2[b]) - node_ids_1);
        } else {
            b = (int)(std::lower_bound(node_ids_2 + b, node_ids_2 + nodes_2, node_ids_1[a]) - node_ids_2);
        }
    }
    if (check_orientation) {
        for (int invalid_idx : ac.invalid()) {
            matched_2_in_1[invalid_idx] = -1;
            --num_matches;
        }
    }
    return num_matches;
}

// M4  projection::match_frame_and_keyframe(curr_frm, keyfrm, already_matched_lms, margin, hamm_dist_thr): every live keyframe
// landmark that is not already matched is reprojected with the CURRENT frame's pose; it must lie inside its valid distance range;
// predicted level from the distance (landmark::predict_scale_level); window margin * scale_factors[pred], levels [pred-1, pred+1];
// current keypoints that already hold ANY landmark are skipped (sequential claim); best Hamming only, accept iff best <=
// hamm_dist_thr; orientation histogram keyed by the current keypoint. assigned[i] = current keypoint or -1.
int ovo_projection_match_frame_and_keyframe(const ovo_camera* cam, const ovo_grid_params* gp, const float* xs, const float* ys,
                                            const int32_t* octaves, const float* angles, const uint8_t* desc, const uint8_t* occupied,
                                            int n_curr, const double* pose_cw_curr, const float* kf_angles, const double* kf_pos_w,
                                            const float* kf_dist_min_max, const uint8_t* kf_lm_desc, const uint8_t* kf_valid, int n_kf,
                                            const float* scale_factors, int num_scale_levels, float log_scale_factor, float margin,
                                            unsigned hamm_dist_thr, int check_orientation, int32_t* assigned) {
    Grid g;
    build_grid(g, *gp, xs, ys, n_curr);
    std::vector<uint8_t> occ((size_t)n_curr, 0);
    if (occupied) occ.assign(occupied, occupied + n_curr);
    const double* R = pose_cw_curr;
    const double* t = pose_cw_curr + 9;
    const double cc[3] = {-((R[0] * t[0] + R[3] * t[1]) + R[6] * t[2]), -((R[1] * t[0] + R[4] * t[1]) + R[7] * t[2]),
                          -((R[2] * t[0] + R[5] * t[1]) + R[8] * t[2])};
    int num_matches = 0;
    AngleChecker ac;
    std::vector<int> owner((size_t)n_curr, -1);
    for (int i = 0; i < n_kf; ++i) {
        assigned[i] = -1;
        if (kf_valid && !kf_valid[i]) continue;
        const double* X = kf_pos_w + 3 * (size_t)i;
        double reproj[2];
        float x_right;
        if (!reproject_to_image(*cam, *gp, pose_cw_curr, X, reproj, &x_right)) continue;
        const double v[3] = {X[0] - cc[0], X[1] - cc[1], X[2] - cc[2]};
        const double dist = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        const float dmin = kf_dist_min_max[2 * i], dmax = kf_dist_min_max[2 * i + 1];
        if (dist < valid_min(dmin) || valid_max(dmax) < dist) continue;
        const float ratio = dmax / (float)dist;
        int pred = (int)std::ceil(ovs_det_logf(ratio) / log_scale_factor);
        if (pred < 0) pred = 0;
        else if (num_scale_levels <= pred) pred = num_scale_levels - 1;
        const float r = margin * scale_factors[pred];
        unsigned best = OVO_MAX_HAMMING_DIST;
        int best_idx = -1;
        for_keypoints_in_cell(g, xs, ys, octaves, (float)reproj[0], (float)reproj[1], r, pred - 1, pred + 1, [&](int idx) {
            if (occ[idx]) return;
            const unsigned d = distance_32(kf_lm_desc + (size_t)i * 32, desc + (size_t)idx * 32);
            if (d < best) {
                best = d;
                best_idx = idx;
            }
        });
        if (hamm_dist_thr < best) continue;
        assigned[i] = best_idx;
        occ[best_idx] = 1;
        owner[best_idx] = i;
        ++num_matches;
        if (check_orientation) ac.append(kf_angles[i] - angles[best_idx], best_idx);
    }
    if (check_orientation) {
        for (int invalid_idx : ac.invalid()) {
            assigned[owner[invalid_idx]] = -1;
            --num_matches;
        }
    }
    return num_matches;
}

// Sim3_cw = [sR | t'] (12 doubles, rows of sR then t') -> rot_cw = sR / s, trans_cw = t' / s with s = |first row of sR|; camera centre
// -rot_cw^T trans_cw (as fuse::detect_duplication / projection::match_by_Sim3_transform decompose the 4x4 Sim3).
static void decompose_sim3(const double* S, double* P, double* cc) {
    const double sc = std::sqrt((S[0] * S[0] + S[1] * S[1]) + S[2] * S[2]);
    for (int i = 0; i < 9; ++i) P[i] = S[i] / sc;
    for (int i = 0; i < 3; ++i) P[9 + i] = S[9 + i] / sc;
    cc[0] = -((P[0] * P[9] + P[3] * P[10]) + P[6] * P[11]);
    cc[1] = -((P[1] * P[9] + P[4] * P[10]) + P[7] * P[11]);
    cc[2] = -((P[2] * P[9] + P[5] * P[10]) + P[8] * P[11]);
}

// M8  fuse::detect_duplication(keyfrm, Sim3_cw, landmarks_to_check, margin, duplicated_lms_in_keyfrm), candidate search (expected:
// src/openvslam/match/fuse.cc): as replace_duplication's search but with the Sim3-corrected pose and WITHOUT the chi-square gate: reproject,
// distance range, viewing angle, predicted level, all grid candidates within margin * scale_factors[pred] at levels [pred-1, pred], best
// Hamming (strict <), accept iff best <= THR_LOW. lm_valid carries "!will_be_erased && not already a landmark of this keyframe". The
// bookkeeping that follows (duplicated_lms_in_keyfrm / add_observation) is host-side graph surgery on best_idx and outside the ABI.
int ovo_fuse_detect_duplication(const ovo_camera* cam, const ovo_grid_params* gp, const float* xs, const float* ys, const int32_t* octaves,
                                const uint8_t* desc, int n, const double* sim3_cw, const double* lm_pos_w, const float* lm_dist_min_max,
                                const double* lm_normal, const uint8_t* lm_desc, const uint8_t* lm_valid, int m, const float* scale_factors,
                                int num_scale_levels, float log_scale_factor, float margin, int32_t* best_idx_out) {
    Grid g;
    build_grid(g, *gp, xs, ys, n);
    double P[12], cc[3];
    decompose_sim3(sim3_cw, P, cc);
    int num = 0;
    for (int l = 0; l < m; ++l) {
        best_idx_out[l] = -1;
        if (lm_valid && !lm_valid[l]) continue;
        const double* X = lm_pos_w + 3 * (size_t)l;
        double reproj[2];
        float x_right;
        if (!reproject_to_image(*cam, *gp, P, X, reproj, &x_right)) continue;
        const double v[3] = {X[0] - cc[0], X[1] - cc[1], X[2] - cc[2]};
        const double dist = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        const float dmin = lm_dist_min_max[2 * l], dmax = lm_dist_min_max[2 * l + 1];
        if (dist < valid_min(dmin) || valid_max(dmax) < dist) continue;
        const double* nrm = lm_normal + 3 * (size_t)l;
        if ((v[0] * nrm[0] + v[1] * nrm[1]) + v[2] * nrm[2] < 0.5 * dist) continue;
        const float ratio = dmax / (float)dist;
        int pred = (int)std::ceil(ovs_det_logf(ratio) / log_scale_factor);
        if (pred < 0) pred = 0;
        else if (num_scale_levels <= pred) pred = num_scale_levels - 1;
        const float r = margin * scale_factors[pred];
        unsigned best = OVO_MAX_HAMMING_DIST;
        int best_idx = -1;
        for_keypoints_in_cell(g, xs, ys, octaves, (float)reproj[0], (float)reproj[1], r, -1, -1, [&](int idx) {
            const int level = octaves[idx];
            if (level < pred - 1 || pred < level) return;
            const unsigned d = distance_32(lm_desc + (size_t)l * 32, desc + (size_t)idx * 32);
            if (d < best) {
                best = d;
                best_idx = idx;
            }
        });
        if (OVO_HAMMING_DIST_THR_LOW < best) continue;
        best_idx_out[l] = best_idx;
        ++num;
    }
    return num;
}

// M4  projection::match_by_Sim3_transform(keyfrm, Sim3_cw, landmarks, matched_lms_in_keyfrm, margin) (expected: src/openvslam/match/
// projection.cc; ORB-SLAM2 SearchByProjection(pKF, Scw, vpPoints, vpMatched, th)): landmarks in order; skipped when erased or already in
// matched_lms_in_keyfrm (lm_valid); Sim3-corrected pose; distance range; viewing angle; predicted level; window margin * sf[pred], levels
// [pred-1, pred]; keypoints that already hold a match are skipped (sequential claim); best Hamming, accept iff best <= THR_LOW; no
// orientation check. assigned[l] = keypoint index or -1.
int ovo_projection_match_by_sim3_transform(const ovo_camera* cam, const ovo_grid_params* gp, const float* xs, const float* ys,
                                           const int32_t* octaves, const uint8_t* desc, const uint8_t* occupied, int n, const double* sim3_cw,
                                           const double* lm_pos_w, const float* lm_dist_min_max, const double* lm_normal, const uint8_t* lm_desc,
                                           const uint8_t* lm_valid, int m, const float* scale_factors, int num_scale_levels,
                                           float log_scale_factor, float margin, int32_t* assigned) {
    Grid g;
    build_grid(g, *gp, xs, ys, n);
    std::vector<uint8_t> occ((size_t)n, 0);
    if (occupied) occ.assign(occupied, occupied + n);
    double P[12], cc[3];
    decompose_sim3(sim3_cw, P, cc);
    int num_matches = 0;
    for (int l = 0; l < m; ++l) {
        assigned[l] = -1;
        if (lm_valid && !lm_valid[l]) continue;
        const double* X = lm_pos_w + 3 * (size_t)l;
        double reproj[2];
        float x_right;
        if (!reproject_to_image(*cam, *gp, P, X, reproj, &x_right)) continue;
        const double v[3] = {X[0] - cc[0], X[1] - cc[1], X[2] - cc[2]};
        const double dist = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        const float dmin = lm_dist_min_max[2 * l], dmax = lm_dist_min_max[2 * l + 1];
        if (dist < valid_min(dmin) || valid_max(dmax) < dist) continue;
        const double* nrm = lm_normal + 3 * (size_t)l;
        if ((v[0] * nrm[0] + v[1] * nrm[1]) + v[2] * nrm[2] < 0.5 * dist) continue;
        const float ratio = dmax / (float)dist;
        int pred = (int)std::ceil(ovs_det_logf(ratio) / log_scale_factor);
        if (pred < 0) pred = 0;
        else if (num_scale_levels <= pred) pred = num_scale_levels - 1;
        const float r = margin * scale_factors[pred];
        unsigned best = OVO_MAX_HAMMING_DIST;
        int best_idx = -1;
        for_keypoints_in_cell(g, xs, ys, octaves, (float)reproj[0], (float)reproj[1], r, pred - 1, pred, [&](int idx) {
            if (occ[idx]) return;
            const unsigned d = distance_32(lm_desc + (size_t)l * 32, desc + (size_t)idx * 32);
            if (d < best) {
                best = d;
                best_idx = idx;
            }
        });
        if (OVO_HAMMING_DIST_THR_LOW < best) continue;
        assigned[l] = best_idx;
        occ[best_idx] = 1;
        ++num_matches;
    }
    return num_matches;
}

// one direction of match_keyframes_mutually: landmarks of keyframe A into keyframe B through pos_a = R_aw X + t_aw, pos_b = S_ba pos_a
static void mutual_pass(const ovo_camera& cam_b, const ovo_grid_params& gp_b, const float* xs, const float* ys, const int32_t* octaves,
                        const uint8_t* desc, int n_b, const double* pose_cw_a, const double* S_ba, const double* lm_pos_w,
                        const float* lm_dist_min_max, const uint8_t* lm_desc, const uint8_t* lm_valid, int n_a, const float* scale_factors,
                        int num_scale_levels, float log_scale_factor, float margin, std::vector<int>& out) {
    Grid g;
    build_grid(g, gp_b, xs, ys, n_b);
    out.assign((size_t)n_a, -1);
    const double* Pa = pose_cw_a;
    for (int i = 0; i < n_a; ++i) {
        if (lm_valid && !lm_valid[i]) continue;
        const double* X = lm_pos_w + 3 * (size_t)i;
        const double pa[3] = {(Pa[0] * X[0] + Pa[1] * X[1]) + Pa[2] * X[2] + Pa[9], (Pa[3] * X[0] + Pa[4] * X[1]) + Pa[5] * X[2] + Pa[10],
                              (Pa[6] * X[0] + Pa[7] * X[1]) + Pa[8] * X[2] + Pa[11]};
        double reproj[2];
        float x_right;
        if (!reproject_to_image(cam_b, gp_b, S_ba, pa, reproj, &x_right)) continue;
        const double pb[3] = {(S_ba[0] * pa[0] + S_ba[1] * pa[1]) + S_ba[2] * pa[2] + S_ba[9],
                              (S_ba[3] * pa[0] + S_ba[4] * pa[1]) + S_ba[5] * pa[2] + S_ba[10],
                              (S_ba[6] * pa[0] + S_ba[7] * pa[1]) + S_ba[8] * pa[2] + S_ba[11]};
        const double dist = std::sqrt((pb[0] * pb[0] + pb[1] * pb[1]) + pb[2] * pb[2]);
        const float dmin = lm_dist_min_max[2 * i], dmax = lm_dist_min_max[2 * i + 1];
        if (dist < valid_min(dmin) || valid_max(dmax) < dist) continue;
        const float ratio = dmax / (float)dist;
        int pred = (int)std::ceil(ovs_det_logf(ratio) / log_scale_factor);
        if (pred < 0) pred = 0;
        else if (num_scale_levels <= pred) pred = num_scale_levels - 1;
        const float r = margin * scale_factors[pred];
        unsigned best = OVO_MAX_HAMMING_DIST;
        int best_idx = -1;
        for_keypoints_in_cell(g, xs, ys, octaves, (float)reproj[0], (float)reproj[1], r, -1, -1, [&](int idx) {
            const int level = octaves[idx];
            if (level < pred - 1 || pred < level) return;
            const unsigned d = distance_32(lm_desc + (size_t)i * 32, desc + (size_t)idx * 32);
            if (d < best) {
                best = d;
                best_idx = idx;
            }
        });
        if (best <= OVO_HAMMING_DIST_THR_HIGH) out[(size_t)i] = best_idx;
    }
}

// M4  projection::match_keyframes_mutually(keyfrm_1, keyfrm_2, matched_lms_in_keyfrm_1, s_12, rot_12, trans_12, margin) (expected:
// src/openvslam/match/projection.cc; ORB-SLAM2 SearchBySim3): every unmatched live landmark of keyframe 1 is carried into keyframe 2
// with Sim3_21 = [R_12^T / s_12 | -(R_12^T / s_12) t_12] and vice versa with Sim3_12 = [s_12 R_12 | t_12]; each side independently takes the
// best Hamming candidate (no claim) in the window margin * sf[pred], levels [pred-1, pred], accept iff best <= THR_HIGH; a pair survives
// only if both directions agree. lm_valid_k: keypoint k has a live landmark that is not already matched (for side 2: whose keypoint is not
// the partner of an already matched landmark). matched_2_in_1[idx_1] = idx_2 or -1.
int ovo_projection_match_keyframes_mutually(const ovo_camera* cam_1, const ovo_grid_params* gp_1, const float* xs_1, const float* ys_1,
                                            const int32_t* octaves_1, const uint8_t* desc_1, int n1, const double* pose_cw_1,
                                            const double* lm_pos_w_1, const float* lm_dist_1, const uint8_t* lm_desc_1, const uint8_t* lm_valid_1,
                                            const ovo_camera* cam_2, const ovo_grid_params* gp_2, const float* xs_2, const float* ys_2,
                                            const int32_t* octaves_2, const uint8_t* desc_2, int n2, const double* pose_cw_2,
                                            const double* lm_pos_w_2, const float* lm_dist_2, const uint8_t* lm_desc_2, const uint8_t* lm_valid_2,
                                            double s_12, const double* rot_12, const double* trans_12, const float* scale_factors,
                                            int num_scale_levels, float log_scale_factor, float margin, int32_t* matched_2_in_1) {
    double S12[12], S21[12];
    for (int i = 0; i < 9; ++i) S12[i] = s_12 * rot_12[i];
    for (int i = 0; i < 3; ++i) S12[9 + i] = trans_12[i];
    const double inv_s = 1.0 / s_12;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S21[3 * r + c] = inv_s * rot_12[3 * c + r];
    for (int r = 0; r < 3; ++r) S21[9 + r] = -((S21[3 * r] * trans_12[0] + S21[3 * r + 1] * trans_12[1]) + S21[3 * r + 2] * trans_12[2]);
    std::vector<int> m21, m12;
    mutual_pass(*cam_2, *gp_2, xs_2, ys_2, octaves_2, desc_2, n2, pose_cw_1, S21, lm_pos_w_1, lm_dist_1, lm_desc_1, lm_valid_1, n1, scale_factors,
                num_scale_levels, log_scale_factor, margin, m21);
    mutual_pass(*cam_1, *gp_1, xs_1, ys_1, octaves_1, desc_1, n1, pose_cw_2, S12, lm_pos_w_2, lm_dist_2, lm_desc_2, lm_valid_2, n2, scale_factors,
                num_scale_levels, log_scale_factor, margin, m12);
    int num = 0;
    for (int i = 0; i < n1; ++i) {
        matched_2_in_1[i] = -1;
        const int j = m21[(size_t)i];
        if (j < 0) continue;
        if (m12[(size_t)j] == i) {
            matched_2_in_1[i] = j;
            ++num;
        }
    }
    return num;
}

// M2  robust::match_for_triangulation(keyfrm_1, keyfrm_2, E_12, matched_idx_pairs) + robust::check_epipolar_constraint
// (expected: src/openvslam/match/robust.cc): common BoW nodes; keyframe-1 keypoints WITHOUT a landmark against keyframe-2
// keypoints without a landmark that no earlier keypoint took; a candidate needs d <= THR_LOW and d <= the best so far (a later
// equal distance replaces an earlier one), must not sit within 3 degrees of the epipole when neither side is a stereo keypoint,
// and must satisfy the epipolar constraint: angle between bearing_1 and the epipolar plane E_12 * bearing_2 below
// 0.2 deg * scale_factors[octave_1]. Orientation histogram keyed by idx_1. matched_2_in_1[idx_1] = idx_2 or -1.
int ovo_robust_match_for_triangulation(const uint8_t* desc_1, const float* angles_1, const int32_t* octaves_1, const uint8_t* has_lm_1,
                                       const float* x_right_1, const double* bearings_1, int n1, const int32_t* node_ids_1,
                                       const int32_t* node_start_1, const int32_t* items_1, int nodes_1, const uint8_t* desc_2,
                                       const float* angles_2, const uint8_t* has_lm_2, const float* x_right_2, const double* bearings_2,
                                       int n2, const int32_t* node_ids_2, const int32_t* node_start_2, const int32_t* items_2, int nodes_2,
                                       const double* E_12, const double* epipole_in_2, const float* scale_factors, int check_orientation,
                                       int32_t* matched_2_in_1) {
    int num_matches = 0;
    AngleChecker ac;
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
    std::vector<uint8_t> already_2((size_t)n2, 0);
    const double kPi = 3.14159265358979323846;
    int a = 0, b = 0;
    while (a < nodes_1 && b < nodes_2) {
        if (node_ids_1[a] == node_ids_2[b]) {
            for (int ka = node_start_1[a]; ka < node_start_1[a + 1]; ++ka) {
                const int idx_1 = items_1[ka];
                if (has_lm_1 && has_lm_1[idx_1]) continue;   // only keypoints without a 3D point are triangulated
                const bool is_stereo_1 = x_right_1 && 0 <= x_right_1[idx_1];
                const double* b1 = bearings_1 + 3 * (size_t)idx_1;
                unsigned best = OVO_HAMMING_DIST_THR_LOW;
                int best_idx_2 = -1;
                for (int kb = node_start_2[b]; kb < node_start_2[b + 1]; ++kb) {
                    const int idx_2 = items_2[kb];
                    if (has_lm_2 && has_lm_2[idx_2]) continue;
                    if (already_2[idx_2]) continue;
                    const bool is_stereo_2 = x_right_2 && 0 <= x_right_2[idx_2];
                    const double* b2 = bearings_2 + 3 * (size_t)idx_2;
                    const unsigned d = distance_32(desc_1 + (size_t)idx_1 * 32, desc_2 + (size_t)idx_2 * 32);
                    if (OVO_HAMMING_DIST_THR_LOW < d || best < d) continue;
                    if (!is_stereo_1 && !is_stereo_2) {
                        const double cos_dist = (epipole_in_2[0] * b2[0] + epipole_in_2[1] * b2[1]) + epipole_in_2[2] * b2[2];
                        if (0.99862953475 < cos_dist) continue;
                    }
                    // check_epipolar_constraint
                    const double ep[3] = {(E_12[0] * b2[0] + E_12[1] * b2[1]) + E_12[2] * b2[2], (E_12[3] * b2[0] + E_12[4] * b2[1]) + E_12[5] * b2[2],
                                          (E_12[6] * b2[0] + E_12[7] * b2[1]) + E_12[8] * b2[2]};
                    const double nrm = std::sqrt((ep[0] * ep[0] + ep[1] * ep[1]) + ep[2] * ep[2]);
                    const double cos_residual = ((ep[0] * b1[0] + ep[1] * b1[1]) + ep[2] * b1[2]) / nrm;
                    const double residual_rad = kPi / 2.0 - std::fabs(ovs_det_acos(cos_residual));
                    const double residual_rad_thr = 0.2 * kPi / 180.0;
                    if (!(residual_rad < residual_rad_thr * scale_factors[octaves_1[idx_1]])) continue;
                    best_idx_2 = idx_2;
                    best = d;
                }
                if (best_idx_2 < 0) continue;
                already_2[best_idx_2] = 1;
                matched_2_in_1[idx_1] = best_idx_2;
                ++num_matches;
                if (check_orientation) ac.append(angles_1[idx_1] - angles_2[best_idx_2], idx_1);
            }
            ++a;
            ++b;
        } else if (node_ids_1[a] < node_ids_2[b]) {
            a = (int)(std::lower_bound(node_ids_1 + a, node_ids_1 + nodes_1, node_ids_2[b]) - node_ids_1);
        } else {
            b = (int)(std::lower_bound(node_ids_2 + b, node_ids_2 + nodes_2, node_ids_1[a]) - node_ids_2);
        }
    }
    if (check_orientation) {
        for (int invalid_idx : ac.invalid()) {
            matched_2_in_1[invalid_idx] = -1;
            --num_matches;
        }
    }
    return num_matches;
}

// M6  stereo::compute(stereo_x_right, depths). Keypoints are cv::KeyPoint records (level-0 coordinates, octave); the two
// pyramids are the extractors' image_pyramid_ (unblurred). Steps as upstream / ORB-SLAM2 ComputeStereoMatches:
//   rows: right keypoint i is a candidate for every image row in [floor(y - 2 s_i), ceil(y + 2 s_i)], s_i = scale_factors[octave];
//   per left keypoint: candidates of row (int)y_left with |octave difference| <= 1 and x_right in [x_left - max_disp, x_left],
//     max_disp = focal_x_baseline / true_baseline; best Hamming with strict `<`, must be < (THR_HIGH + THR_LOW) / 2;
//   sub-pixel: on the left keypoint's pyramid level, 11x11 windows (centre value subtracted from each window), L1 distance
//     for the 11 shifts -5..+5 of the right window; reject a best shift at either end; parabola through the three distances
//     around the minimum, |delta| <= 1; x_right = scale * (x_r_scaled + best_shift + delta); disparity in [0, max_disp),
//     a non-positive disparity becomes 0.01; depth = focal_x_baseline / disparity;
//   finally matches whose L1 distance exceeds 2 x the median accepted distance are dropped (ORACLE_SPEC rule 20: upstream's
//   factor is recalled as 2.0 with a strict comparison; ORB-SLAM2 used 1.5 * 1.4 = 2.1).
// variant (ORACLE_SPEC rule 20's two L-tagged choices as run-time switches): bit 0 = outlier factor 2.1 (ORB-SLAM2's 1.5f * 1.4f) instead of 2.0,
// bit 1 = the sub-pixel parabola evaluated in double and rounded to float once instead of float arithmetic
int ovo_stereo_compute_v(const uint8_t* const* pyr_left, const uint8_t* const* pyr_right, const int32_t* level_rows,
                         const int32_t* level_cols, const size_t* stride_left, const size_t* stride_right, int num_levels,
                         const ovo_keypoint* kps_left, const uint8_t* desc_left, int n_left, const ovo_keypoint* kps_right,
                         const uint8_t* desc_right, int n_right, const float* scale_factors, const float* inv_scale_factors,
                         float focal_x_baseline, float true_baseline, float* stereo_x_right, float* depths, int variant) {
    (void)num_levels;
    const int rows0 = level_rows[0];
    for (int i = 0; i < n_left; ++i) stereo_x_right[i] = depths[i] = -1.0f;
    // get_right_keypoint_indices_in_each_row(margin = 2.0)
    std::vector<std::vector<int>> in_row((size_t)rows0);
    for (int i = 0; i < n_right; ++i) {
        const float y = kps_right[i].y;
        const float r = 2.0f * scale_factors[kps_right[i].octave];
        const int max_r = cv_ceil(y + r), min_r = cv_floor(y - r);
        for (int row = min_r; row <= max_r; ++row)
            if (0 <= row && row < rows0) in_row[row].push_back(i);   // upstream relies on the 19-px extraction border instead of clamping
    }
    const float min_disp = 0.0f;
    const float max_disp = focal_x_baseline / true_baseline;
    const unsigned hamm_dist_thr = (OVO_HAMMING_DIST_THR_HIGH + OVO_HAMMING_DIST_THR_LOW) / 2;
    std::vector<std::pair<int, int>> correlation_and_idx_left;
    for (int il = 0; il < n_left; ++il) {
        const ovo_keypoint& kl = kps_left[il];
        const int level_l = kl.octave;
        const float y_left = kl.y, x_left = kl.x;
        const int row = (int)y_left;
        if (row < 0 || row >= rows0) continue;
        const std::vector<int>& cands = in_row[row];
        if (cands.empty()) continue;
        const float min_x_right = x_left - max_disp, max_x_right = x_left - min_disp;
        if (max_x_right < 0) continue;
        // find_closest_keypoints_in_stereo
        unsigned best_dist = hamm_dist_thr;
        int best_idx_right = 0;
        for (int ir : cands) {
            const ovo_keypoint& kr = kps_right[ir];
            if (kr.octave < level_l - 1 || kr.octave > level_l + 1) continue;
            if (kr.x < min_x_right || max_x_right < kr.x) continue;
            const unsigned d = distance_32(desc_left + (size_t)il * 32, desc_right + (size_t)ir * 32);
            if (d < best_dist) {
                best_idx_right = ir;
                best_dist = d;
            }
        }
        if (hamm_dist_thr <= best_dist) continue;
        // compute_subpixel_disparity
        const ovo_keypoint& kr = kps_right[best_idx_right];
        const float isf = inv_scale_factors[level_l];
        const int sxl = cv_round(kl.x * isf), syl = cv_round(kl.y * isf), sxr = cv_round(kr.x * isf);
        constexpr int w = 5, Lr = 5;
        const int lc = level_cols[level_l], lr = level_rows[level_l];
        const int ini_x = sxr - Lr - w, end_x = sxr + Lr + w + 1;
        if (ini_x < 0 || lc <= end_x) continue;
        if (syl - w < 0 || lr <= syl + w || sxl - w < 0 || lc <= sxl + w) continue;   // (cannot trigger for extractor output)
        const uint8_t* IL = pyr_left[level_l];
        const uint8_t* IR = pyr_right[level_l];
        const size_t sl = stride_left[level_l], sr = stride_right[level_l];
        const int cl = IL[(size_t)syl * sl + sxl];
        int best_correlation = INT_MAX, best_offset = 0;
        float correlations[2 * Lr + 1];
        for (int offset = -Lr; offset <= Lr; ++offset) {
            const int cr = IR[(size_t)syl * sr + sxr + offset];
            int sad = 0;   // cv::norm(L1) of two float patches holding integers: exact
            for (int dy = -w; dy <= w; ++dy)
                for (int dx = -w; dx <= w; ++dx) {
                    const int a = (int)IL[(size_t)(syl + dy) * sl + sxl + dx] - cl;
                    const int b = (int)IR[(size_t)(syl + dy) * sr + sxr + offset + dx] - cr;
                    sad += std::abs(a - b);
                }
            const float correlation = (float)sad;
            if (correlation < (float)best_correlation) {
                best_correlation = (int)correlation;
                best_offset = offset;
            }
            correlations[Lr + offset] = correlation;
        }
        if (best_offset == -Lr || best_offset == Lr) continue;
        const float c1 = correlations[Lr + best_offset - 1], c2 = correlations[Lr + best_offset], c3 = correlations[Lr + best_offset + 1];
        const float delta = (variant & 2) ? (float)(((double)c1 - (double)c3) / (2.0 * (((double)c1 + (double)c3) - 2.0 * (double)c2)))
                                          : (c1 - c3) / (2.0f * (c1 + c3 - 2.0f * c2));
        if (delta < -1.0f || 1.0f < delta) continue;   // the denominator is > 0: c1 > c2 (first strict minimum) and c3 >= c2
        float best_x_right = scale_factors[level_l] * ((float)sxr + (float)best_offset + delta);
        float disp = x_left - best_x_right;
        if (disp < min_disp || max_disp <= disp) continue;
        if (disp <= 0.0f) {
            disp = 0.01f;
            best_x_right = x_left - 0.01f;
        }
        depths[il] = focal_x_baseline / disp;
        stereo_x_right[il] = best_x_right;
        correlation_and_idx_left.emplace_back(best_correlation, il);
    }
    std::sort(correlation_and_idx_left.begin(), correlation_and_idx_left.end());
    int n_ok = (int)correlation_and_idx_left.size();
    if (!correlation_and_idx_left.empty()) {
        const float median = (float)correlation_and_idx_left[correlation_and_idx_left.size() / 2].first;
        const float thr = ((variant & 1) ? 2.1f : 2.0f) * median;
        for (const auto& ci : correlation_and_idx_left) {
            if (thr < (float)ci.first) {
                stereo_x_right[ci.second] = -1.0f;
                depths[ci.second] = -1.0f;
                --n_ok;
            }
        }
    }
    return n_ok;
}

int ovo_stereo_compute(const uint8_t* const* pyr_left, const uint8_t* const* pyr_right, const int32_t* level_rows,
                       const int32_t* level_cols, const size_t* stride_left, const size_t* stride_right, int num_levels,
                       const ovo_keypoint* kps_left, const uint8_t* desc_left, int n_left, const ovo_keypoint* kps_right,
                       const uint8_t* desc_right, int n_right, const float* scale_factors, const float* inv_scale_factors,
                       float focal_x_baseline, float true_baseline, float* stereo_x_right, float* depths) {
    return ovo_stereo_compute_v(pyr_left, pyr_right, level_rows, level_cols, stride_left, stride_right, num_levels, kps_left, desc_left, n_left, kps_right,
                                desc_right, n_right, scale_factors, inv_scale_factors, focal_x_baseline, true_baseline, stereo_x_right, depths, 0);
}

// SURVEY 8(f) #4  DBoW2 TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (third-party, absent: restated from the
// published DBoW2 algorithm, ORACLE_SPEC rule 29): from the root, at each level the child with the smallest Hamming distance, first
// minimum in child order (strict `<`); nid = the node reached at level L - levelsup (the root if that is <= 0); stop at a leaf.
int ovo_bow_transform(int n_nodes, const int32_t* child_start, const int32_t* children, const uint8_t* node_desc, const double* node_weight,
                      const int32_t* node_word_id, int depth, const uint8_t* desc, int n, int levelsup, int32_t* word_id, double* weight,
                      int32_t* node_id) {
    (void)n_nodes;
    const int nid_level = depth - levelsup;
    for (int f = 0; f < n; ++f) {
        const uint8_t* q = desc + (size_t)32 * f;
        int final_id = 0, current_level = 0, nid = 0;
        while (child_start[final_id] != child_start[final_id + 1]) {
            ++current_level;
            const int c0 = child_start[final_id], c1 = child_start[final_id + 1];
            int best_id = children[c0];
            unsigned best_d = distance_32(q, node_desc + (size_t)32 * best_id);
            for (int i = c0 + 1; i < c1; ++i) {
                const int id = children[i];
                const unsigned d = distance_32(q, node_desc + (size_t)32 * id);
                if (d < best_d) {
                    best_d = d;
                    best_id = id;
                }
            }
            final_id = best_id;
            if (current_level == nid_level) nid = final_id;
        }
        word_id[f] = node_word_id[final_id];
        weight[f] = node_weight[final_id];
        node_id[f] = nid_level <= 0 ? 0 : nid;
    }
    return 0;
}

}   // extern "C"
