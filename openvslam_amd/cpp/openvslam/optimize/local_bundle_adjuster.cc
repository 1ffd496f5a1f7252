// optimize::local_bundle_adjuster::optimize over the C ABI. Replaces that function's body in
// src/openvslam/optimize/local_bundle_adjuster.cc: steps 1 (collect the local map), 7 (collect outliers) and 8 (write back) are
// upstream's host code restated; steps 2-6 (build the g2o graph, optimise, reject outliers, optimise again) are one call.
#include "local_bundle_adjuster.h"

#include <ovslam_hip.h>

#include "../util/device_policy.h"

#include <cmath>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace openvslam {
namespace optimize {

namespace {

// util::converter::to_g2o_SE3: rotation matrix -> unit quaternion (x, y, z, w) exactly as Eigen::Quaterniond(Matrix3d) does (the
// trace / largest-diagonal branches), then g2o::SE3Quat's normalisation with w >= 0
void pose_to_se3quat(const Mat44_t& T, double* p7) {
    double q[4];
    const double tr = (T(0, 0) + T(1, 1)) + T(2, 2);
    if (tr > 0.0) {
        double s = std::sqrt(tr + 1.0);
        q[3] = 0.5 * s;
        s = 0.5 / s;
        q[0] = (T(2, 1) - T(1, 2)) * s;
        q[1] = (T(0, 2) - T(2, 0)) * s;
        q[2] = (T(1, 0) - T(0, 1)) * s;
    } else {
        int i = 0;
        if (T(1, 1) > T(0, 0)) i = 1;
        if (T(2, 2) > T(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(T(i, i) - T(j, j) - T(k, k) + 1.0);
        q[i] = 0.5 * s;
        s = 0.5 / s;
        q[3] = (T(k, j) - T(j, k)) * s;
        q[j] = (T(j, i) + T(i, j)) * s;
        q[k] = (T(k, i) + T(i, k)) * s;
    }
    if (q[3] < 0.0)
        for (double& v : q) v = -v;
    const double n = std::sqrt((q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]));
    for (int a = 0; a < 3; ++a) p7[a] = T(a, 3);
    for (int a = 0; a < 4; ++a) p7[3 + a] = q[a] / n;
}

// shot_vertex::estimate() -> Mat44 (Eigen::Quaterniond::toRotationMatrix)
Mat44_t se3quat_to_pose(const double* p7) {
    const double x = p7[3], y = p7[4], z = p7[5], w = p7[6];
    Mat44_t T;
    T(0, 0) = 1 - 2 * (y * y + z * z);
    T(0, 1) = 2 * (x * y - z * w);
    T(0, 2) = 2 * (x * z + y * w);
    T(1, 0) = 2 * (x * y + z * w);
    T(1, 1) = 1 - 2 * (x * x + z * z);
    T(1, 2) = 2 * (y * z - x * w);
    T(2, 0) = 2 * (x * z - y * w);
    T(2, 1) = 2 * (y * z + x * w);
    T(2, 2) = 1 - 2 * (x * x + y * y);
    for (int a = 0; a < 3; ++a) T(a, 3) = p7[a];
    return T;
}

}   // namespace

void local_bundle_adjuster::optimize(data::keyframe* curr_keyfrm, bool* const force_stop_flag) const {
    // 1. aggregate the local and fixed keyframes, and the local landmarks

    // local keyframes: the current keyframe and its covisibilities
    std::unordered_map<unsigned int, data::keyframe*> local_keyfrms;
    local_keyfrms[curr_keyfrm->id_] = curr_keyfrm;
    const auto curr_covisibilities = curr_keyfrm->graph_node_->get_covisibilities();
    for (auto local_keyfrm : curr_covisibilities) {
        if (!local_keyfrm) continue;
        if (local_keyfrm->will_be_erased()) continue;
        local_keyfrms[local_keyfrm->id_] = local_keyfrm;
    }

    // local landmarks: everything the local keyframes observe
    std::unordered_map<unsigned int, data::landmark*> local_lms;
    for (auto local_keyfrm : local_keyfrms) {
        const auto landmarks = local_keyfrm.second->get_landmarks();
        for (auto local_lm : landmarks) {
            if (!local_lm) continue;
            if (local_lm->will_be_erased()) continue;
            if (local_lms.count(local_lm->id_)) continue;   // avoid duplication
            local_lms[local_lm->id_] = local_lm;
        }
    }

    // fixed keyframes: keyframes which observe local landmarks but are not local keyframes
    std::unordered_map<unsigned int, data::keyframe*> fixed_keyfrms;
    for (auto local_lm : local_lms) {
        const auto observations = local_lm.second->get_observations();
        for (auto& obs : observations) {
            auto fixed_keyfrm = obs.first;
            if (!fixed_keyfrm) continue;
            if (fixed_keyfrm->will_be_erased()) continue;
            if (local_keyfrms.count(fixed_keyfrm->id_)) continue;
            if (fixed_keyfrms.count(fixed_keyfrm->id_)) continue;
            fixed_keyfrms[fixed_keyfrm->id_] = fixed_keyfrm;
        }
    }

    // 2.-4. flatten what upstream hands to g2o: shot vertices (local ones free unless id 0, fixed ones fixed), landmark vertices,
    // one reprojection edge per observation -- in upstream's insertion order
    std::vector<data::keyframe*> keyfrms;
    std::unordered_map<data::keyframe*, int32_t> pose_index;
    std::vector<double> poses;
    std::vector<uint8_t> pose_fixed;
    const camera::base* camera = curr_keyfrm->camera_;
    auto add_keyfrm = [&](data::keyframe* keyfrm, const bool is_constant) {
        // upstream's reproj_edge_wrapper switches on keyfrm->camera_->model_type_: Perspective, Fisheye and RadialDivision use the perspective
        // reprojection edges (undistorted keypoints, the model's fx, fy, cx, cy), Equirectangular its own (ORACLE_SPEC rule 31)
        if (keyfrm->camera_ != camera &&
            (keyfrm->camera_->model_type_ != camera->model_type_ || keyfrm->camera_->fx_ != camera->fx_ || keyfrm->camera_->fy_ != camera->fy_ ||
             keyfrm->camera_->cx_ != camera->cx_ || keyfrm->camera_->cy_ != camera->cy_ || keyfrm->camera_->cols_ != camera->cols_ ||
             keyfrm->camera_->rows_ != camera->rows_))
            throw std::runtime_error("local_bundle_adjuster: all keyframes of the local map must share one camera");
        pose_index[keyfrm] = (int32_t)keyfrms.size();
        keyfrms.push_back(keyfrm);
        poses.resize(poses.size() + 7);
        pose_to_se3quat(keyfrm->get_cam_pose(), &poses[poses.size() - 7]);
        pose_fixed.push_back(is_constant ? 1 : 0);
    };
    for (auto& id_local_keyfrm_pair : local_keyfrms) add_keyfrm(id_local_keyfrm_pair.second, id_local_keyfrm_pair.second->id_ == 0);
    for (auto& id_fixed_keyfrm_pair : fixed_keyfrms) add_keyfrm(id_fixed_keyfrm_pair.second, true);

    struct edge_ref {   // reproj_edge_wrapper: who observes what, and which of the two edge arrays holds it
        data::keyframe* shot_;
        data::landmark* lm_;
        bool is_monocular_;
        size_t slot_;
    };
    std::vector<edge_ref> reproj_edge_wraps;
    std::vector<data::landmark*> lms;
    std::vector<double> points;
    std::vector<ovs_ba_edge> mono;
    std::vector<ovs_ba_edge_stereo> stereo;
    lms.reserve(local_lms.size());
    for (auto& id_local_lm_pair : local_lms) {
        auto local_lm = id_local_lm_pair.second;
        const int32_t point_idx = (int32_t)lms.size();
        lms.push_back(local_lm);
        const Vec3_t pos_w = local_lm->get_pos_in_world();
        for (int a = 0; a < 3; ++a) points.push_back(pos_w(a));
        const auto observations = local_lm->get_observations();
        for (const auto& obs : observations) {
            auto keyfrm = obs.first;
            auto idx = obs.second;
            if (!keyfrm) continue;
            if (keyfrm->will_be_erased()) continue;
            const auto pit = pose_index.find(keyfrm);
            if (pit == pose_index.end()) continue;   // cannot happen: every observer is local or fixed
            const auto& undist_keypt = keyfrm->undist_keypts_.at(idx);
            const float x_right = keyfrm->stereo_x_right_.empty() ? -1.0f : keyfrm->stereo_x_right_.at(idx);
            const float inv_sigma_sq = keyfrm->inv_level_sigma_sq_.at((size_t)undist_keypt.octave);
            const bool is_monocular = x_right < 0 || camera->model_type_ == camera::model_type_t::Equirectangular;
            if (is_monocular) {
                ovs_ba_edge e;
                e.pose_idx = pit->second;
                e.point_idx = point_idx;
                e.obs_x = undist_keypt.pt.x;
                e.obs_y = undist_keypt.pt.y;
                e.inv_sigma_sq = inv_sigma_sq;
                reproj_edge_wraps.push_back({keyfrm, local_lm, true, mono.size()});
                mono.push_back(e);
            } else {
                ovs_ba_edge_stereo e;
                e.pose_idx = pit->second;
                e.point_idx = point_idx;
                e.obs_x = undist_keypt.pt.x;
                e.obs_y = undist_keypt.pt.y;
                e.obs_x_right = x_right;
                e.inv_sigma_sq = inv_sigma_sq;
                reproj_edge_wraps.push_back({keyfrm, local_lm, false, stereo.size()});
                stereo.push_back(e);
            }
        }
    }

    // 5. + 6. both optimisation rounds and the outlier test in between (upstream returns before optimising if the flag is already set)
    if (force_stop_flag && *force_stop_flag) return;
    if (lms.empty() || reproj_edge_wraps.empty()) return;
    const ovs_ba_cam cam = {camera->fx_, camera->fy_, camera->cx_, camera->cy_};
    std::vector<uint8_t> mono_outlier(mono.size() + 1), stereo_outlier(stereo.size() + 1);
    static_assert(sizeof(bool) == 1, "force_stop_flag is polled as a byte");
    // failure policy (util/device_policy.h): one retry from the same start state, then the local map is left exactly as it was -- for the
    // mapping module that is a local BA that was aborted before its first iteration
    const std::vector<double> poses_in = poses, points_in = points;
    if (!util::run_guarded(
            "ovs_local_ba_optimize",
            [&] {
                poses = poses_in;     // in / out arguments
                points = points_in;
                return camera->model_type_ == camera::model_type_t::Equirectangular
                           ? ovs_local_ba_optimize_equirect(0, poses.data(), pose_fixed.data(), (int32_t)keyfrms.size(), points.data(),
                                                            (int32_t)lms.size(), mono.empty() ? nullptr : mono.data(), (int32_t)mono.size(),
                                                            (int32_t)camera->cols_, (int32_t)camera->rows_, (int32_t)num_first_iter_,
                                                            (int32_t)num_second_iter_, reinterpret_cast<const volatile uint8_t*>(force_stop_flag),
                                                            mono_outlier.data(), nullptr)
                           : ovs_local_ba_optimize(0, poses.data(), pose_fixed.data(), (int32_t)keyfrms.size(), points.data(), (int32_t)lms.size(),
                                                   mono.empty() ? nullptr : mono.data(), (int32_t)mono.size(),
                                                   stereo.empty() ? nullptr : stereo.data(), (int32_t)stereo.size(), &cam,
                                                   camera->focal_x_baseline_, (int32_t)camera->setup_type_, (int32_t)num_first_iter_,
                                                   (int32_t)num_second_iter_, reinterpret_cast<const volatile uint8_t*>(force_stop_flag),
                                                   mono_outlier.data(), stereo_outlier.data(), nullptr);
            },
            [] {}))
        return;

    // 7. count the outlier observations
    std::vector<std::pair<data::keyframe*, data::landmark*>> outlier_observations;
    outlier_observations.reserve(reproj_edge_wraps.size());
    for (auto& reproj_edge_wrap : reproj_edge_wraps) {
        auto local_lm = reproj_edge_wrap.lm_;
        if (local_lm->will_be_erased()) continue;
        const bool is_outlier = reproj_edge_wrap.is_monocular_ ? mono_outlier[reproj_edge_wrap.slot_] != 0 : stereo_outlier[reproj_edge_wrap.slot_] != 0;
        if (is_outlier) outlier_observations.emplace_back(std::make_pair(reproj_edge_wrap.shot_, reproj_edge_wrap.lm_));
    }

    // 8. update the information
    {
        std::lock_guard<std::mutex> lock(data::map_database::mtx_database_);
        for (auto& outlier_obs : outlier_observations) {
            auto keyfrm = outlier_obs.first;
            auto lm = outlier_obs.second;
            keyfrm->erase_landmark(lm);
            lm->erase_observation(keyfrm);
        }
        for (auto id_local_keyfrm_pair : local_keyfrms) {
            auto local_keyfrm = id_local_keyfrm_pair.second;
            local_keyfrm->set_cam_pose(se3quat_to_pose(&poses[(size_t)7 * pose_index.at(local_keyfrm)]));
        }
        for (size_t j = 0; j < lms.size(); ++j) {
            Vec3_t pos_w;
            for (int a = 0; a < 3; ++a) pos_w(a) = points[3 * j + a];
            lms[j]->set_pos_in_world(pos_w);
            lms[j]->update_normal_and_depth();
        }
    }
}

}   // namespace optimize
}   // namespace openvslam
