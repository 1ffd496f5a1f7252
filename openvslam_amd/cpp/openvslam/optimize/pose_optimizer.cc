// optimize::pose_optimizer::optimize over the C ABI. Replaces that function's body in src/openvslam/optimize/pose_optimizer.cc.
#include "pose_optimizer.h"

#include <ovslam_hip.h>

#include "../util/device_policy.h"

#include <stdexcept>
#include <string>
#include <vector>

namespace openvslam {
namespace optimize {

unsigned int pose_optimizer::optimize(data::frame& frm) const {
    if (num_trials_ != 4 || num_each_iter_ != 10) throw std::runtime_error("pose_optimizer: only upstream's 4 x 10 schedule is implemented");
    // upstream's `switch (frm.camera_->model_type_)`: Perspective, Fisheye and RadialDivision all create perspective_pose_opt_edge (mono / stereo)
    // with the model's fx, fy, cx, cy -- the keypoints are undistorted --, Equirectangular its own edge (ORACLE_SPEC rule 31)
    const bool equirect = frm.camera_->model_type_ == camera::model_type_t::Equirectangular;
    const unsigned int n = frm.num_keypts_;
    std::vector<ovs_pose_obs> obs;
    std::vector<unsigned int> idx_of;
    obs.reserve(n);
    idx_of.reserve(n);
    if (frm.outlier_flags_.size() != n) frm.outlier_flags_.assign(n, false);
    const bool has_stereo = !frm.stereo_x_right_.empty();
    for (unsigned int idx = 0; idx < n; ++idx) {
        const auto* lm = frm.landmarks_[idx];
        if (!lm || lm->will_be_erased()) continue;
        frm.outlier_flags_[idx] = false;
        const auto& kp = frm.undist_keypts_[idx];
        const Vec3_t p = lm->get_pos_in_world();
        ovs_pose_obs o{};
        o.pos_w[0] = p(0);
        o.pos_w[1] = p(1);
        o.pos_w[2] = p(2);
        o.obs_x = kp.pt.x;
        o.obs_y = kp.pt.y;
        o.is_stereo = !equirect && has_stereo && frm.stereo_x_right_[idx] >= 0;
        o.obs_x_right = o.is_stereo ? frm.stereo_x_right_[idx] : 0.0;
        o.inv_sigma_sq = frm.inv_level_sigma_sq_.at((size_t)kp.octave);
        obs.push_back(o);
        idx_of.push_back(idx);
    }
    if (obs.size() < 5) return 0;
    double pose_in[12], pose_out[12];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) pose_in[3 * i + j] = frm.cam_pose_cw_(i, j);
        pose_in[9 + i] = frm.cam_pose_cw_(i, 3);
    }
    const ovs_ba_cam cam = {frm.camera_->fx_, frm.camera_->fy_, frm.camera_->cx_, frm.camera_->cy_};
    std::vector<uint8_t> outlier(obs.size());
    int32_t num_valid = 0;
    // failure policy (util/device_policy.h): one retry, then "no inliers" with the pose and the outlier flags left as they were -- tracking
    // treats that as a failed track of this frame
    if (!util::run_guarded(
            "ovs_pose_optimize",
            [&] {
                return equirect ? ovs_pose_optimize_equirect(0, pose_in, obs.data(), (int32_t)obs.size(), (int32_t)frm.camera_->cols_,
                                                             (int32_t)frm.camera_->rows_, pose_out, outlier.data(), &num_valid)
                                : ovs_pose_optimize(0, pose_in, obs.data(), (int32_t)obs.size(), &cam, frm.camera_->focal_x_baseline_,
                                                    (int32_t)frm.camera_->setup_type_, pose_out, outlier.data(), &num_valid);
            },
            [] {}))
        return 0;
    for (size_t k = 0; k < obs.size(); ++k) frm.outlier_flags_[idx_of[k]] = outlier[k] != 0;
    Mat44_t T;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T(i, j) = pose_out[3 * i + j];
        T(i, 3) = pose_out[9 + i];
    }
    frm.set_cam_pose(T);
    return (unsigned int)num_valid;
}

}   // namespace optimize
}   // namespace openvslam
