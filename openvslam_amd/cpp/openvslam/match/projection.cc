// match::projection::match_frame_and_landmarks over the C ABI. Replaces that function's body in src/openvslam/match/projection.cc:
// the object graph is flattened in the order of local_landmarks, the device returns per landmark the keypoint it is written to.
#include "projection.h"

#include <cstring>

#include "window_ctx.h"

namespace openvslam {
namespace match {

unsigned int projection::match_frame_and_landmarks(data::frame& frm, const std::vector<data::landmark*>& local_landmarks, const float margin) const {
    const int n = (int)frm.num_keypts_, m = (int)local_landmarks.size();
    if (n == 0 || m == 0) return 0;
    std::vector<float> lm_xy((size_t)2 * m), lm_x_right((size_t)m);
    std::vector<int32_t> lm_level((size_t)m);
    std::vector<uint8_t> lm_valid((size_t)m), lm_desc((size_t)32 * m), occupied((size_t)n);
    for (int l = 0; l < m; ++l) {
        const auto* lm = local_landmarks[l];
        lm_valid[l] = lm && lm->is_observable_in_tracking_ && !lm->will_be_erased();
        if (!lm_valid[l]) continue;
        lm_xy[2 * l] = (float)lm->reproj_in_tracking_(0);
        lm_xy[2 * l + 1] = (float)lm->reproj_in_tracking_(1);
        lm_x_right[l] = lm->x_right_in_tracking_;
        lm_level[l] = lm->scale_level_in_tracking_;
        const cv::Mat d = lm->get_descriptor();
        std::memcpy(&lm_desc[(size_t)32 * l], d.data, 32);
    }
    for (int i = 0; i < n; ++i) occupied[i] = frm.landmarks_[i] && frm.landmarks_[i]->has_observation();
    const bool stereo = !frm.stereo_x_right_.empty();
    const ovs_grid_params gp = detail::grid_of(frm.camera_);
    std::vector<int32_t> assigned((size_t)m, -1);
    int32_t num_matches = 0;
    detail::check(ovs_projection_match_frame_and_landmarks(
                      detail::window_ctx().get(n, m), &gp, reinterpret_cast<const ovs_keypoint*>(frm.undist_keypts_.data()), frm.descriptors_.data,
                      stereo ? frm.stereo_x_right_.data() : nullptr, occupied.data(), n, lm_xy.data(), stereo ? lm_x_right.data() : nullptr,
                      lm_level.data(), lm_desc.data(), lm_valid.data(), m, frm.scale_factors_.data(), (int)frm.scale_factors_.size(), margin,
                      lowe_ratio_, assigned.data(), &num_matches),
                  "ovs_projection_match_frame_and_landmarks");
    for (int l = 0; l < m; ++l)
        if (assigned[l] >= 0) frm.landmarks_[assigned[l]] = local_landmarks[l];
    return (unsigned int)num_matches;
}

unsigned int projection::match_current_and_last_frames(data::frame& curr_frm, const data::frame& last_frm, const float margin) const {
    const int n_curr = (int)curr_frm.num_keypts_, n_last = (int)last_frm.num_keypts_;
    if (n_curr == 0 || n_last == 0) return 0;
    std::vector<double> last_pos((size_t)3 * n_last);
    std::vector<uint8_t> last_valid((size_t)n_last), last_desc((size_t)32 * n_last), occupied((size_t)n_curr);
    for (int i = 0; i < n_last; ++i) {
        const auto* lm = last_frm.landmarks_[i];
        last_valid[i] = lm && !(i < (int)last_frm.outlier_flags_.size() && last_frm.outlier_flags_[i]);
        if (!last_valid[i]) continue;
        const Vec3_t pos_w = lm->get_pos_in_world();
        for (int a = 0; a < 3; ++a) last_pos[(size_t)3 * i + a] = pos_w(a);
        const cv::Mat d = lm->get_descriptor();
        std::memcpy(&last_desc[(size_t)32 * i], d.data, 32);
    }
    for (int i = 0; i < n_curr; ++i) occupied[i] = curr_frm.landmarks_[i] && curr_frm.landmarks_[i]->has_observation();
    const bool stereo = !curr_frm.stereo_x_right_.empty();
    const ovs_grid_params gp = detail::grid_of(curr_frm.camera_);
    const ovs_camera cam = detail::camera_of(curr_frm.camera_);
    double pose_curr[12], pose_last[12];
    detail::pose12(curr_frm.cam_pose_cw_, pose_curr);
    detail::pose12(last_frm.cam_pose_cw_, pose_last);
    std::vector<int32_t> assigned((size_t)n_last, -1);
    int32_t num_matches = 0;
    detail::check(ovs_projection_match_current_and_last_frames(
                      detail::window_ctx().get(n_curr, n_last), &cam, &gp, reinterpret_cast<const ovs_keypoint*>(curr_frm.undist_keypts_.data()),
                      curr_frm.descriptors_.data, stereo ? curr_frm.stereo_x_right_.data() : nullptr, occupied.data(), n_curr, pose_curr,
                      reinterpret_cast<const ovs_keypoint*>(last_frm.undist_keypts_.data()), last_pos.data(), last_desc.data(), last_valid.data(),
                      n_last, pose_last, curr_frm.scale_factors_.data(), (int)curr_frm.scale_factors_.size(), margin, check_orientation_ ? 1 : 0,
                      assigned.data(), &num_matches),
                  "ovs_projection_match_current_and_last_frames");
    for (int i = 0; i < n_last; ++i)
        if (assigned[i] >= 0) curr_frm.landmarks_[assigned[i]] = last_frm.landmarks_[i];
    return (unsigned int)num_matches;
}

}   // namespace match
}   // namespace openvslam
