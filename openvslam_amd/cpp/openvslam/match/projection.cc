// match::projection::match_frame_and_landmarks over the C ABI. Replaces that function's body in src/openvslam/match/projection.cc:
// the object graph is flattened in the order of local_landmarks, the device returns per landmark the keypoint it is written to.
#include "projection.h"

#include <algorithm>
#include <cstring>

#include "window_ctx.h"

namespace openvslam {
namespace match {

unsigned int projection::match_frame_and_landmarks(data::frame& frm, const std::vector<data::landmark*>& local_landmarks, const float margin) const {
    const int n = (int)frm.num_keypts_, m = (int)local_landmarks.size();
    if (n == 0 || m == 0) return 0;
    std::vector<float>&lm_xy = detail::scratch_vec<float>(0, (size_t)2 * m), &lm_x_right = detail::scratch_vec<float>(1, (size_t)m);
    std::vector<int32_t>& lm_level = detail::scratch_vec<int32_t>(0, (size_t)m);
    std::vector<uint8_t>&lm_valid = detail::scratch_vec<uint8_t>(0, (size_t)m), &lm_desc = detail::scratch_vec<uint8_t>(1, (size_t)32 * m),
                         &occupied = detail::scratch_vec<uint8_t>(2, (size_t)n);
    for (int l = 0; l < m; ++l) {
        const auto* lm = local_landmarks[l];
        lm_valid[l] = lm && lm->is_observable_in_tracking_ && !lm->will_be_erased();
        if (!lm_valid[l]) {   // defined values for an entry the kernels mask out (the staging vectors keep earlier calls' contents)
            lm_xy[2 * l] = lm_xy[2 * l + 1] = lm_x_right[l] = 0.0f;
            lm_level[l] = 0;
            continue;
        }
        lm_xy[2 * l] = (float)lm->reproj_in_tracking_(0);
        lm_xy[2 * l + 1] = (float)lm->reproj_in_tracking_(1);
        lm_x_right[l] = lm->x_right_in_tracking_;
        lm_level[l] = lm->scale_level_in_tracking_;
        const cv::Mat d = lm->get_descriptor();
        std::memcpy(&lm_desc[(size_t)32 * l], d.data, 32);
    }
    for (int i = 0; i < n; ++i) occupied[i] = frm.landmarks_[i] && frm.landmarks_[i]->has_observation();
    const bool stereo = !frm.stereo_x_right_.empty();
    std::vector<int32_t>& assigned = detail::scratch_vec<int32_t>(1, (size_t)m);
    std::fill(assigned.begin(), assigned.end(), -1);
    int32_t num_matches = 0;
    // the frame's keypoints, descriptors and grid are resident (uploaded by the first matcher call on this frame)
    const int device = detail::device_of(frm);
    if (!detail::guarded("ovs_projection_match_frame_and_landmarks_f", [&] {
            const auto h = detail::device_handle_of(frm);
            return ovs_projection_match_frame_and_landmarks_f(detail::window_ctx(device).get(n, m), detail::dev(h), occupied.data(), lm_xy.data(),
                                                             stereo ? lm_x_right.data() : nullptr, lm_level.data(), lm_desc.data(), lm_valid.data(), m,
                                                             frm.scale_factors_.data(), (int)frm.scale_factors_.size(), margin, lowe_ratio_,
                                                             assigned.data(), &num_matches);
        }, {frm.device_cache_.get()}, device)) {
        return 0;
    }
    for (int l = 0; l < m; ++l)
        if (assigned[l] >= 0) frm.landmarks_[assigned[l]] = local_landmarks[l];
    return (unsigned int)num_matches;
}

unsigned int projection::match_current_and_last_frames(data::frame& curr_frm, const data::frame& last_frm, const float margin) const {
    const int n_curr = (int)curr_frm.num_keypts_, n_last = (int)last_frm.num_keypts_;
    if (n_curr == 0 || n_last == 0) return 0;
    std::vector<double>& last_pos = detail::scratch_vec<double>(0, (size_t)3 * n_last);
    std::vector<uint8_t>&last_valid = detail::scratch_vec<uint8_t>(0, (size_t)n_last), &last_desc = detail::scratch_vec<uint8_t>(1, (size_t)32 * n_last),
                         &occupied = detail::scratch_vec<uint8_t>(2, (size_t)n_curr);
    for (int i = 0; i < n_last; ++i) {
        const auto* lm = last_frm.landmarks_[i];
        last_valid[i] = lm && !(i < (int)last_frm.outlier_flags_.size() && last_frm.outlier_flags_[i]);
        if (!last_valid[i]) {   // defined values for an entry the kernels mask out
            for (int a = 0; a < 3; ++a) last_pos[(size_t)3 * i + a] = 0.0;
            continue;
        }
        const Vec3_t pos_w = lm->get_pos_in_world();
        for (int a = 0; a < 3; ++a) last_pos[(size_t)3 * i + a] = pos_w(a);
        const cv::Mat d = lm->get_descriptor();
        std::memcpy(&last_desc[(size_t)32 * i], d.data, 32);
    }
    for (int i = 0; i < n_curr; ++i) occupied[i] = curr_frm.landmarks_[i] && curr_frm.landmarks_[i]->has_observation();
    const ovs_camera cam = detail::camera_of(curr_frm.camera_);
    double pose_curr[12], pose_last[12];
    detail::pose12(curr_frm.cam_pose_cw_, pose_curr);
    detail::pose12(last_frm.cam_pose_cw_, pose_last);
    std::vector<int32_t>& assigned = detail::scratch_vec<int32_t>(1, (size_t)n_last);
    std::fill(assigned.begin(), assigned.end(), -1);
    int32_t num_matches = 0;
    const int device = detail::device_of(curr_frm);
    if (!detail::guarded("ovs_projection_match_current_and_last_frames_f", [&] {
            const auto h = detail::device_handle_of(curr_frm);
            return ovs_projection_match_current_and_last_frames_f(
                      detail::window_ctx(device).get(n_curr, n_last), &cam, detail::dev(h), occupied.data(), pose_curr,
                      reinterpret_cast<const ovs_keypoint*>(last_frm.undist_keypts_.data()), last_pos.data(), last_desc.data(), last_valid.data(),
                      n_last, pose_last, curr_frm.scale_factors_.data(), (int)curr_frm.scale_factors_.size(), margin, check_orientation_ ? 1 : 0,
                      assigned.data(), &num_matches);
        }, {curr_frm.device_cache_.get()}, device)) {
        return 0;
    }
    for (int i = 0; i < n_last; ++i)
        if (assigned[i] >= 0) curr_frm.landmarks_[assigned[i]] = last_frm.landmarks_[i];
    return (unsigned int)num_matches;
}

namespace {
// per keypoint of a keyframe: its landmark's position / valid distance range / mean normal / descriptor (rows of keypoints without a
// usable landmark stay zero and are masked by `valid`)
struct kf_landmarks {
    std::vector<double> pos, normal;
    std::vector<float> dist;
    std::vector<uint8_t> desc, valid;
    template <typename PRED>
    kf_landmarks(const std::vector<data::landmark*>& lms, PRED usable) : pos(3 * lms.size()), normal(3 * lms.size()), dist(2 * lms.size()),
                                                                          desc(32 * lms.size()), valid(lms.size()) {
        for (size_t i = 0; i < lms.size(); ++i) {
            data::landmark* lm = lms[i];
            valid[i] = usable(lm, i) ? 1 : 0;
            if (!valid[i]) continue;
            const Vec3_t p = lm->get_pos_in_world(), n = lm->get_obs_mean_normal();
            for (int a = 0; a < 3; ++a) {
                pos[3 * i + a] = p(a);
                normal[3 * i + a] = n(a);
            }
            dist[2 * i] = lm->min_valid_dist_;   // raw members: the kernel widens the gate as the getters do and predicts the level from the raw maximum
            dist[2 * i + 1] = lm->max_valid_dist_;
            const cv::Mat d = lm->get_descriptor();
            std::memcpy(&desc[32 * i], d.data, 32);
        }
    }
};
}   // namespace

unsigned int projection::match_frame_and_keyframe(data::frame& curr_frm, data::keyframe* keyfrm, const std::set<data::landmark*>& already_matched_lms,
                                                  const float margin, const unsigned int hamm_dist_thr) const {
    const int n_curr = (int)curr_frm.num_keypts_, n_kf = (int)keyfrm->num_keypts_;
    if (n_curr == 0 || n_kf == 0) return 0;
    const auto lms = keyfrm->get_landmarks();
    const kf_landmarks f(lms, [&](data::landmark* lm, size_t) { return lm && !lm->will_be_erased() && !already_matched_lms.count(lm); });
    std::vector<uint8_t> occupied((size_t)n_curr);
    for (int i = 0; i < n_curr; ++i) occupied[i] = curr_frm.landmarks_[i] != nullptr;
    const ovs_camera cam = detail::camera_of(curr_frm.camera_);
    double pose[12];
    detail::pose12(curr_frm.cam_pose_cw_, pose);
    std::vector<int32_t> assigned((size_t)n_kf, -1);
    int32_t num_matches = 0;
    const int device = detail::device_of(curr_frm);
    // the current frame is resident (relocalisation calls this for every candidate keyframe on the same frame)
    if (!detail::guarded("ovs_projection_match_frame_and_keyframe_f", [&] {
            const auto h = detail::device_handle_of(curr_frm);
            return ovs_projection_match_frame_and_keyframe_f(
                      detail::window_ctx(device).get(n_curr, n_kf), &cam, detail::dev(h), occupied.data(), pose,
                      reinterpret_cast<const ovs_keypoint*>(keyfrm->undist_keypts_.data()), f.pos.data(), f.dist.data(), f.desc.data(), f.valid.data(), n_kf,
                      curr_frm.scale_factors_.data(), (int)curr_frm.scale_factors_.size(), curr_frm.log_scale_factor_, margin, hamm_dist_thr,
                      check_orientation_ ? 1 : 0, assigned.data(), &num_matches);
        }, {curr_frm.device_cache_.get()}, device)) {
        return 0;
    }
    for (int i = 0; i < n_kf; ++i)
        if (assigned[i] >= 0) curr_frm.landmarks_[assigned[i]] = lms[i];
    return (unsigned int)num_matches;
}

unsigned int projection::match_by_Sim3_transform(data::keyframe* keyfrm, const Mat44_t& Sim3_cw, const std::vector<data::landmark*>& landmarks,
                                                 std::vector<data::landmark*>& matched_lms_in_keyfrm, const float margin) const {
    const int n = (int)keyfrm->num_keypts_, m = (int)landmarks.size();
    if (n == 0 || m == 0) return 0;
    std::set<data::landmark*> already_matched(matched_lms_in_keyfrm.begin(), matched_lms_in_keyfrm.end());
    already_matched.erase(static_cast<data::landmark*>(nullptr));
    const kf_landmarks f(landmarks, [&](data::landmark* lm, size_t) { return lm && !lm->will_be_erased() && !already_matched.count(lm); });
    std::vector<uint8_t> occupied((size_t)n);
    for (int k = 0; k < n; ++k) occupied[k] = matched_lms_in_keyfrm.at((size_t)k) != nullptr;
    const ovs_camera cam = detail::camera_of(keyfrm->camera_);
    double sim3[12];
    detail::pose12(Sim3_cw, sim3);
    std::vector<int32_t> assigned((size_t)m, -1);
    int32_t num_matches = 0;
    const int device = detail::device_of(*keyfrm);
    if (!detail::guarded("ovs_projection_match_by_sim3_transform_f", [&] {
            const auto h = detail::device_handle_of(*keyfrm);
            return ovs_projection_match_by_sim3_transform_f(
                      detail::window_ctx(device).get(n, m), &cam, detail::dev(h), occupied.data(), sim3, f.pos.data(), f.dist.data(), f.normal.data(),
                      f.desc.data(), f.valid.data(), m, keyfrm->scale_factors_.data(), (int)keyfrm->scale_factors_.size(), keyfrm->log_scale_factor_,
                      margin, assigned.data(), &num_matches);
        }, {keyfrm->device_cache_.get()}, device)) {
        return 0;
    }
    for (int l = 0; l < m; ++l)
        if (assigned[l] >= 0) matched_lms_in_keyfrm[(size_t)assigned[l]] = landmarks[l];
    return (unsigned int)num_matches;
}

unsigned int projection::match_keyframes_mutually(data::keyframe* keyfrm_1, data::keyframe* keyfrm_2, std::vector<data::landmark*>& matched_lms_in_keyfrm_1,
                                                  const float& s_12, const Mat33_t& rot_12, const Vec3_t& trans_12, const float margin) const {
    const int n1 = (int)keyfrm_1->num_keypts_, n2 = (int)keyfrm_2->num_keypts_;
    if (n1 == 0 || n2 == 0) return 0;
    const auto lms_1 = keyfrm_1->get_landmarks(), lms_2 = keyfrm_2->get_landmarks();
    // upstream's two is_already_matched vectors
    std::vector<bool> matched_1((size_t)n1, false), matched_2((size_t)n2, false);
    for (int i = 0; i < n1; ++i) {
        auto* lm = matched_lms_in_keyfrm_1.at((size_t)i);
        if (!lm) continue;
        matched_1[i] = true;
        const int idx_2 = lm->get_index_in_keyframe(keyfrm_2);
        if (0 <= idx_2 && idx_2 < n2) matched_2[idx_2] = true;
    }
    const kf_landmarks f1(lms_1, [&](data::landmark* lm, size_t i) { return lm && !matched_1[i] && !lm->will_be_erased(); });
    const kf_landmarks f2(lms_2, [&](data::landmark* lm, size_t i) { return lm && !matched_2[i] && !lm->will_be_erased(); });
    const ovs_camera cam_1 = detail::camera_of(keyfrm_1->camera_), cam_2 = detail::camera_of(keyfrm_2->camera_);
    double pose_1[12], pose_2[12], R[9], t[3];
    detail::pose12(keyfrm_1->get_cam_pose(), pose_1);
    detail::pose12(keyfrm_2->get_cam_pose(), pose_2);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R[3 * i + j] = rot_12(i, j);
        t[i] = trans_12(i);
    }
    std::vector<int32_t> m21((size_t)n1, -1);
    int32_t num_matches = 0;
    const int nmax = n1 > n2 ? n1 : n2;
    const int device = detail::device_of(*keyfrm_1);
    if (!detail::guarded("ovs_projection_match_keyframes_mutually_f", [&] {
            const auto h1 = detail::device_handle_of(*keyfrm_1), h2 = detail::device_handle_on(*keyfrm_2, device);
            return ovs_projection_match_keyframes_mutually_f(
                      detail::window_ctx(device).get(nmax, nmax), &cam_1, detail::dev(h1), pose_1, f1.pos.data(), f1.dist.data(), f1.desc.data(),
                      f1.valid.data(), &cam_2, detail::dev(h2), pose_2, f2.pos.data(), f2.dist.data(), f2.desc.data(), f2.valid.data(), (double)s_12, R, t,
                      keyfrm_1->scale_factors_.data(), (int)keyfrm_1->scale_factors_.size(), keyfrm_1->log_scale_factor_, margin, m21.data(),
                      &num_matches);
        }, {keyfrm_1->device_cache_.get(), keyfrm_2->device_cache_.get()}, device)) {
        return 0;
    }
    for (int i = 0; i < n1; ++i)
        if (m21[i] >= 0) matched_lms_in_keyfrm_1[(size_t)i] = lms_2[(size_t)m21[i]];
    return (unsigned int)num_matches;
}

}   // namespace match
}   // namespace openvslam
