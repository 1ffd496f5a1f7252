// match::projection (expected: src/openvslam/match/projection.h). every matcher runs on the MI355X.
#pragma once
#include <set>
#include <vector>

#include "../data/frame_stub.h"
#include "base.h"

namespace openvslam {
namespace match {

class projection final : public base {
public:
    explicit projection(const float lowe_ratio = 0.6, const bool check_orientation = true) : base(lowe_ratio, check_orientation) {}
    ~projection() final = default;

    //! 3D points already projected by tracking_module::search_local_landmarks -> frame keypoints (frm.landmarks_ is updated)
    unsigned int match_frame_and_landmarks(data::frame& frm, const std::vector<data::landmark*>& local_landmarks, const float margin = 5.0) const;

    //! last frame's 3D points reprojected with the current pose (motion model) -> current keypoints (curr_frm.landmarks_ is updated)
    unsigned int match_current_and_last_frames(data::frame& curr_frm, const data::frame& last_frm, const float margin) const;

    //! a keyframe's 3D points (except already_matched_lms) reprojected with the current pose -> current keypoints (relocalisation)
    unsigned int match_frame_and_keyframe(data::frame& curr_frm, data::keyframe* keyfrm, const std::set<data::landmark*>& already_matched_lms,
                                          const float margin, const unsigned int hamm_dist_thr) const;

    //! loop closing: 3D points reprojected with the Sim3-corrected pose -> keyframe keypoints (matched_lms_in_keyfrm is extended)
    unsigned int match_by_Sim3_transform(data::keyframe* keyfrm, const Mat44_t& Sim3_cw, const std::vector<data::landmark*>& landmarks,
                                         std::vector<data::landmark*>& matched_lms_in_keyfrm, const float margin) const;

    //! Sim3 refinement: keyframe 1's points into keyframe 2 and back, pairs both directions agree on
    unsigned int match_keyframes_mutually(data::keyframe* keyfrm_1, data::keyframe* keyfrm_2, std::vector<data::landmark*>& matched_lms_in_keyfrm_1,
                                          const float& s_12, const Mat33_t& rot_12, const Vec3_t& trans_12, const float margin) const;
};

}   // namespace match
}   // namespace openvslam
