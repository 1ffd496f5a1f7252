// match::projection (expected: src/openvslam/match/projection.h). match_frame_and_landmarks and
// match_current_and_last_frames run on the MI355X (the other overloads: python mirrors in openvslam_amd/match.py, same flattening).
#pragma once
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
};

}   // namespace match
}   // namespace openvslam
