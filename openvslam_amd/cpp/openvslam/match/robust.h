// match::robust (expected: src/openvslam/match/robust.h). brute_force_match and match_for_triangulation (incl.
// check_epipolar_constraint) run on the MI355X; match_frame_and_keyframe = brute_force_match + upstream's host-side essential-matrix RANSAC.
#pragma once
#include <utility>
#include <vector>

#include "../data/frame_stub.h"
#include "base.h"

namespace openvslam {
namespace match {

class robust final : public base {
public:
    explicit robust(const float lowe_ratio, const bool check_orientation) : base(lowe_ratio, check_orientation) {}
    ~robust() final = default;

    //! brute-force match + upstream's host-side essential-matrix RANSAC (solve::essential_solver); only the inliers are written to
    //! matched_lms_in_frm (indexed by frame keypoint)
    unsigned int match_frame_and_keyframe(data::frame& frm, data::keyframe* keyfrm, std::vector<data::landmark*>& matched_lms_in_frm);

    unsigned int brute_force_match(data::frame& frm, data::keyframe* keyfrm, std::vector<std::pair<int, int>>& matches) const;

    //! keypoints of two keyframes without landmarks, through the common BoW nodes, gated by the epipolar constraint of E_12
    unsigned int match_for_triangulation(data::keyframe* keyfrm_1, data::keyframe* keyfrm_2, const Mat33_t& E_12,
                                         std::vector<std::pair<unsigned int, unsigned int>>& matched_idx_pairs) const;
};

}   // namespace match
}   // namespace openvslam
