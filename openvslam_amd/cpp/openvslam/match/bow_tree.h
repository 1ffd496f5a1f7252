// match::bow_tree (expected: src/openvslam/match/bow_tree.h).
#pragma once
#include <vector>

#include "../data/frame_stub.h"
#include "base.h"

namespace openvslam {
namespace match {

class bow_tree final : public base {
public:
    explicit bow_tree(const float lowe_ratio = 0.6, const bool check_orientation = true) : base(lowe_ratio, check_orientation) {}
    ~bow_tree() final = default;

    unsigned int match_frame_and_keyframe(data::keyframe* keyfrm, data::frame& frm, std::vector<data::landmark*>& matched_lms_in_frm) const;

    //! loop detection: landmarks of two keyframes through their common vocabulary nodes
    unsigned int match_keyframes(data::keyframe* keyfrm_1, data::keyframe* keyfrm_2, std::vector<data::landmark*>& matched_lms_in_keyfrm_1) const;
};

}   // namespace match
}   // namespace openvslam
