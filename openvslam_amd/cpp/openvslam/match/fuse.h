// match::fuse (expected: src/openvslam/match/fuse.h): duplicate-landmark detection of the mapping and loop-closing threads. The candidate
// search (reproject -> window -> best Hamming) runs on the MI355X; the landmark-graph surgery stays here, in upstream's order.
#pragma once
#include <vector>

#include "../data/frame_stub.h"
#include "base.h"

namespace openvslam {
namespace match {

class fuse final : public base {
public:
    explicit fuse(const float lowe_ratio = 0.6) : base(lowe_ratio, true) {}
    ~fuse() final = default;

    //! landmarks_to_check against keyfrm: merge with the keyframe's landmark at the best keypoint, or add the observation
    template <typename T>
    unsigned int replace_duplication(data::keyframe* keyfrm, const T& landmarks_to_check, const float margin = 3.0) const;

    //! loop closing: candidates under the Sim3-corrected pose; duplicated_lms_in_keyfrm[i] = the keyframe's landmark to replace
    unsigned int detect_duplication(data::keyframe* keyfrm, const Mat44_t& Sim3_cw, const std::vector<data::landmark*>& landmarks_to_check,
                                    const float margin, std::vector<data::landmark*>& duplicated_lms_in_keyfrm) const;
};

}   // namespace match
}   // namespace openvslam
