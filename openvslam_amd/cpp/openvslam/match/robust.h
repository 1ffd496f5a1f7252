// match::robust (expected: src/openvslam/match/robust.h). brute_force_match runs on the MI355X; the RANSAC / triangulation
// wrappers around it (match_frame_and_keyframe, match_for_triangulation) are callers and stay upstream's.
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

    unsigned int brute_force_match(data::frame& frm, data::keyframe* keyfrm, std::vector<std::pair<int, int>>& matches) const;
};

}   // namespace match
}   // namespace openvslam
