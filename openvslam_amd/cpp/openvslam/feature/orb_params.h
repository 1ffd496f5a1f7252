// feature::orb_params -- same fields and defaults as upstream (expected: src/openvslam/feature/orb_params.h).
#pragma once
#include <vector>

namespace openvslam {
namespace feature {

struct orb_params {
    orb_params() = default;
    orb_params(const unsigned int max_num_keypts, const float scale_factor, const unsigned int num_levels, const unsigned int ini_fast_thr,
               const unsigned int min_fast_thr, const std::vector<std::vector<float>>& mask_rects = {})
        : max_num_keypts_(max_num_keypts), scale_factor_(scale_factor), num_levels_(num_levels), ini_fast_thr_(ini_fast_thr),
          min_fast_thr(min_fast_thr), mask_rects_(mask_rects) {}

    unsigned int max_num_keypts_ = 2000;
    float scale_factor_ = 1.2;
    unsigned int num_levels_ = 8;
    unsigned int ini_fast_thr_ = 20;
    unsigned int min_fast_thr = 7;
    //! each rectangle: [x_min / cols, x_max / cols, y_min / rows, y_max / rows]
    std::vector<std::vector<float>> mask_rects_;
};

}   // namespace feature
}   // namespace openvslam
