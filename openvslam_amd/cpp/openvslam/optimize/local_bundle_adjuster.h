// optimize::local_bundle_adjuster (expected: src/openvslam/optimize/local_bundle_adjuster.h): local bundle adjustment of the mapping
// thread. The graph is collected exactly as upstream collects it (local keyframes = the current keyframe + its covisibilities, local
// landmarks = what they observe, fixed keyframes = the other observers); the two optimisation rounds run through ovs_local_ba_optimize
// (linearisations on the MI355X); outlier observations are erased and poses / positions written back under map_database::mtx_database_.
#pragma once
#include "../data/frame_stub.h"

namespace openvslam {
namespace optimize {

class local_bundle_adjuster {
public:
    explicit local_bundle_adjuster(const unsigned int num_first_iter = 5, const unsigned int num_second_iter = 10)
        : num_first_iter_(num_first_iter), num_second_iter_(num_second_iter) {}
    virtual ~local_bundle_adjuster() = default;

    void optimize(data::keyframe* curr_keyfrm, bool* const force_stop_flag) const;

private:
    const unsigned int num_first_iter_;
    const unsigned int num_second_iter_;
};

}   // namespace optimize
}   // namespace openvslam
