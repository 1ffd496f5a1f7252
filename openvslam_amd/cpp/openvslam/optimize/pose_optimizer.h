// optimize::pose_optimizer (expected: src/openvslam/optimize/pose_optimizer.h): the pose-only bundle adjustment of the tracking thread.
// All four optimisation rounds run in one kernel launch on the MI355X (csrc/pose_opt.hip).
#pragma once
#include "../data/frame_stub.h"

namespace openvslam {
namespace optimize {

class pose_optimizer {
public:
    explicit pose_optimizer(const unsigned int num_trials = 4, const unsigned int num_each_iter = 10)
        : num_trials_(num_trials), num_each_iter_(num_each_iter) {}
    virtual ~pose_optimizer() = default;

    //! optimises frm.cam_pose_cw_, sets frm.outlier_flags_; returns the number of inlier observations
    unsigned int optimize(data::frame& frm) const;

private:
    const unsigned int num_trials_, num_each_iter_;   // upstream's defaults are what the kernel implements
};

}   // namespace optimize
}   // namespace openvslam
