// Minimal stand-ins for data::frame / data::keyframe / data::landmark: only the members match::robust::brute_force_match reads
// (expected: src/openvslam/data/{frame,keyframe,landmark}.h). In an OpenVSLAM checkout the real headers are used instead.
#pragma once
#include <vector>

#include "../../cv_stub.h"

namespace openvslam {
namespace data {

class landmark {
public:
    bool will_be_erased() const { return will_be_erased_; }
    bool will_be_erased_ = false;
};

class frame {
public:
    unsigned int num_keypts_ = 0;
    std::vector<cv::KeyPoint> keypts_;
    cv::Mat descriptors_;
};

class keyframe {
public:
    unsigned int num_keypts_ = 0;
    std::vector<cv::KeyPoint> keypts_;
    cv::Mat descriptors_;
    std::vector<landmark*> landmarks_;
    std::vector<landmark*> get_landmarks() const { return landmarks_; }
};

}   // namespace data
}   // namespace openvslam
