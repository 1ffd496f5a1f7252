// match::stereo (expected: src/openvslam/match/stereo.h). Upstream's ctor takes the two extractors' image_pyramid_ vectors; here the
// pyramids stay in HBM, so the ctor takes the two extractors themselves (the only signature change on this path; the caller,
// data::frame's stereo ctor, owns both extractors).
#pragma once
#include <vector>

#include "../feature/orb_extractor.h"

namespace openvslam {
namespace match {

class stereo {
public:
    stereo(const feature::orb_extractor* extractor_left, const feature::orb_extractor* extractor_right,
           const std::vector<cv::KeyPoint>& keypts_left, const std::vector<cv::KeyPoint>& keypts_right, const cv::Mat& descs_left,
           const cv::Mat& descs_right, const float focal_x_baseline, const float true_baseline)
        : extractor_left_(extractor_left), extractor_right_(extractor_right), keypts_left_(keypts_left), keypts_right_(keypts_right),
          descs_left_(descs_left), descs_right_(descs_right), focal_x_baseline_(focal_x_baseline), true_baseline_(true_baseline) {}

    void compute(std::vector<float>& stereo_x_right, std::vector<float>& depths) const;

private:
    const feature::orb_extractor* extractor_left_;
    const feature::orb_extractor* extractor_right_;
    const std::vector<cv::KeyPoint>& keypts_left_;
    const std::vector<cv::KeyPoint>& keypts_right_;
    const cv::Mat& descs_left_;
    const cv::Mat& descs_right_;
    const float focal_x_baseline_, true_baseline_;
};

}   // namespace match
}   // namespace openvslam
