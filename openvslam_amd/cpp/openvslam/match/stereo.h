// match::stereo (expected: src/openvslam/match/stereo.h) with upstream's constructor: the two image pyramids are the extractors'
// image_pyramid_ members, passed by reference exactly as data::frame's stereo constructor does. The pixels the sub-pixel search reads
// never leave HBM: the device contexts behind the two vectors are found through feature::orb_extractor::device_context_of.
#pragma once
#include <vector>

#include "../feature/orb_extractor.h"

namespace openvslam {
namespace match {

class stereo {
public:
    stereo(const std::vector<cv::Mat>& left_image_pyramid, const std::vector<cv::Mat>& right_image_pyramid,
           const std::vector<cv::KeyPoint>& keypts_left, const std::vector<cv::KeyPoint>& keypts_right, const cv::Mat& descs_left,
           const cv::Mat& descs_right, const std::vector<float>& scale_factors, const std::vector<float>& inv_scale_factors,
           const float focal_x_baseline, const float true_baseline)
        : left_image_pyramid_(left_image_pyramid), right_image_pyramid_(right_image_pyramid), num_keypts_((unsigned int)keypts_left.size()),
          keypts_left_(keypts_left), keypts_right_(keypts_right), descs_left_(descs_left), descs_right_(descs_right),
          scale_factors_(scale_factors), inv_scale_factors_(inv_scale_factors), focal_x_baseline_(focal_x_baseline),
          true_baseline_(true_baseline), min_disp_(0.0f), max_disp_(focal_x_baseline_ / true_baseline_) {}

    virtual ~stereo() = default;

    void compute(std::vector<float>& stereo_x_right, std::vector<float>& depths) const;

private:
    const std::vector<cv::Mat>& left_image_pyramid_;
    const std::vector<cv::Mat>& right_image_pyramid_;
    const unsigned int num_keypts_;
    const std::vector<cv::KeyPoint>& keypts_left_;
    const std::vector<cv::KeyPoint>& keypts_right_;
    const cv::Mat& descs_left_;
    const cv::Mat& descs_right_;
    const std::vector<float>& scale_factors_;       // the device uses the extractor's own tables (identical by construction)
    const std::vector<float>& inv_scale_factors_;
    const float focal_x_baseline_, true_baseline_;
    const float min_disp_, max_disp_;
};

}   // namespace match
}   // namespace openvslam
