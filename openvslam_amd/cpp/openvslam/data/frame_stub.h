// Minimal stand-ins for data::frame / data::keyframe / data::landmark / camera::base: only the members the matcher shims read
// (expected: src/openvslam/data/{frame,keyframe,landmark}.h, src/openvslam/camera/base.h). In an OpenVSLAM checkout the real
// headers are used instead and the shim bodies compile unchanged.
#pragma once
#include <map>
#include <vector>

#include "../../cv_stub.h"

namespace openvslam {

struct Vec2_t {   // Eigen::Vector2d stand-in
    double v[2] = {0, 0};
    double operator()(int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
};

namespace camera {
struct image_bounds {
    float min_x_ = 0, max_x_ = 0, min_y_ = 0, max_y_ = 0;
};
class base {
public:
    unsigned int cols_ = 0, rows_ = 0;
    image_bounds img_bounds_;
    unsigned int num_grid_cols_ = 64, num_grid_rows_ = 48;
    float focal_x_baseline_ = 0, true_baseline_ = 0;
};
}   // namespace camera

namespace data {

using bow_feature_vector = std::map<unsigned int, std::vector<unsigned int>>;   // DBoW2::FeatureVector

class landmark {
public:
    bool will_be_erased() const { return will_be_erased_; }
    bool has_observation() const { return num_observations_ > 0; }
    cv::Mat get_descriptor() const { return descriptor_; }
    bool will_be_erased_ = false;
    unsigned int num_observations_ = 1;
    cv::Mat descriptor_;
    // tracking information
    Vec2_t reproj_in_tracking_;
    float x_right_in_tracking_ = -1.0f;
    bool is_observable_in_tracking_ = false;
    int scale_level_in_tracking_ = 0;
};

class frame {
public:
    unsigned int num_keypts_ = 0;
    std::vector<cv::KeyPoint> keypts_;
    std::vector<cv::KeyPoint> undist_keypts_;
    std::vector<float> stereo_x_right_;
    cv::Mat descriptors_;
    std::vector<landmark*> landmarks_;
    std::vector<float> scale_factors_;
    camera::base* camera_ = nullptr;
    bow_feature_vector bow_feat_vec_;
};

class keyframe {
public:
    unsigned int num_keypts_ = 0;
    std::vector<cv::KeyPoint> keypts_;
    std::vector<cv::KeyPoint> undist_keypts_;
    cv::Mat descriptors_;
    std::vector<landmark*> landmarks_;
    bow_feature_vector bow_feat_vec_;
    std::vector<landmark*> get_landmarks() const { return landmarks_; }
};

}   // namespace data
}   // namespace openvslam
