// feature::orb_extractor over the C ABI (include/ovslam_hip.h). Replaces the body of src/openvslam/feature/orb_extractor.cc.
#include "orb_extractor.h"

#include <ovslam_hip.h>

#include <cassert>
#include <stdexcept>
#include <string>

namespace openvslam {
namespace feature {

namespace {
[[noreturn]] void fail(const char* where, int st) {
    throw std::runtime_error(std::string(where) + " failed (" + std::to_string(st) + "): " + ovs_last_error());
}
}   // namespace

orb_extractor::orb_extractor(const orb_params& orb_params) : orb_params_(orb_params) { initialize(); }

orb_extractor::orb_extractor(const unsigned int max_num_keypts, const float scale_factor, const unsigned int num_levels,
                             const unsigned int ini_fast_thr, const unsigned int min_fast_thr, const std::vector<std::vector<float>>& mask_rects)
    : orb_extractor(orb_params(max_num_keypts, scale_factor, num_levels, ini_fast_thr, min_fast_thr, mask_rects)) {}

orb_extractor::~orb_extractor() { release(); }

void orb_extractor::release() {
    if (h_) ovs_orb_destroy(h_);
    h_ = nullptr;
    h_rows_ = h_cols_ = 0;
}

void orb_extractor::initialize() {
    // upstream: calc_scale_factors etc. -- same cumulative-float-product rule, evaluated on the host (no device needed)
    const unsigned int L = orb_params_.num_levels_;
    scale_factors_.assign(L, 1.0f);
    inv_scale_factors_.assign(L, 1.0f);
    level_sigma_sq_.assign(L, 1.0f);
    inv_level_sigma_sq_.assign(L, 1.0f);
    for (unsigned int l = 1; l < L; ++l) scale_factors_[l] = orb_params_.scale_factor_ * scale_factors_[l - 1];
    for (unsigned int l = 0; l < L; ++l) {
        inv_scale_factors_[l] = 1.0f / scale_factors_[l];
        level_sigma_sq_[l] = scale_factors_[l] * scale_factors_[l];
        inv_level_sigma_sq_[l] = 1.0f / level_sigma_sq_[l];
    }
    image_pyramid_.resize(L);
    release();   // parameters changed: the device handle is rebuilt lazily
}

void orb_extractor::set_max_num_keypoints(const unsigned int v) { orb_params_.max_num_keypts_ = v; initialize(); }
void orb_extractor::set_scale_factor(const float v) { orb_params_.scale_factor_ = v; initialize(); }
void orb_extractor::set_num_scale_levels(const unsigned int v) { orb_params_.num_levels_ = v; initialize(); }
void orb_extractor::set_initial_fast_threshold(const unsigned int v) { orb_params_.ini_fast_thr_ = v; initialize(); }
void orb_extractor::set_minimum_fast_threshold(const unsigned int v) { orb_params_.min_fast_thr = v; initialize(); }

void orb_extractor::ensure_handle(int rows, int cols) {
    if (h_ && rows <= h_rows_ && cols <= h_cols_) return;
    release();
    ovs_orb_params p;
    p.max_num_keypts = (int32_t)orb_params_.max_num_keypts_;
    p.scale_factor = orb_params_.scale_factor_;
    p.num_levels = (int32_t)orb_params_.num_levels_;
    p.ini_fast_thr = (int32_t)orb_params_.ini_fast_thr_;
    p.min_fast_thr = (int32_t)orb_params_.min_fast_thr;
    const int st = ovs_orb_create(&p, rows, cols, 1, 0, &h_);
    if (st != OVS_OK) fail("ovs_orb_create", st);
    h_rows_ = rows;
    h_cols_ = cols;
}

void orb_extractor::create_rectangle_mask(const unsigned int cols, const unsigned int rows) {
    if (!rect_mask_.empty() && rect_mask_.rows == (int)rows && rect_mask_.cols == (int)cols) return;
    rect_mask_ = cv::Mat();
    rect_mask_.create(rows, cols, cv::CV_8UC1);
    std::fill(rect_mask_.storage.begin(), rect_mask_.storage.end(), (uint8_t)255);
    for (const auto& r : orb_params_.mask_rects_) {
        // upstream: rect_mask_.rowRange(rows*y_min, rows*y_max).colRange(cols*x_min, cols*x_max) = 0
        const unsigned x0 = cols * r.at(0), x1 = cols * r.at(1), y0 = rows * r.at(2), y1 = rows * r.at(3);
        for (unsigned y = y0; y < y1 && y < rows; ++y)
            for (unsigned x = x0; x < x1 && x < cols; ++x) rect_mask_.ptr(y)[x] = 0;
    }
}

void orb_extractor::extract(const cv::_InputArray& in_image, const cv::_InputArray& in_image_mask, std::vector<cv::KeyPoint>& keypts,
                            cv::_OutputArray& out_descriptors) {
    if (in_image.empty()) return;   // upstream: early return
    const cv::Mat& image = in_image;
    assert(image.type() == cv::CV_8UC1);
    const cv::Mat* mask = nullptr;
    if (!in_image_mask.empty()) {
        mask = &in_image_mask;
    } else if (!orb_params_.mask_rects_.empty()) {
        create_rectangle_mask(image.cols, image.rows);
        mask = &rect_mask_;
    }
    ensure_handle(image.rows, image.cols);
    const int cap = ovs_orb_max_keypoints(h_);
    keypts.resize(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0;
    const int st = ovs_orb_extract(h_, image.data, image.rows, image.cols, image.step, mask ? mask->data : nullptr, mask ? mask->step : 0,
                                   reinterpret_cast<ovs_keypoint*>(keypts.data()), desc.data(), cap, &n);
    if (st != OVS_OK) fail("ovs_orb_extract", st);   // no silent CPU fallback
    keypts.resize(n);
    out_descriptors = cv::Mat();
    out_descriptors.create(n, 32, cv::CV_8U);
    if (n) std::copy(desc.begin(), desc.begin() + (size_t)n * 32, out_descriptors.data);
    for (unsigned int l = 0; l < orb_params_.num_levels_; ++l) {
        int r = 0, c = 0;
        int s2 = ovs_orb_pyramid_level(h_, 0, l, nullptr, &r, &c);
        if (s2 != OVS_OK) fail("ovs_orb_pyramid_level", s2);
        image_pyramid_[l] = cv::Mat();
        image_pyramid_[l].create(r, c, cv::CV_8U);
        s2 = ovs_orb_pyramid_level(h_, 0, l, image_pyramid_[l].data, &r, &c);
        if (s2 != OVS_OK) fail("ovs_orb_pyramid_level", s2);
    }
}

}   // namespace feature
}   // namespace openvslam
