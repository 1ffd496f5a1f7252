// feature::orb_extractor over the C ABI (include/ovslam_hip.h). Replaces the body of src/openvslam/feature/orb_extractor.cc.
#include "orb_extractor.h"

#include <ovslam_hip.h>

#include "../util/device_policy.h"

#include <algorithm>
#include <cassert>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace openvslam {
namespace feature {

orb_extractor::orb_extractor(const orb_params& orb_params) : orb_params_(orb_params) {
    initialize();
    register_self();
}

orb_extractor::orb_extractor(const unsigned int max_num_keypts, const float scale_factor, const unsigned int num_levels,
                             const unsigned int ini_fast_thr, const unsigned int min_fast_thr, const std::vector<std::vector<float>>& mask_rects)
    : orb_extractor(orb_params(max_num_keypts, scale_factor, num_levels, ini_fast_thr, min_fast_thr, mask_rects)) {}

orb_extractor::~orb_extractor() {
    unregister_self();
    release();
}

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
    const int st = ovs_orb_create(&p, rows, cols, 1, device_, &h_);
    if (st != OVS_OK) {
        h_ = nullptr;
        throw util::device_error(st, std::string("ovs_orb_create: ") + ovs_last_error());   // caught by run_guarded in extract()
    }
    ovs_orb_set_host_pyramid(h_, download_pyramid_ ? 1 : 0);
    h_rows_ = rows;
    h_cols_ = cols;
}

void orb_extractor::create_rectangle_mask(const unsigned int cols, const unsigned int rows) {
    if (!rect_mask_.empty() && rect_mask_.rows == (int)rows && rect_mask_.cols == (int)cols) return;
    rect_mask_ = cv::Mat();
    rect_mask_.create(rows, cols, cv::CV_8UC1);
    for (unsigned y = 0; y < rows; ++y) std::fill(rect_mask_.ptr(y), rect_mask_.ptr(y) + cols, (uint8_t)255);
    for (const auto& r : orb_params_.mask_rects_) {
        // upstream: rect_mask_.rowRange(rows*y_min, rows*y_max).colRange(cols*x_min, cols*x_max) = 0
        const unsigned x0 = cols * r.at(0), x1 = cols * r.at(1), y0 = rows * r.at(2), y1 = rows * r.at(3);
        for (unsigned y = y0; y < y1 && y < rows; ++y)
            for (unsigned x = x0; x < x1 && x < cols; ++x) rect_mask_.ptr(y)[x] = 0;
    }
}

void orb_extractor::extract(const cv::_InputArray& in_image, const cv::_InputArray& in_image_mask, std::vector<cv::KeyPoint>& keypts,
                            const cv::_OutputArray& out_descriptors) {
    if (in_image.empty()) return;   // upstream: early return
    const cv::Mat image = in_image.getMat();
    assert(image.type() == cv::CV_8UC1);
    cv::Mat mask;
    if (!in_image_mask.empty()) {
        mask = in_image_mask.getMat();
    } else if (!orb_params_.mask_rects_.empty()) {
        create_rectangle_mask(image.cols, image.rows);
        mask = rect_mask_;
    }
    int n = 0;
    // ONE call, ONE wait: upload (banded through pinned memory), pyramid -> FAST -> quad-tree -> describe, one D2H of the results and
    // -- when image_pyramid_ is wanted on the host -- one D2H of the whole pyramid block into pinned memory the Mats below alias.
    // No silent CPU fallback; the failure policy of util/device_policy.h instead: one retry on a rebuilt handle, then
    // a frame without keypoints (tracking then fails for this frame and relocalises, as it does for any featureless frame).
    std::vector<std::pair<const uint8_t*, int>> pyr_base(orb_params_.num_levels_);
    std::vector<int> pyr_rows(orb_params_.num_levels_), pyr_cols(orb_params_.num_levels_);
    const bool ok = util::run_guarded(
        "orb_extractor::extract",
        [&] {
            ensure_handle(image.rows, image.cols);
            const int cap = ovs_orb_max_keypoints(h_);
            keypts.resize(cap);
            desc_buf_.resize((size_t)cap * 32);
            int st = ovs_orb_extract(h_, image.data, image.rows, image.cols, image.step, mask.empty() ? nullptr : mask.data,
                                     mask.empty() ? 0 : mask.step, reinterpret_cast<ovs_keypoint*>(keypts.data()), desc_buf_.data(), cap, &n);
            for (unsigned int l = 1; st == OVS_OK && download_pyramid_ && l < orb_params_.num_levels_; ++l) {
                int pitch = 0;
                st = ovs_orb_host_pyramid_level(h_, (int)l, &pyr_base[l].first, &pyr_rows[l], &pyr_cols[l], &pitch);
                pyr_base[l].second = pitch;
            }
            return st;
        },
        [&] { release(); });
    if (!ok) {
        release();
        keypts.clear();
        out_descriptors.create(0, 32, cv::CV_8U);
        image_pyramid_[0] = image;
        for (unsigned int l = 1; l < orb_params_.num_levels_; ++l) image_pyramid_[l] = cv::Mat();
        return;
    }
    keypts.resize(n);
    out_descriptors.create(n, 32, cv::CV_8U);
    if (n) {
        cv::Mat descriptors = out_descriptors.getMat();
        for (int i = 0; i < n; ++i) std::copy(desc_buf_.begin() + (size_t)i * 32, desc_buf_.begin() + (size_t)(i + 1) * 32, descriptors.ptr(i));
    }
    // image_pyramid_: level 0 aliases the caller's image (as upstream: `image_pyramid_.at(0) = image`), levels >= 1 alias the handle's
    // pinned copy of the device pyramid (no per-level copies); with the download disabled the vector holds empty Mats above level 0
    image_pyramid_[0] = image;
    for (unsigned int l = 1; l < orb_params_.num_levels_; ++l) {
        if (!download_pyramid_) {
            image_pyramid_[l] = cv::Mat();
            continue;
        }
        image_pyramid_[l] = cv::Mat(pyr_rows[l], pyr_cols[l], cv::CV_8U, const_cast<uint8_t*>(pyr_base[l].first), (size_t)pyr_base[l].second);
    }
}

void orb_extractor::set_image_pyramid_download(const bool enable) {
    download_pyramid_ = enable;
    if (h_) ovs_orb_set_host_pyramid(h_, enable ? 1 : 0);
}

// ---- registry: &extractor->image_pyramid_ -> device context. match::stereo keeps upstream's ctor (it receives the two extractors'
// image_pyramid_ members by reference) and finds the pyramids where they lie, in HBM, through this table.
namespace {
std::mutex g_registry_mtx;
std::unordered_map<const std::vector<cv::Mat>*, const orb_extractor*> g_registry;
}   // namespace

void orb_extractor::register_self() {
    std::lock_guard<std::mutex> lock(g_registry_mtx);
    g_registry[&image_pyramid_] = this;
}
void orb_extractor::unregister_self() {
    std::lock_guard<std::mutex> lock(g_registry_mtx);
    g_registry.erase(&image_pyramid_);
}
const ovs_orb* orb_extractor::device_context_of(const std::vector<cv::Mat>& image_pyramid) {
    std::lock_guard<std::mutex> lock(g_registry_mtx);
    const auto it = g_registry.find(&image_pyramid);
    return it == g_registry.end() ? nullptr : it->second->h_;
}

}   // namespace feature
}   // namespace openvslam
