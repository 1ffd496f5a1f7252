// feature::orb_extractor -- upstream's public surface (expected: src/openvslam/feature/orb_extractor.h), body on the MI355X.
// Header-compatible drop-in: construct with orb_params, call extract(image, mask, keypts, descriptors), read image_pyramid_.
#pragma once
#include <vector>

#include "../../cv_stub.h"
#include "orb_params.h"

struct ovs_orb;

namespace openvslam {
namespace feature {

class orb_extractor {
public:
    orb_extractor() = delete;
    explicit orb_extractor(const orb_params& orb_params);
    orb_extractor(const unsigned int max_num_keypts, const float scale_factor, const unsigned int num_levels, const unsigned int ini_fast_thr,
                  const unsigned int min_fast_thr, const std::vector<std::vector<float>>& mask_rects = {});
    virtual ~orb_extractor();
    orb_extractor(const orb_extractor&) = delete;
    orb_extractor& operator=(const orb_extractor&) = delete;

    //! Extract keypoints and each descriptor of them
    void extract(const cv::_InputArray& in_image, const cv::_InputArray& in_image_mask, std::vector<cv::KeyPoint>& keypts,
                 const cv::_OutputArray& out_descriptors);

    unsigned int get_max_num_keypoints() const { return orb_params_.max_num_keypts_; }
    void set_max_num_keypoints(const unsigned int max_num_keypts);
    float get_scale_factor() const { return orb_params_.scale_factor_; }
    void set_scale_factor(const float scale_factor);
    unsigned int get_num_scale_levels() const { return orb_params_.num_levels_; }
    void set_num_scale_levels(const unsigned int num_levels);
    unsigned int get_initial_fast_threshold() const { return orb_params_.ini_fast_thr_; }
    void set_initial_fast_threshold(const unsigned int initial_fast_threshold);
    unsigned int get_minimum_fast_threshold() const { return orb_params_.min_fast_thr; }
    void set_minimum_fast_threshold(const unsigned int minimum_fast_threshold);

    std::vector<float> get_scale_factors() const { return scale_factors_; }
    std::vector<float> get_inv_scale_factors() const { return inv_scale_factors_; }
    std::vector<float> get_level_sigma_sq() const { return level_sigma_sq_; }
    std::vector<float> get_inv_level_sigma_sq() const { return inv_level_sigma_sq_; }

    //! Image pyramid (public upstream: match::stereo reads it)
    std::vector<cv::Mat> image_pyramid_;

    // ---- additions of the MI355X backend (no upstream counterpart; defaults keep upstream's behaviour) ----
    //! the device context (match::stereo reads this extractor's pyramid where it lies, in HBM)
    const ovs_orb* handle() const { return h_; }
    //! the device context behind an image_pyramid_ member (nullptr if `image_pyramid` is not one): lets match::stereo keep
    //! upstream's constructor, which receives the two extractors' image_pyramid_ by reference
    static const ovs_orb* device_context_of(const std::vector<cv::Mat>& image_pyramid);
    //! false: image_pyramid_ levels >= 1 stay on the device (monocular tracking never reads them; saves a 4.3 MB D2H per 1080p frame)
    void set_image_pyramid_download(const bool enable);
    //! HIP device this extractor runs on (before the first extract). Stereo rigs may put left / right on two GPUs.
    //! the device the frame's matcher-side cache belongs on: data::frame's constructor sets device_cache_->device = extractor_->get_device()
    int get_device() const { return device_; }
    void set_device(const int device) {
        if (device != device_) release();
        device_ = device;
    }

private:
    void initialize();
    void release();
    void ensure_handle(int rows, int cols);
    void create_rectangle_mask(const unsigned int cols, const unsigned int rows);
    void register_self();
    void unregister_self();

    orb_params orb_params_;
    std::vector<float> scale_factors_, inv_scale_factors_, level_sigma_sq_, inv_level_sigma_sq_;
    cv::Mat rect_mask_;
    ovs_orb* h_ = nullptr;
    int h_rows_ = 0, h_cols_ = 0;
    int device_ = 0;
    bool download_pyramid_ = true;
    std::vector<uint8_t> desc_buf_;
};

}   // namespace feature
}   // namespace openvslam
