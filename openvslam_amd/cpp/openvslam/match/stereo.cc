// match::stereo::compute over the C ABI. Replaces that function's body in src/openvslam/match/stereo.cc.
#include "stereo.h"

#include "window_ctx.h"

namespace openvslam {
namespace match {

namespace {
struct stereo_holder {
    ovs_stereo* s = nullptr;
    int cap_rows = 0, cap_kps = 0;
    ~stereo_holder() {
        if (s) ovs_stereo_destroy(s);
    }
    ovs_stereo* get(int rows, int kps) {
        if (s && rows <= cap_rows && kps <= cap_kps) return s;
        if (s) ovs_stereo_destroy(s);
        s = nullptr;
        cap_rows = rows < 2160 ? 2160 : rows;
        cap_kps = kps < 8192 ? 8192 : kps;
        detail::check(ovs_stereo_create(cap_rows, cap_kps, 0, &s), "ovs_stereo_create");
        return s;
    }
};
thread_local stereo_holder g_stereo;
}   // namespace

void stereo::compute(std::vector<float>& stereo_x_right, std::vector<float>& depths) const {
    const int n_left = (int)keypts_left_.size(), n_right = (int)keypts_right_.size();
    stereo_x_right.assign((size_t)n_left, -1.0f);
    depths.assign((size_t)n_left, -1.0f);
    if (n_left == 0 || n_right == 0) return;
    const ovs_orb* left = feature::orb_extractor::device_context_of(left_image_pyramid_);
    const ovs_orb* right = feature::orb_extractor::device_context_of(right_image_pyramid_);
    if (!left || !right)
        throw std::runtime_error("match::stereo: the image pyramids must be the image_pyramid_ members of two feature::orb_extractor objects "
                                 "that have extracted (their pixels are read on the device)");
    const int rows0 = left_image_pyramid_.at(0).rows;
    detail::check(ovs_stereo_compute(g_stereo.get(rows0, n_left > n_right ? n_left : n_right), left, right,
                                     reinterpret_cast<const ovs_keypoint*>(keypts_left_.data()), descs_left_.data, n_left,
                                     reinterpret_cast<const ovs_keypoint*>(keypts_right_.data()), descs_right_.data, n_right, focal_x_baseline_,
                                     true_baseline_, stereo_x_right.data(), depths.data(), nullptr),
                  "ovs_stereo_compute");
}

}   // namespace match
}   // namespace openvslam
