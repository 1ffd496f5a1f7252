// match::stereo::compute over the C ABI. Replaces that function's body in src/openvslam/match/stereo.cc.
#include "stereo.h"

#include "window_ctx.h"

namespace openvslam {
namespace match {

namespace {
struct stereo_holder {
    ovs_stereo* s = nullptr;
    int device = 0;   // the device of the extractors whose pyramids are read (match::stereo runs where they live)
    int cap_rows = 0, cap_kps = 0;
    ~stereo_holder() {
        if (s) ovs_stereo_destroy(s);
    }
    ovs_stereo* get(int rows, int kps, int dev) {
        if (s && rows <= cap_rows && kps <= cap_kps && dev == device) return s;
        device = dev;
        if (s) ovs_stereo_destroy(s);
        s = nullptr;
        cap_rows = rows < 2160 ? 2160 : rows;
        cap_kps = kps < 8192 ? 8192 : kps;
        const int st = ovs_stereo_create(cap_rows, cap_kps, device, &s);
        if (st != OVS_OK) {
            s = nullptr;
            cap_rows = cap_kps = 0;
            throw util::device_error(st, std::string("ovs_stereo_create: ") + ovs_last_error());   // caught by run_guarded
        }
        return s;
    }
    void reset() {
        if (s) ovs_stereo_destroy(s);
        s = nullptr;
        cap_rows = cap_kps = 0;
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
    if (!left || !right) {
        // the pyramids must be the image_pyramid_ members of two feature::orb_extractor objects that have extracted (their pixels are read
        // on the device); an extractor that lost its device context (util/device_policy.h) has none: no stereo matches for this frame
        ++util::device_failures().failed_calls;
        ++util::device_failures().degraded;
        util::detail::log_failure("match::stereo::compute", OVS_ERR_INVALID, "an image pyramid has no device context", "returning the empty result");
        return;
    }
    const int rows0 = left_image_pyramid_.at(0).rows;
    // failure policy (util/device_policy.h): one retry on a fresh context, then "no keypoint has a stereo match" (-1 everywhere), which
    // is what upstream's loop leaves for a frame without matches
    if (!util::run_guarded(
            "ovs_stereo_compute",
            [&] {
                return ovs_stereo_compute(g_stereo.get(rows0, n_left > n_right ? n_left : n_right, ovs_orb_device(left)), left, right,
                                          reinterpret_cast<const ovs_keypoint*>(keypts_left_.data()), descs_left_.data, n_left,
                                          reinterpret_cast<const ovs_keypoint*>(keypts_right_.data()), descs_right_.data, n_right, focal_x_baseline_,
                                          true_baseline_, stereo_x_right.data(), depths.data(), nullptr);
            },
            [] { g_stereo.reset(); })) {
        stereo_x_right.assign((size_t)n_left, -1.0f);
        depths.assign((size_t)n_left, -1.0f);
    }
}

}   // namespace match
}   // namespace openvslam
