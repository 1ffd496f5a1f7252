// match::area::match_in_consistent_area over the C ABI. Replaces that function's body in src/openvslam/match/area.cc.
#include "area.h"

#include "window_ctx.h"

namespace openvslam {
namespace match {

unsigned int area::match_in_consistent_area(data::frame& frm_1, data::frame& frm_2, std::vector<cv::Point2f>& prev_matched_pts,
                                            std::vector<int>& matched_indices_2_in_frm_1, int margin) {
    const int n1 = (int)frm_1.undist_keypts_.size(), n2 = (int)frm_2.undist_keypts_.size();
    matched_indices_2_in_frm_1 = std::vector<int>((size_t)n1, -1);
    if (n1 == 0 || n2 == 0) return 0;
    static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f is two floats");
    static_assert(sizeof(int) == sizeof(int32_t), "int is 32 bit");
    int32_t num_matches = 0;
    const std::vector<cv::Point2f> prev_in = prev_matched_pts;   // in / out: restored if the device call fails half way
    // both frames resident: module::initializer matches its init frame against every incoming frame until the map is created
    const int device = detail::device_of(frm_2);
    if (!detail::guarded("ovs_area_match_in_consistent_area_f", [&] {
            const auto h1 = detail::device_handle_on(frm_1, device), h2 = detail::device_handle_of(frm_2);
            return ovs_area_match_in_consistent_area_f(detail::window_ctx(device).get(n2, n1), detail::dev(h1), detail::dev(h2),
                                                      reinterpret_cast<float*>(prev_matched_pts.data()), matched_indices_2_in_frm_1.data(), margin,
                                                      lowe_ratio_, check_orientation_ ? 1 : 0, &num_matches);
        }, {frm_1.device_cache_.get(), frm_2.device_cache_.get()}, device)) {
        matched_indices_2_in_frm_1.assign((size_t)n1, -1);
        prev_matched_pts = prev_in;
        return 0;
    }
    return (unsigned int)num_matches;
}

}   // namespace match
}   // namespace openvslam
