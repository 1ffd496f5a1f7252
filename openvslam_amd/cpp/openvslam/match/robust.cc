// match::robust::brute_force_match over the C ABI. Replaces that function's body in src/openvslam/match/robust.cc.
#include "robust.h"

#include <ovslam_hip.h>

#include <stdexcept>
#include <string>

namespace openvslam {
namespace match {

namespace {
// upstream stack-constructs a matcher per call; the device context is kept per thread so a call does not allocate
struct matcher_holder {
    ovs_matcher* m = nullptr;
    int cap1 = 0, cap2 = 0;
    ~matcher_holder() {
        if (m) ovs_matcher_destroy(m);
    }
    ovs_matcher* get(int n1, int n2) {
        if (m && n1 <= cap1 && n2 <= cap2) return m;
        if (m) ovs_matcher_destroy(m);
        m = nullptr;
        cap1 = n1 < 4096 ? 4096 : n1;
        cap2 = n2 < 4096 ? 4096 : n2;
        const int st = ovs_matcher_create(cap1, cap2, 1, 0, &m);
        if (st != OVS_OK) throw std::runtime_error(std::string("ovs_matcher_create failed: ") + ovs_last_error());
        return m;
    }
};
thread_local matcher_holder g_matcher;
}   // namespace

unsigned int robust::brute_force_match(data::frame& frm, data::keyframe* keyfrm, std::vector<std::pair<int, int>>& matches) const {
    const auto num_keypts_1 = frm.num_keypts_;
    const auto num_keypts_2 = keyfrm->num_keypts_;
    if (num_keypts_1 == 0 || num_keypts_2 == 0) return 0;
    const auto lms_2 = keyfrm->get_landmarks();
    // select only those keyframe keypoints that are associated with 3D points
    std::vector<uint8_t> valid(num_keypts_2);
    for (unsigned int idx_2 = 0; idx_2 < num_keypts_2; ++idx_2) {
        const auto lm_2 = idx_2 < lms_2.size() ? lms_2[idx_2] : nullptr;
        valid[idx_2] = lm_2 && !lm_2->will_be_erased();
    }
    std::vector<int32_t> pairs((size_t)2 * num_keypts_2);
    int n = 0;
    const int st = ovs_robust_brute_force_match(g_matcher.get(num_keypts_1, num_keypts_2), frm.descriptors_.data, (int)num_keypts_1,
                                                keyfrm->descriptors_.data, (int)num_keypts_2, valid.data(), lowe_ratio_, pairs.data(),
                                                (int)num_keypts_2, &n);
    if (st != OVS_OK) throw std::runtime_error(std::string("ovs_robust_brute_force_match failed: ") + ovs_last_error());
    for (int i = 0; i < n; ++i) matches.emplace_back(std::make_pair(pairs[2 * i], pairs[2 * i + 1]));
    return (unsigned int)n;
}

}   // namespace match
}   // namespace openvslam
