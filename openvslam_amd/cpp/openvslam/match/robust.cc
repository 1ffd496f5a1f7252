// match::robust::brute_force_match over the C ABI. Replaces that function's body in src/openvslam/match/robust.cc.
#include "robust.h"

#include <cmath>

#include "../solve/essential_solver.h"
#include "angle_checker.h"
#include "window_ctx.h"

#include <ovslam_hip.h>

#include <stdexcept>
#include <string>

namespace openvslam {
namespace match {

namespace {
// upstream stack-constructs a matcher per call; the device context is kept per thread so a call does not allocate
struct matcher_holder {
    ovs_matcher* m = nullptr;
    int device = 0;
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
        const int st = ovs_matcher_create(cap1, cap2, 1, device, &m);
        if (st != OVS_OK) {
            m = nullptr;
            cap1 = cap2 = 0;
            throw util::device_error(st, std::string("ovs_matcher_create: ") + ovs_last_error());   // caught by run_guarded
        }
        return m;
    }
    void reset() {
        if (m) ovs_matcher_destroy(m);
        m = nullptr;
        cap1 = cap2 = 0;
    }
};
// one context per (thread, device): the frame's device decides
matcher_holder& matcher_ctx(int device) {
    thread_local std::vector<std::unique_ptr<matcher_holder>> per_device;
    if (device < 0) device = 0;
    if ((size_t)device >= per_device.size()) per_device.resize((size_t)device + 1);
    if (!per_device[(size_t)device]) {
        per_device[(size_t)device] = std::make_unique<matcher_holder>();
        per_device[(size_t)device]->device = device;
    }
    return *per_device[(size_t)device];
}
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
    // rule 14's frame-side test (ovs_match_set_variant(OVS_MATCH_VARIANT_BF_FRAME_MASK, 1)): frame keypoints that already own a landmark are
    // skipped like already matched ones; default: no such test (upstream as recalled overwrites curr_frm.landmarks_ wholesale afterwards)
    std::vector<uint8_t> valid_1;
    if (ovs_match_get_variant(OVS_MATCH_VARIANT_BF_FRAME_MASK) == 1) {
        valid_1.assign(num_keypts_1, 1);
        for (unsigned int idx_1 = 0; idx_1 < num_keypts_1 && idx_1 < frm.landmarks_.size(); ++idx_1)
            if (frm.landmarks_[idx_1]) valid_1[idx_1] = 0;
    }
    std::vector<int32_t> pairs((size_t)2 * num_keypts_2);
    int n = 0;
    matches.clear();
    // failure policy (util/device_policy.h): one retry on a fresh matcher context, then zero matches
    const int device = detail::device_of(frm);
    if (!util::run_guarded(
            "ovs_robust_brute_force_match",
            [&] {
                return ovs_robust_brute_force_match(matcher_ctx(device).get(num_keypts_1, num_keypts_2), frm.descriptors_.data, (int)num_keypts_1,
                                                    /*valid_1: by default upstream's inner loop skips only already matched idx_1*/ valid_1.empty() ? nullptr : valid_1.data(),
                                                    keyfrm->descriptors_.data, (int)num_keypts_2, valid.data(), lowe_ratio_, pairs.data(),
                                                    (int)num_keypts_2, &n);
            },
            [device] { matcher_ctx(device).reset(); }))
        return 0;
    matches.reserve((size_t)n);
    for (int i = 0; i < n; ++i) matches.emplace_back(std::make_pair(pairs[2 * i], pairs[2 * i + 1]));
    unsigned int num_matches = (unsigned int)n;
    if (check_orientation_) {
        // upstream fills the histogram while matching and erases the invalid entries at the end: a pure post-filter
        angle_checker<int> angle_checker;
        for (const auto& m : matches)
            angle_checker.append_delta_angle(frm.keypts_.at((size_t)m.first).angle - keyfrm->keypts_.at((size_t)m.second).angle, m.first);
        for (const auto invalid_idx_1 : angle_checker.get_invalid_matches()) {
            for (auto itr = matches.begin(); itr != matches.end(); ++itr) {
                if (itr->first == invalid_idx_1) {
                    matches.erase(itr);
                    --num_matches;
                    break;
                }
            }
        }
    }
    return num_matches;
}

unsigned int robust::match_frame_and_keyframe(data::frame& frm, data::keyframe* keyfrm, std::vector<data::landmark*>& matched_lms_in_frm) {
    // initialisation
    const auto num_frm_keypts = frm.num_keypts_;
    const auto keyfrm_lms = keyfrm->get_landmarks();
    unsigned int num_inlier_matches = 0;
    matched_lms_in_frm = std::vector<data::landmark*>(num_frm_keypts, nullptr);

    // brute-force match on the device
    std::vector<std::pair<int, int>> matches;
    brute_force_match(frm, keyfrm, matches);

    // eight-point RANSAC on the bearings keeps only the inliers (upstream: solve::essential_solver, 50 iterations, no recompute)
    solve::essential_solver solver(frm.bearings_, keyfrm->bearings_, matches);
    solver.find_via_ransac(50, false);
    if (!solver.solution_is_valid()) return 0;
    const auto is_inlier_matches = solver.get_inlier_matches();

    for (unsigned int i = 0; i < matches.size(); ++i) {
        if (!is_inlier_matches.at(i)) continue;
        const auto frm_idx = matches.at(i).first;
        const auto keyfrm_idx = matches.at(i).second;
        matched_lms_in_frm.at((size_t)frm_idx) = keyfrm_lms.at((size_t)keyfrm_idx);
        ++num_inlier_matches;
    }
    return num_inlier_matches;
}

unsigned int robust::match_for_triangulation(data::keyframe* keyfrm_1, data::keyframe* keyfrm_2, const Mat33_t& E_12,
                                             std::vector<std::pair<unsigned int, unsigned int>>& matched_idx_pairs) const {
    matched_idx_pairs.clear();
    const int n1 = (int)keyfrm_1->num_keypts_, n2 = (int)keyfrm_2->num_keypts_;
    if (n1 == 0 || n2 == 0) return 0;
    // keyfrm_1's camera centre seen from keyfrm_2, as a bearing (upstream: camera->reproject_to_bearing of rot_2w * center_1 + trans_2w)
    const Vec3_t c1 = keyfrm_1->get_cam_center();
    const Mat33_t rot_2w = keyfrm_2->get_rotation();
    const Vec3_t trans_2w = keyfrm_2->get_translation();
    double epipole[3];
    for (int i = 0; i < 3; ++i) epipole[i] = ((rot_2w(i, 0) * c1(0) + rot_2w(i, 1) * c1(1)) + rot_2w(i, 2) * c1(2)) + trans_2w(i);
    const double norm = std::sqrt((epipole[0] * epipole[0] + epipole[1] * epipole[1]) + epipole[2] * epipole[2]);
    for (int i = 0; i < 3; ++i) epipole[i] /= norm;
    auto flatten_kf = [](const data::keyframe* kf, int n, std::vector<uint8_t>& has_lm) {
        has_lm.resize((size_t)n);
        for (int i = 0; i < n; ++i) has_lm[i] = kf->get_landmark((unsigned)i) != nullptr;
    };
    std::vector<uint8_t> has_1, has_2;
    flatten_kf(keyfrm_1, n1, has_1);
    flatten_kf(keyfrm_2, n2, has_2);
    std::vector<int32_t> id1, st1, it1, id2, st2, it2;
    detail::flatten_bow(keyfrm_1->bow_feat_vec_, id1, st1, it1);
    detail::flatten_bow(keyfrm_2->bow_feat_vec_, id2, st2, it2);
    double E[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) E[3 * i + j] = E_12(i, j);
    std::vector<int32_t> matched((size_t)n1, -1);
    int32_t num_matches = 0;
    // both keyframes resident, bearings included (attached to a handle the first time a triangulation asks for them): mapping_module::
    // create_new_landmarks calls this for ~10-20 covisible keyframes per new keyframe
    const int device = detail::device_of(*keyfrm_1);
    if (!detail::guarded("ovs_robust_match_for_triangulation_f", [&] {
            const auto h1 = detail::device_handle_of(*keyfrm_1, true), h2 = detail::device_handle_on(*keyfrm_2, device, true);
            return ovs_robust_match_for_triangulation_f(detail::window_ctx(device).get(n2, n1), detail::dev(h1), has_1.data(), id1.data(), st1.data(),
                                                       it1.data(), (int)id1.size(), detail::dev(h2), has_2.data(), id2.data(), st2.data(), it2.data(),
                                                       (int)id2.size(), E, epipole, keyfrm_1->scale_factors_.data(),
                                                       (int)keyfrm_1->scale_factors_.size(), check_orientation_ ? 1 : 0, matched.data(), &num_matches);
        }, {keyfrm_1->device_cache_.get(), keyfrm_2->device_cache_.get()}, device)) {
        return 0;
    }
    matched_idx_pairs.reserve((size_t)num_matches);
    for (int i = 0; i < n1; ++i)
        if (matched[i] >= 0) matched_idx_pairs.emplace_back((unsigned)i, (unsigned)matched[i]);
    return (unsigned int)matched_idx_pairs.size();
}

}   // namespace match
}   // namespace openvslam
