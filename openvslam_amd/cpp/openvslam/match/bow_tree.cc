// match::bow_tree::match_frame_and_keyframe over the C ABI. Replaces that function's body in src/openvslam/match/bow_tree.cc.
#include "bow_tree.h"

#include "window_ctx.h"

namespace openvslam {
namespace match {

using detail::flatten_bow;

unsigned int bow_tree::match_frame_and_keyframe(data::keyframe* keyfrm, data::frame& frm, std::vector<data::landmark*>& matched_lms_in_frm) const {
    const int n_kf = (int)keyfrm->num_keypts_, n_frm = (int)frm.num_keypts_;
    matched_lms_in_frm = std::vector<data::landmark*>((size_t)n_frm, nullptr);
    if (n_kf == 0 || n_frm == 0) return 0;
    const auto keyfrm_lms = keyfrm->get_landmarks();
    std::vector<uint8_t> valid((size_t)n_kf);
    for (int i = 0; i < n_kf; ++i) valid[i] = keyfrm_lms[i] && !keyfrm_lms[i]->will_be_erased();
    std::vector<int32_t> kid, kst, kit, fid, fst, fit;
    flatten_bow(keyfrm->bow_feat_vec_, kid, kst, kit);
    flatten_bow(frm.bow_feat_vec_, fid, fst, fit);
    std::vector<int32_t> matched((size_t)n_frm, -1);
    int32_t num_matches = 0;
    // both sides resident (the handles hold undist_keypts_: the matcher reads angles only, which undistortion does not change); only the
    // keyframe's landmark flags and the two BoW feature vectors travel
    const int device = detail::device_of(frm);
    if (!detail::guarded("ovs_bow_match_frame_and_keyframe_f", [&] {
            const auto hk = detail::device_handle_on(*keyfrm, device), hf = detail::device_handle_of(frm);
            return ovs_bow_match_frame_and_keyframe_f(detail::window_ctx(device).get(n_frm, n_kf), detail::dev(hk), valid.data(), kid.data(), kst.data(),
                                                     kit.data(), (int)kid.size(), detail::dev(hf), fid.data(), fst.data(), fit.data(), (int)fid.size(),
                                                     lowe_ratio_, check_orientation_ ? 1 : 0, matched.data(), &num_matches);
        }, {keyfrm->device_cache_.get(), frm.device_cache_.get()}, device)) {
        return 0;
    }
    for (int j = 0; j < n_frm; ++j)
        if (matched[j] >= 0) matched_lms_in_frm[j] = keyfrm_lms[matched[j]];
    return (unsigned int)num_matches;
}

unsigned int bow_tree::match_keyframes(data::keyframe* keyfrm_1, data::keyframe* keyfrm_2, std::vector<data::landmark*>& matched_lms_in_keyfrm_1) const {
    const int n1 = (int)keyfrm_1->num_keypts_, n2 = (int)keyfrm_2->num_keypts_;
    matched_lms_in_keyfrm_1 = std::vector<data::landmark*>((size_t)n1, nullptr);
    if (n1 == 0 || n2 == 0) return 0;
    const auto lms_1 = keyfrm_1->get_landmarks(), lms_2 = keyfrm_2->get_landmarks();
    std::vector<uint8_t> v1((size_t)n1), v2((size_t)n2);
    for (int i = 0; i < n1; ++i) v1[i] = lms_1[i] && !lms_1[i]->will_be_erased();
    for (int i = 0; i < n2; ++i) v2[i] = lms_2[i] && !lms_2[i]->will_be_erased();
    std::vector<int32_t> id1, st1, it1, id2, st2, it2;
    flatten_bow(keyfrm_1->bow_feat_vec_, id1, st1, it1);
    flatten_bow(keyfrm_2->bow_feat_vec_, id2, st2, it2);
    std::vector<int32_t> matched((size_t)n1, -1);
    int32_t num_matches = 0;
    const int device = detail::device_of(*keyfrm_1);
    if (!detail::guarded("ovs_bow_match_keyframes_f", [&] {
            const auto h1 = detail::device_handle_of(*keyfrm_1), h2 = detail::device_handle_on(*keyfrm_2, device);
            return ovs_bow_match_keyframes_f(detail::window_ctx(device).get(n2, n1), detail::dev(h1), v1.data(), id1.data(), st1.data(), it1.data(),
                                            (int)id1.size(), detail::dev(h2), v2.data(), id2.data(), st2.data(), it2.data(), (int)id2.size(), lowe_ratio_,
                                            check_orientation_ ? 1 : 0, matched.data(), &num_matches);
        }, {keyfrm_1->device_cache_.get(), keyfrm_2->device_cache_.get()}, device)) {
        return 0;
    }
    for (int i = 0; i < n1; ++i)
        if (matched[i] >= 0) matched_lms_in_keyfrm_1[(size_t)i] = lms_2[(size_t)matched[i]];
    return (unsigned int)num_matches;
}

}   // namespace match
}   // namespace openvslam
