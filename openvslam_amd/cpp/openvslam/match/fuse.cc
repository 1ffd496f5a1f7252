// match::fuse over the C ABI. Replaces the bodies of fuse::replace_duplication / detect_duplication in src/openvslam/match/fuse.cc.
#include "fuse.h"

#include <cstring>
#include <set>
#include <unordered_set>

#include "window_ctx.h"

namespace openvslam {
namespace match {

namespace {
// the landmarks to check as five flat arrays. Per THREAD and reused from call to call (mapping_module calls this ~40 times per new keyframe with
// ~2000 landmarks each: five fresh 2 - 64 KB vectors per call were a tenth of a resident call's 0.11 ms in allocation, zero-fill and page faults);
// entries of landmarks that are not usable get zeros in their numeric fields (their descriptor bytes keep what an earlier call left); the kernel reads nothing of an entry whose `valid` is 0
struct flat_landmarks {
    std::vector<double> pos, normal;
    std::vector<float> dist;
    std::vector<uint8_t> desc, valid;
    std::vector<int32_t> best;
    template <typename IT, typename PRED>
    void fill(IT begin, IT end, size_t m, PRED is_valid) {
        pos.resize(3 * m);
        normal.resize(3 * m);
        dist.resize(2 * m);
        desc.resize(32 * m);
        valid.resize(m);
        best.assign(m, -1);
        size_t l = 0;
        for (IT it = begin; it != end; ++it, ++l) {
            data::landmark* lm = *it;
            valid[l] = is_valid(lm) ? 1 : 0;
            if (!valid[l]) {   // defined values for an entry the kernel masks out
                for (int a = 0; a < 3; ++a) pos[3 * l + a] = normal[3 * l + a] = 0.0;
                dist[2 * l] = dist[2 * l + 1] = 0.0f;
                continue;
            }
            const Vec3_t p = lm->get_pos_in_world(), n = lm->get_obs_mean_normal();
            for (int a = 0; a < 3; ++a) {
                pos[3 * l + a] = p(a);
                normal[3 * l + a] = n(a);
            }
            dist[2 * l] = lm->min_valid_dist_;   // raw members: the kernel widens the gate as the getters do and predicts the level from the raw maximum
            dist[2 * l + 1] = lm->max_valid_dist_;
            const cv::Mat d = lm->get_descriptor();
            std::memcpy(&desc[32 * l], d.data, 32);
        }
    }
};
flat_landmarks& flat_scratch() {
    thread_local flat_landmarks f;
    return f;
}
}   // namespace

template <typename T>
unsigned int fuse::replace_duplication(data::keyframe* keyfrm, const T& landmarks_to_check, const float margin) const {
    const int n = (int)keyfrm->num_keypts_, m = (int)landmarks_to_check.size();
    if (n == 0 || m == 0) return 0;
    auto usable = [&](data::landmark* lm) { return lm && !lm->will_be_erased() && !lm->is_observed_in_keyframe(keyfrm); };
    flat_landmarks& f = flat_scratch();
    f.fill(landmarks_to_check.begin(), landmarks_to_check.end(), (size_t)m, usable);
    std::vector<int32_t>& best = f.best;
    const ovs_camera cam = detail::camera_of(keyfrm->camera_);
    double pose[12];
    detail::pose12(keyfrm->get_cam_pose(), pose);
    int32_t num = 0;
    const int device = detail::device_of(*keyfrm);
    // the keyframe is resident: mapping_module::fuse_landmark_duplication calls this on ~20 covisible keyframes per new keyframe, and again with
    // the roles swapped -- only the landmarks to check travel
    if (!detail::guarded("ovs_fuse_replace_duplication_f", [&] {
            const auto h = detail::device_handle_of(*keyfrm);
            return ovs_fuse_replace_duplication_f(detail::window_ctx(device).get(n, m), &cam, detail::dev(h), pose, f.pos.data(), f.dist.data(), f.normal.data(),
                                                  f.desc.data(), f.valid.data(), m, keyfrm->scale_factors_.data(), keyfrm->inv_level_sigma_sq_.data(),
                                                  (int)keyfrm->scale_factors_.size(), keyfrm->log_scale_factor_, margin, best.data(), &num);
        }, {keyfrm->device_cache_.get()}, device)) {
        return 0;
    }
    // upstream's write-back, in the order of landmarks_to_check; an earlier replacement can erase or attach a later landmark
    unsigned int num_fused = 0;
    int l = 0;
    for (auto it = landmarks_to_check.begin(); it != landmarks_to_check.end(); ++it, ++l) {
        data::landmark* lm = *it;
        if (best[l] < 0 || !usable(lm)) continue;
        auto* lm_in_keyfrm = keyfrm->get_landmark((unsigned)best[l]);
        if (lm_in_keyfrm) {
            if (!lm_in_keyfrm->will_be_erased()) {
                if (lm->num_observations() < lm_in_keyfrm->num_observations())   // keep the more reliable one
                    lm->replace(lm_in_keyfrm);
                else
                    lm_in_keyfrm->replace(lm);
            }
        } else {
            lm->add_observation(keyfrm, (unsigned)best[l]);
            keyfrm->add_landmark(lm, (unsigned)best[l]);
        }
        ++num_fused;
    }
    return num_fused;
}

// the two instantiations upstream uses (mapping_module::fuse_landmark_duplication)
template unsigned int fuse::replace_duplication(data::keyframe*, const std::vector<data::landmark*>&, const float) const;
template unsigned int fuse::replace_duplication(data::keyframe*, const std::unordered_set<data::landmark*>&, const float) const;

unsigned int fuse::detect_duplication(data::keyframe* keyfrm, const Mat44_t& Sim3_cw, const std::vector<data::landmark*>& landmarks_to_check,
                                      const float margin, std::vector<data::landmark*>& duplicated_lms_in_keyfrm) const {
    const int n = (int)keyfrm->num_keypts_, m = (int)landmarks_to_check.size();
    duplicated_lms_in_keyfrm = std::vector<data::landmark*>((size_t)m, nullptr);
    if (n == 0 || m == 0) return 0;
    const auto valid_lms = keyfrm->get_landmarks();
    const std::set<data::landmark*> already_matched(valid_lms.begin(), valid_lms.end());
    auto usable = [&](data::landmark* lm) { return lm && !lm->will_be_erased() && !already_matched.count(lm); };
    flat_landmarks& f = flat_scratch();
    f.fill(landmarks_to_check.begin(), landmarks_to_check.end(), (size_t)m, usable);
    std::vector<int32_t>& best = f.best;
    const ovs_camera cam = detail::camera_of(keyfrm->camera_);
    double sim3[12];
    detail::pose12(Sim3_cw, sim3);
    int32_t num = 0;
    const int device = detail::device_of(*keyfrm);
    if (!detail::guarded("ovs_fuse_detect_duplication_f", [&] {
            const auto h = detail::device_handle_of(*keyfrm);
            return ovs_fuse_detect_duplication_f(detail::window_ctx(device).get(n, m), &cam, detail::dev(h), sim3, f.pos.data(), f.dist.data(), f.normal.data(),
                                                 f.desc.data(), f.valid.data(), m, keyfrm->scale_factors_.data(), (int)keyfrm->scale_factors_.size(),
                                                 keyfrm->log_scale_factor_, margin, best.data(), &num);
        }, {keyfrm->device_cache_.get()}, device)) {
        return 0;
    }
    unsigned int num_fused = 0;
    for (int l = 0; l < m; ++l) {
        if (best[l] < 0) continue;
        data::landmark* lm = landmarks_to_check[l];
        auto* lm_in_keyfrm = keyfrm->get_landmark((unsigned)best[l]);
        if (lm_in_keyfrm) {
            if (!lm_in_keyfrm->will_be_erased()) duplicated_lms_in_keyfrm[l] = lm_in_keyfrm;   // replaced by the caller under the map mutex
        } else {
            lm->add_observation(keyfrm, (unsigned)best[l]);
            keyfrm->add_landmark(lm, (unsigned)best[l]);
        }
        ++num_fused;
    }
    return num_fused;
}

}   // namespace match
}   // namespace openvslam
