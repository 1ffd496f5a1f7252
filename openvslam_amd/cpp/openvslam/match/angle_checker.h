// match::angle_checker<T> (expected: src/openvslam/match/angle_checker.h), host side: the 30-bin rotation histogram that keeps the
// three fullest bins. The windowed matchers evaluate it inside their device resolver; robust::brute_force_match applies it on the
// host AFTER the device match, which is exact because upstream, too, only removes matches at the very end.
// Rule (oracle/ORACLE_SPEC.md 17): delta wrapped into [0, 360), bin = cvRound(delta * (1 / 30)) (bins are 30 degrees wide: upstream's
// inherited quirk), bin 30 -> 0; the three fullest bins stay, ties resolved towards the lower bin index.
#pragma once
#include <ovslam_hip.h>

#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

namespace openvslam {
namespace match {

template <typename T>
class angle_checker {
public:
    explicit angle_checker(const unsigned int histogram_length = 30, const unsigned int num_bins_thr = 3)
        : histogram_length_(histogram_length), inv_histogram_length_(1.0f / histogram_length), num_bins_thr_(num_bins_thr),
          angle_histogram_(histogram_length) {}

    void append_delta_angle(float delta_angle, const T& match) {
        if (delta_angle < 0.0f) delta_angle += 360.0f;
        if (360.0f <= delta_angle) delta_angle -= 360.0f;
        unsigned int bin = (unsigned int)std::nearbyintf(delta_angle * inv_histogram_length_);   // cvRound: half to even
        if (bin == histogram_length_) bin = 0;
        angle_histogram_.at(bin).push_back(match);
    }

    std::vector<T> get_valid_matches() const { return collect(true); }
    std::vector<T> get_invalid_matches() const { return collect(false); }

private:
    std::vector<T> collect(const bool kept) const {
        std::vector<unsigned int> order(histogram_length_);
        std::iota(order.begin(), order.end(), 0u);
        // equal sizes: the lower bin first, or (ovs_match_set_variant(OVS_MATCH_VARIANT_ANGLE_TIE_ORDER, 1)) the higher one -- upstream's std::sort
        // on the sizes leaves this to the library; the device resolvers read the same switch
        const bool higher_first = ovs_match_get_variant(OVS_MATCH_VARIANT_ANGLE_TIE_ORDER) == 1;
        std::stable_sort(order.begin(), order.end(), [&](unsigned int a, unsigned int b) {
            const size_t sa = angle_histogram_[a].size(), sb = angle_histogram_[b].size();
            return sa > sb || (higher_first && sa == sb && a > b);
        });
        // rule 17's alternative, the same process-wide switch the device resolvers read (ovs_match_set_variant): ORB-SLAM2's 0.1 x max rule
        unsigned int n_keep = num_bins_thr_ < histogram_length_ ? num_bins_thr_ : histogram_length_;
        if (ovs_match_get_variant(OVS_MATCH_VARIANT_ANGLE_KEEP_RULE) == 1 && n_keep == 3) {
            const float max1 = (float)angle_histogram_[order[0]].size();
            if ((float)angle_histogram_[order[1]].size() < 0.1f * max1) n_keep = 1;
            else if ((float)angle_histogram_[order[2]].size() < 0.1f * max1) n_keep = 2;
        }
        std::vector<T> out;
        for (unsigned int bin = 0; bin < histogram_length_; ++bin) {
            bool is_kept = false;
            for (unsigned int k = 0; k < n_keep; ++k) is_kept |= order[k] == bin;
            if (is_kept == kept) out.insert(out.end(), angle_histogram_[bin].begin(), angle_histogram_[bin].end());
        }
        return out;
    }

    const unsigned int histogram_length_;
    const float inv_histogram_length_;
    const unsigned int num_bins_thr_;
    std::vector<std::vector<T>> angle_histogram_;
};

}   // namespace match
}   // namespace openvslam
