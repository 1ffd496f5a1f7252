// ovo_match.cc -- CPU ORACLE (test infrastructure, see ovo_oracle.h): match::base / match::robust restated from spec.
// PARITY UNPINNED (upstream absent). Expected upstream paths: src/openvslam/match/base.h, robust.{h,cc}.
#include "ovo_oracle.h"

#include <cstring>
#include <vector>

namespace {

// M1  match::base::compute_descriptor_distance_32: 8 x u32 XOR + SWAR population count.
inline uint32_t distance_32(const uint8_t* a, const uint8_t* b) {
    uint32_t dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (((v + (v >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24;
    }
    return dist;
}

}   // namespace

extern "C" {

uint32_t ovo_descriptor_distance_32(const uint8_t* a, const uint8_t* b) { return distance_32(a, b); }

// M2  match::robust::brute_force_match: outer loop over keyframe keypoints that hold a live
// landmark, inner loop over ALL frame keypoints not yet claimed by an earlier keyframe keypoint; strict `<`
// keeps the FIRST minimum; accept iff best <= HAMMING_DIST_THR_LOW and !(lowe_ratio*second < best).
int ovo_robust_brute_force_match(const uint8_t* desc_frm, int n_frm, const uint8_t* frm_valid, const uint8_t* desc_kf, int n_kf,
                                 const uint8_t* kf_valid, float lowe_ratio, int32_t* pairs, int cap) {
    int num_matches = 0;
    std::vector<uint8_t> already_matched_1((size_t)(n_frm > 0 ? n_frm : 0), 0);
    // optional frame-side mask (ORACLE_SPEC rule 14): an excluded frame keypoint is skipped like an already matched one
    if (frm_valid)
        for (int i = 0; i < n_frm; ++i) already_matched_1[i] = frm_valid[i] ? 0 : 1;
    for (int idx_2 = 0; idx_2 < n_kf; ++idx_2) {
        if (kf_valid && !kf_valid[idx_2]) continue;
        const uint8_t* desc_2 = desc_kf + (size_t)idx_2 * 32;
        int best_idx_1 = -1;
        unsigned best = OVO_MAX_HAMMING_DIST, second = OVO_MAX_HAMMING_DIST;
        for (int idx_1 = 0; idx_1 < n_frm; ++idx_1) {
            if (already_matched_1[idx_1]) continue;
            const unsigned d = distance_32(desc_2, desc_frm + (size_t)idx_1 * 32);
            if (d < best) {
                second = best;
                best = d;
                best_idx_1 = idx_1;
            } else if (d < second) {
                second = d;
            }
        }
        if (OVO_HAMMING_DIST_THR_LOW < best) continue;
        if (lowe_ratio * second < static_cast<float>(best)) continue;
        if (num_matches < cap) {
            pairs[2 * num_matches] = best_idx_1;
            pairs[2 * num_matches + 1] = idx_2;
        }
        already_matched_1[best_idx_1] = 1;
        ++num_matches;
    }
    return num_matches;
}

int ovo_hamming_best2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint8_t* t_valid, int32_t* best_idx, uint16_t* best,
                      uint16_t* second) {
    for (int i = 0; i < nq; ++i) {
        int bi = -1;
        unsigned b = OVO_MAX_HAMMING_DIST, s = OVO_MAX_HAMMING_DIST;
        for (int j = 0; j < nt; ++j) {
            if (t_valid && !t_valid[j]) continue;
            const unsigned d = distance_32(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < b) { s = b; b = d; bi = j; }
            else if (d < s) s = d;
        }
        best_idx[i] = bi;
        best[i] = (uint16_t)b;
        second[i] = (uint16_t)s;
    }
    return 0;
}

}   // extern "C"
