// Stand-in for solve::essential_solver (expected: src/openvslam/solve/essential_solver.{h,cc}) -- HOST code that is NOT part of the
// MI355X hot path: robust::match_frame_and_keyframe calls it after the device brute-force match, exactly as upstream does, and in an
// OpenVSLAM checkout upstream's own solver is used unchanged. This header only exists so the shim tree compiles and can be tested
// without Eigen: same interface (ctor from two bearing vectors + matches, find_via_ransac, solution_is_valid, get_inlier_matches,
// get_best_E_21), eight-point estimate by a cyclic-Jacobi eigen-decomposition, upstream's inlier rule (angle between a bearing and
// the epipolar plane of its partner below 1 degree, both directions), a deterministic sampler instead of upstream's random_device.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

#include "../data/frame_stub.h"

namespace openvslam {
namespace solve {

class essential_solver {
public:
    essential_solver(const std::vector<Vec3_t>& bearings_1, const std::vector<Vec3_t>& bearings_2, const std::vector<std::pair<int, int>>& matches_12)
        : bearings_1_(bearings_1), bearings_2_(bearings_2), matches_12_(matches_12) {}

    void find_via_ransac(const unsigned int max_num_iter, const bool recompute = true) {
        const unsigned int num_matches = (unsigned int)matches_12_.size();
        solution_is_valid_ = false;
        is_inlier_match_.assign(num_matches, false);
        if (num_matches < min_set_size_) return;
        best_score_ = 0.0;
        uint64_t rng = 0x9E3779B97F4A7C15ull;
        for (unsigned int iter = 0; iter < max_num_iter; ++iter) {
            std::vector<unsigned int> pick;
            while (pick.size() < min_set_size_) {   // sample without replacement (splitmix-style generator: reproducible)
                rng += 0x9E3779B97F4A7C15ull;
                uint64_t z = rng;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                const unsigned int k = (unsigned int)((z ^ (z >> 31)) % num_matches);
                bool dup = false;
                for (auto p : pick) dup |= p == k;
                if (!dup) pick.push_back(k);
            }
            const Mat33_t E_21 = compute_E_21(pick);
            std::vector<bool> inl;
            const double score = check_inliers(E_21, inl);
            if (best_score_ < score) {
                best_score_ = score;
                best_E_21_ = E_21;
                is_inlier_match_ = inl;
            }
        }
        unsigned int num_inliers = 0;
        for (const bool b : is_inlier_match_) num_inliers += b;
        solution_is_valid_ = best_score_ > 0.0 && num_inliers >= min_set_size_;
        if (!solution_is_valid_ || !recompute) return;
        std::vector<unsigned int> all;
        for (unsigned int i = 0; i < num_matches; ++i)
            if (is_inlier_match_[i]) all.push_back(i);
        best_E_21_ = compute_E_21(all);
        best_score_ = check_inliers(best_E_21_, is_inlier_match_);
    }

    bool solution_is_valid() const { return solution_is_valid_; }
    double get_best_score() const { return best_score_; }
    Mat33_t get_best_E_21() const { return best_E_21_; }
    std::vector<bool> get_inlier_matches() const { return is_inlier_match_; }

private:
    // null vector of the stacked epipolar constraints b2^T E b1 = 0 (smallest eigenvector of A^T A, cyclic Jacobi)
    Mat33_t compute_E_21(const std::vector<unsigned int>& idx) const {
        double M[81] = {0};
        for (const unsigned int k : idx) {
            const Vec3_t& b1 = bearings_1_.at((size_t)matches_12_[k].first);
            const Vec3_t& b2 = bearings_2_.at((size_t)matches_12_[k].second);
            double a[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) a[3 * r + c] = b2(r) * b1(c);
            for (int i = 0; i < 9; ++i)
                for (int j = 0; j < 9; ++j) M[9 * i + j] += a[i] * a[j];
        }
        double V[81] = {0};
        for (int i = 0; i < 9; ++i) V[10 * i] = 1.0;
        for (int sweep = 0; sweep < 60; ++sweep) {
            double off = 0;
            for (int p = 0; p < 9; ++p)
                for (int q = p + 1; q < 9; ++q) off += M[9 * p + q] * M[9 * p + q];
            if (off < 1e-30) break;
            for (int p = 0; p < 9; ++p)
                for (int q = p + 1; q < 9; ++q) {
                    if (std::fabs(M[9 * p + q]) < 1e-300) continue;
                    const double theta = (M[9 * q + q] - M[9 * p + p]) / (2.0 * M[9 * p + q]);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                    const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                    for (int k = 0; k < 9; ++k) {
                        const double mkp = M[9 * k + p], mkq = M[9 * k + q];
                        M[9 * k + p] = c * mkp - s * mkq;
                        M[9 * k + q] = s * mkp + c * mkq;
                    }
                    for (int k = 0; k < 9; ++k) {
                        const double mpk = M[9 * p + k], mqk = M[9 * q + k];
                        M[9 * p + k] = c * mpk - s * mqk;
                        M[9 * q + k] = s * mpk + c * mqk;
                    }
                    for (int k = 0; k < 9; ++k) {
                        const double vkp = V[9 * k + p], vkq = V[9 * k + q];
                        V[9 * k + p] = c * vkp - s * vkq;
                        V[9 * k + q] = s * vkp + c * vkq;
                    }
                }
        }
        int best = 0;
        for (int i = 1; i < 9; ++i)
            if (M[10 * i] < M[10 * best]) best = i;
        Mat33_t E;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) E(r, c) = V[9 * (3 * r + c) + best];
        return E;
    }

    double check_inliers(const Mat33_t& E_21, std::vector<bool>& is_inlier) const {
        const unsigned int num_matches = (unsigned int)matches_12_.size();
        is_inlier.assign(num_matches, false);
        const double residual_cos_thr = std::cos(M_PI / 2.0 - 0.01745240643728351);   // 1 degree
        double score = 0.0;
        for (unsigned int i = 0; i < num_matches; ++i) {
            const Vec3_t& b1 = bearings_1_.at((size_t)matches_12_[i].first);
            const Vec3_t& b2 = bearings_2_.at((size_t)matches_12_[i].second);
            double plane_2[3], plane_1[3];   // E_21 b1 and E_21^T b2
            for (int r = 0; r < 3; ++r) {
                plane_2[r] = (E_21(r, 0) * b1(0) + E_21(r, 1) * b1(1)) + E_21(r, 2) * b1(2);
                plane_1[r] = (E_21(0, r) * b2(0) + E_21(1, r) * b2(1)) + E_21(2, r) * b2(2);
            }
            const double n2 = std::sqrt((plane_2[0] * plane_2[0] + plane_2[1] * plane_2[1]) + plane_2[2] * plane_2[2]);
            const double n1 = std::sqrt((plane_1[0] * plane_1[0] + plane_1[1] * plane_1[1]) + plane_1[2] * plane_1[2]);
            if (!(n1 > 0) || !(n2 > 0)) continue;
            const double r2 = std::fabs(((plane_2[0] * b2(0) + plane_2[1] * b2(1)) + plane_2[2] * b2(2)) / n2);
            const double r1 = std::fabs(((plane_1[0] * b1(0) + plane_1[1] * b1(1)) + plane_1[2] * b1(2)) / n1);
            if (residual_cos_thr < r2 || residual_cos_thr < r1) continue;
            is_inlier[i] = true;
            score += (residual_cos_thr - r2) * (residual_cos_thr - r2) + (residual_cos_thr - r1) * (residual_cos_thr - r1);
        }
        return score;
    }

    const std::vector<Vec3_t>& bearings_1_;
    const std::vector<Vec3_t>& bearings_2_;
    const std::vector<std::pair<int, int>>& matches_12_;
    static constexpr unsigned int min_set_size_ = 8;
    bool solution_is_valid_ = false;
    double best_score_ = 0.0;
    Mat33_t best_E_21_;
    std::vector<bool> is_inlier_match_;
};

}   // namespace solve
}   // namespace openvslam
