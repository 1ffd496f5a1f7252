// match::base (expected: src/openvslam/match/base.h): thresholds and the host-side 256-bit Hamming distance.
#pragma once
#include <cstdint>
#include <cstring>

#include "../../cv_stub.h"

namespace openvslam {
namespace match {

static constexpr unsigned int HAMMING_DIST_THR_LOW = 50;
static constexpr unsigned int HAMMING_DIST_THR_HIGH = 100;
static constexpr unsigned int MAX_HAMMING_DIST = 256;

//! ORB descriptor distance (8 x 32 bit words); the device kernels compute the same number with v_bcnt_u32_b32
inline unsigned int compute_descriptor_distance_32(const cv::Mat& desc_1, const cv::Mat& desc_2) {
    unsigned int dist = 0;
    for (unsigned int i = 0; i < 8; ++i) {
        uint32_t a, b;
        std::memcpy(&a, desc_1.data + 4 * i, 4);
        std::memcpy(&b, desc_2.data + 4 * i, 4);
        dist += (unsigned int)__builtin_popcount(a ^ b);
    }
    return dist;
}

class base {
public:
    base(const float lowe_ratio, const bool check_orientation) : lowe_ratio_(lowe_ratio), check_orientation_(check_orientation) {}
    virtual ~base() = default;

protected:
    const float lowe_ratio_;
    const bool check_orientation_;
};

}   // namespace match
}   // namespace openvslam
