// orb_describe.hip -- A5 + A6 + A7 + A8: compute_orientation / ic_angle, the 7x7 sigma=2 blur that precedes description,
// compute_orb_descriptor and correct_keypoint_scale (expected: src/openvslam/feature/orb_extractor.cc,
// util/trigonometric.h; OpenCV fastAtan2, GaussianBlur, cvRound).
//
// One 256-thread workgroup per selected keypoint. Everything a keypoint needs lies in the 43x43 patch around it
// (rBRIEF reaches +-18 px after rotation, the blur adds 3), so the patch is staged in LDS once and
//   * the intensity-centroid moments are exact integer sums over the radius-15 disc (wave + LDS reduction),
//   * the angle is cv::fastAtan2's float polynomial evaluated with explicit non-fused IEEE ops (__fmul_rn/__fadd_rn/
//     __fdiv_rn) in the oracle's order, so the float result is bit-identical to the CPU,
//   * the blur is evaluated ONLY where it is sampled: row pass 8.8 fixed point over 43x37 into LDS, column pass at the 512
//     sample points (instead of blurring 6.4 MP per frame, upstream's cv::GaussianBlur of every level),
//   * thread t evaluates test t; a wave's ballot is 8 descriptor bytes (bit i of byte j = test 8j+i).
// Keypoints are written level-major at their final position: frame offset = sum of lower levels' counts.
#include "ovs_common.h"

namespace ovs {

__constant__ int8_t c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
__constant__ int32_t c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
__constant__ int32_t c_gauss7[7] = {18, 49, 33, 56, 33, 49, 18};

constexpr int kPatch = 43;       // 2*(18+3)+1
constexpr int kPatchR = 21;
constexpr int kBlurW = 37;       // 2*18+1
constexpr int kBlurR = 18;

// cv::fastAtan2 (scalar form), degrees in [0, 360). Every operation individually rounded (no FMA contraction).
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = 2.2204460492503131e-16f;   // (float)DBL_EPSILON
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// util::cos / util::sin: range reduction + 3-term even polynomial, oracle's op order.
__device__ __forceinline__ float poly_cos(float v) {
    const float c1 = 0.99940307f, c2 = -0.49558072f, c3 = 0.03679168f;
    const float v2 = __fmul_rn(v, v);
    return __fadd_rn(c1, __fmul_rn(v2, __fadd_rn(c2, __fmul_rn(c3, v2))));
}
__device__ __forceinline__ float util_cos(float v) {
    const float kPi = 3.14159265358979323846f, kTwoPi = 6.28318530717958647692f, kHalfPi = 1.57079632679489661923f,
                kThreeHalfPi = 4.71238898038468985769f, kInvTwoPi = 0.15915494309189533577f;
    v = __fsub_rn(v, __fmul_rn(floorf(__fmul_rn(v, kInvTwoPi)), kTwoPi));
    v = (0.0f < v) ? v : -v;
    if (v < kHalfPi) return poly_cos(v);
    if (v < kPi) return -poly_cos(__fsub_rn(kPi, v));
    if (v < kThreeHalfPi) return -poly_cos(__fsub_rn(v, kPi));
    return poly_cos(__fsub_rn(kTwoPi, v));
}
__device__ __forceinline__ float util_sin(float v) { return util_cos(__fsub_rn(1.57079632679489661923f, v)); }

__global__ __launch_bounds__(256) void k_describe(const FrameGeo* __restrict__ geo, const uint8_t* __restrict__ img0, size_t stride0,
                                                 size_t frame_stride0, const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                 const uint64_t* __restrict__ lvl_kps, const uint32_t* __restrict__ lvl_count,
                                                 ovs_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                 int32_t* __restrict__ counts, int cap) {
    __shared__ uint8_t patch[kPatch][kPatch + 1];
    __shared__ uint16_t hblur[kPatch][kBlurW + 1];
    __shared__ int red[2][4];
    __shared__ float s_trig[3];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int frame = blockIdx.y;
    const int L = geo->num_levels;
    int level = 0;
    for (int l = 1; l < L; ++l)
        if ((int)blockIdx.x >= geo->lv[l].kp_base) level = l;
    const LevelGeo& g = geo->lv[level];
    const int slot = blockIdx.x - g.kp_base;
    const uint32_t* cnt = lvl_count + frame * L;
    if (blockIdx.x == 0 && tid == 0) {
        int total = 0;
        for (int l = 0; l < L; ++l) total += cnt[l];
        counts[frame] = total < cap ? total : cap;
    }
    if (slot >= (int)cnt[level]) return;
    int out_idx = slot;
    for (int l = 0; l < level; ++l) out_idx += cnt[l];
    if (out_idx >= cap) return;

    const uint64_t kp = lvl_kps[(size_t)frame * geo->total_kp_cap + g.kp_base + slot];
    const int x = (int)cand_x(kp), y = (int)cand_y(kp);
    const uint8_t* img;
    int pitch;
    if (level == 0) {
        img = img0 + (size_t)frame * frame_stride0;
        pitch = (int)stride0;
    } else {
        img = pyr + (size_t)frame * pyr_frame_bytes + g.plane_off;
        pitch = g.pitch;
    }
    // ---- stage the 43x43 patch (keypoints are >= 22 px from every border, so it is always inside the level)
    for (int i = tid; i < kPatch * kPatch; i += 256) {
        const int r = i / kPatch, c = i - r * kPatch;
        patch[r][c] = img[(size_t)(y - kPatchR + r) * pitch + (x - kPatchR + c)];
    }
    __syncthreads();
    // ---- ic_angle: m10 = sum u*I, m01 = sum v*I over the disc |u| <= u_max[|v|], |v| <= 15 (exact integers)
    int m10 = 0, m01 = 0;
    for (int i = tid; i < 31 * 31; i += 256) {
        const int v = i / 31 - 15, u = i - (v + 15) * 31 - 15;
        const int av = v < 0 ? -v : v;
        if ((u < 0 ? -u : u) <= c_umax[av]) {
            const int val = patch[kPatchR + v][kPatchR + u];
            m10 += u * val;
            m01 += v * val;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m10 += __shfl_xor(m10, off);
        m01 += __shfl_xor(m01, off);
    }
    if (lane == 0) {
        red[0][wv] = m10;
        red[1][wv] = m01;
    }
    // ---- blur row pass (8.8 fixed point): hblur[r][c] = sum_k g[k] * patch[r][c + k], c <-> dx = c - 18
    for (int i = tid; i < kPatch * kBlurW; i += 256) {
        const int r = i / kBlurW, c = i - r * kBlurW;
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) acc += (uint32_t)c_gauss7[k] * patch[r][c + k];
        hblur[r][c] = (uint16_t)acc;
    }
    __syncthreads();
    if (tid == 0) {
        const int M10 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const int M01 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const float angle = fast_atan2_deg((float)M01, (float)M10);
        const float rad = __fmul_rn(angle, 0.017453292519943295f);
        s_trig[0] = angle;
        s_trig[1] = util_cos(rad);
        s_trig[2] = util_sin(rad);
    }
    __syncthreads();
    const float cos_a = s_trig[1], sin_a = s_trig[2];
    // ---- steered BRIEF: test `tid`
    auto blurred_at = [&](int px, int py) -> int {
        const float fx = (float)px, fy = (float)py;
        const int dy = __float2int_rn(__fadd_rn(__fmul_rn(fx, sin_a), __fmul_rn(fy, cos_a)));
        const int dx = __float2int_rn(__fsub_rn(__fmul_rn(fx, cos_a), __fmul_rn(fy, sin_a)));
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) acc += (uint32_t)c_gauss7[k] * hblur[dy + kBlurR + k][dx + kBlurR];
        return (int)((acc + 32768u) >> 16);
    };
    const int8_t* p = c_pattern + tid * 4;
    const int t0 = blurred_at(p[0], p[1]), t1 = blurred_at(p[2], p[3]);
    const unsigned long long bits = __ballot(t0 < t1);
    if (lane == 0) *reinterpret_cast<unsigned long long*>(desc + ((size_t)frame * cap + out_idx) * 32 + wv * 8) = bits;
    if (tid == 0) {
        ovs_keypoint k;
        k.x = __fmul_rn((float)x, g.scale);   // correct_keypoint_scale
        k.y = __fmul_rn((float)y, g.scale);
        k.size = g.kp_size;
        k.angle = s_trig[0];
        k.response = (float)cand_score(kp);
        k.octave = level;
        k.class_id = -1;
        kps[(size_t)frame * cap + out_idx] = k;
    }
}

hipError_t launch_describe(const FrameGeo& hgeo, const DevBuffers& d, const uint8_t* img0, size_t stride0, size_t frame_stride0,
                           ovs_keypoint* kps, uint8_t* desc, int32_t* counts, int cap, int batch, hipStream_t s) {
    dim3 grid(hgeo.total_kp_cap, batch);
    hipLaunchKernelGGL(k_describe, grid, dim3(256), 0, s, d.geo, img0, stride0, frame_stride0, d.pyr, d.pyr_frame_bytes, d.lvl_kps,
                       d.lvl_count, kps, desc, counts, cap);
    return hipGetLastError();
}

}   // namespace ovs
