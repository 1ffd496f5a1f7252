// orb_describe.hip -- A5 + A6 + A7 + A8: compute_orientation / ic_angle, the 7x7 sigma=2 blur that precedes description,
// compute_orb_descriptor and correct_keypoint_scale (expected: src/openvslam/feature/orb_extractor.cc,
// util/trigonometric.h; OpenCV fastAtan2, GaussianBlur, cvRound).
//
// One WAVE (a 64-thread workgroup) per selected keypoint -- no workgroup barriers, 30 keypoints in flight per CU.
// Everything a keypoint needs lies in the 43x43 patch around it (rBRIEF reaches +-18 px after rotation, the blur adds 3):
//   * the patch is staged in LDS as 43 aligned 64-byte row segments (three 16-byte loads per lane),
//   * the intensity-centroid moments are exact integer sums over the radius-15 disc (two lanes per disc row + wave reduce),
//   * the angle is cv::fastAtan2's float polynomial evaluated with explicit non-fused IEEE ops (__fmul_rn/__fadd_rn/
//     __fdiv_rn) in the oracle's order, so the float result is bit-identical to the CPU; every lane evaluates it (uniform),
//   * the blur is evaluated ONLY where it is sampled: row pass in 8.8 fixed point over 43x37 (four outputs per lane-step from
//     word reads, v_alignbyte_b32 + v_dot4_u32_u8), column pass at the 512 sample points (instead of blurring 6.4 MP per
//     frame, upstream's cv::GaussianBlur of every level),
//   * lane l evaluates tests l, l+64, l+128, l+192; each ballot is 8 descriptor bytes (bit i of byte j = test 8j+i).
// Keypoints are written level-major at their final position: frame offset = sum of lower levels' counts.
#include <cstdlib>

#include "ovs_common.h"

namespace ovs {

__constant__ int8_t c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
__constant__ int32_t c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
// ic_angle: per disc row |v| and half (u < 0 / u >= 0), four words of 0/1 mask bytes and four of (u + 16) * mask weight bytes for the
// sixteen pixels u = -16 .. -1 / 0 .. 15 (from u_max above)
__constant__ uint32_t c_ic_tab[16 * 2 * 8] = {
    0x01010100u, 0x01010101u, 0x01010101u, 0x01010101u, 0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 0, left half
    0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x1f1e1d1cu,   // |v| = 0, right half
    0x01010100u, 0x01010101u, 0x01010101u, 0x01010101u, 0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 1, left half
    0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x1f1e1d1cu,   // |v| = 1, right half
    0x01010100u, 0x01010101u, 0x01010101u, 0x01010101u, 0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 2, left half
    0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x1f1e1d1cu,   // |v| = 2, right half
    0x01010100u, 0x01010101u, 0x01010101u, 0x01010101u, 0x03020100u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 3, left half
    0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x1f1e1d1cu,   // |v| = 3, right half
    0x01010000u, 0x01010101u, 0x01010101u, 0x01010101u, 0x03020000u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 4, left half
    0x01010101u, 0x01010101u, 0x01010101u, 0x00010101u, 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x001e1d1cu,   // |v| = 4, right half
    0x01010000u, 0x01010101u, 0x01010101u, 0x01010101u, 0x03020000u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 5, left half
    0x01010101u, 0x01010101u, 0x01010101u, 0x00010101u, 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x001e1d1cu,   // |v| = 5, right half
    0x01010000u, 0x01010101u, 0x01010101u, 0x01010101u, 0x03020000u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 6, left half
    0x01010101u, 0x01010101u, 0x01010101u, 0x00010101u, 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x001e1d1cu,   // |v| = 6, right half
    0x01000000u, 0x01010101u, 0x01010101u, 0x01010101u, 0x03000000u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 7, left half
    0x01010101u, 0x01010101u, 0x01010101u, 0x00000101u, 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x00001d1cu,   // |v| = 7, right half
    0x01000000u, 0x01010101u, 0x01010101u, 0x01010101u, 0x03000000u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 8, left half
    0x01010101u, 0x01010101u, 0x01010101u, 0x00000101u, 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x00001d1cu,   // |v| = 8, right half
    0x00000000u, 0x01010101u, 0x01010101u, 0x01010101u, 0x00000000u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 9, left half
    0x01010101u, 0x01010101u, 0x01010101u, 0x00000001u, 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x0000001cu,   // |v| = 9, right half
    0x00000000u, 0x01010100u, 0x01010101u, 0x01010101u, 0x00000000u, 0x07060500u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 10, left half
    0x01010101u, 0x01010101u, 0x01010101u, 0x00000000u, 0x13121110u, 0x17161514u, 0x1b1a1918u, 0x00000000u,   // |v| = 10, right half
    0x00000000u, 0x01010000u, 0x01010101u, 0x01010101u, 0x00000000u, 0x07060000u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 11, left half
    0x01010101u, 0x01010101u, 0x00010101u, 0x00000000u, 0x13121110u, 0x17161514u, 0x001a1918u, 0x00000000u,   // |v| = 11, right half
    0x00000000u, 0x01000000u, 0x01010101u, 0x01010101u, 0x00000000u, 0x07000000u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 12, left half
    0x01010101u, 0x01010101u, 0x00000101u, 0x00000000u, 0x13121110u, 0x17161514u, 0x00001918u, 0x00000000u,   // |v| = 12, right half
    0x00000000u, 0x00000000u, 0x01010101u, 0x01010101u, 0x00000000u, 0x00000000u, 0x0b0a0908u, 0x0f0e0d0cu,   // |v| = 13, left half
    0x01010101u, 0x01010101u, 0x00000001u, 0x00000000u, 0x13121110u, 0x17161514u, 0x00000018u, 0x00000000u,   // |v| = 13, right half
    0x00000000u, 0x00000000u, 0x01010000u, 0x01010101u, 0x00000000u, 0x00000000u, 0x0b0a0000u, 0x0f0e0d0cu,   // |v| = 14, left half
    0x01010101u, 0x00010101u, 0x00000000u, 0x00000000u, 0x13121110u, 0x00161514u, 0x00000000u, 0x00000000u,   // |v| = 14, right half
    0x00000000u, 0x00000000u, 0x00000000u, 0x01010100u, 0x00000000u, 0x00000000u, 0x00000000u, 0x0f0e0d00u,   // |v| = 15, left half
    0x01010101u, 0x00000000u, 0x00000000u, 0x00000000u, 0x13121110u, 0x00000000u, 0x00000000u, 0x00000000u,   // |v| = 15, right half
};
constexpr uint32_t kG0123 = 18u | (34u << 8) | (48u << 16) | (56u << 24);
constexpr uint32_t kG456 = 48u | (34u << 8) | (18u << 16);
// variant (geo->variant bit 2, ORACLE_SPEC rule 10): independently rounded taps 18, 34, 49, 55 (sum 257, result saturates at 255)
constexpr uint32_t kG0123Indep = 18u | (34u << 8) | (49u << 16) | (55u << 24);
constexpr uint32_t kG456Indep = 49u | (34u << 8) | (18u << 16);

constexpr int kPatch = 43;       // 2*(18+3)+1
constexpr int kPatchR = 21;
constexpr int kBlurW = 37;       // 2*18+1 (columns of the row-blurred patch that are read)
constexpr int kBlurR = 18;
constexpr int kPatchPitch = 64;   // bytes: one aligned 64-byte window of each image row
constexpr int kHbPitch = 40;      // u16 per row of the row-blurred patch (37 used)
static_assert(kBlurW <= kHbPitch && kBlurW + 6 == kPatch, "blur window geometry");

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

template <int BYTE>
__device__ __forceinline__ float cvt_s8(uint32_t w) {   // (float)(int8_t)(w >> 8 * BYTE)
    float f;
    if (BYTE == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(f) : "v"(w));
    if (BYTE == 1) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(f) : "v"(w));
    if (BYTE == 2) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(f) : "v"(w));
    if (BYTE == 3) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(f) : "v"(w));
    return f;
}

__device__ __forceinline__ int wave_total_in_lane63(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return v;
}

__global__ __launch_bounds__(64) void k_describe(const FrameGeo* __restrict__ geo, const uint8_t* __restrict__ img0, size_t stride0,
                                                size_t frame_stride0, const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                const uint64_t* __restrict__ lvl_kps, const uint32_t* __restrict__ lvl_count,
                                                ovs_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                int32_t* __restrict__ counts, int cap, int batch, int xcd_map, ovs_keypoint* __restrict__ kps_m,
                                                uint8_t* __restrict__ desc_m, int32_t* __restrict__ counts_m) {
    // kps_m / desc_m / counts_m (round 6, one-frame host calls): a second copy of every output, written straight into the caller-side pinned
    // block -- the frame's results then need no D2H copy command after the kernel (nullptr: batches, device-resident callers)
    __shared__ __attribute__((aligned(16))) uint8_t patch[kPatch * kPatchPitch];
    __shared__ __attribute__((aligned(16))) uint16_t hblur[kPatch * kHbPitch];

    const int lane = threadIdx.x;
    // Workgroup -> (frame, keypoint slot). Workgroups are handed to the eight XCDs round-robin in launch order, and each XCD has its own
    // 4 MB L2: with the plain (slot, frame) order the patches of one frame -- which overlap heavily: 2000 x 43 rows x one or two 128-byte
    // lines = 14 MB of line fetches over a 6.8 MB pyramid -- are spread over all eight L2s and every line comes in from the fabric again
    // (13.8 MB per 1080p frame measured). Mapped so that XCD x walks the slots of frame 8 g + x in order, a level's plane (<= 2 MB)
    // stays in that XCD's L2 while its keypoints are described: 5.7 MB per frame, 2.4x less (profiles/r03_describe_pmc.txt).
    // The kernel's TIME does not move with it: it is bound by the vector ALU (637 VALU instructions per keypoint-wave x 4 cycles x 253
    // waves per SIMD = the 0.33 ms per 128 frames), see DESIGN.md section 3.4 for what was tried against that.
    int frame, kslot;
    const int total_cap = geo->total_kp_cap;
    if (xcd_map) {
        const uint32_t w = blockIdx.x, x = w & 7u, j = w >> 3;
        const uint32_t grp = j / (uint32_t)total_cap;
        kslot = (int)(j - grp * (uint32_t)total_cap);
        frame = (int)(grp * 8u + x);
        if (frame >= batch) return;
    } else {
        frame = blockIdx.x / (uint32_t)total_cap;
        kslot = blockIdx.x - frame * total_cap;
    }
    const int L = geo->num_levels;
    int level = 0;
    for (int l = 1; l < L; ++l)
        if (kslot >= geo->lv[l].kp_base) level = l;
    const LevelGeo& g = geo->lv[level];
    const int slot = kslot - g.kp_base;
    const uint32_t* cnt = lvl_count + frame * L;
    if (kslot == 0 && lane == 0) {
        int total = 0;
        for (int l = 0; l < L; ++l) total += cnt[l];
        counts[frame] = total < cap ? total : cap;
        if (counts_m) counts_m[frame] = total < cap ? total : cap;
    }
    if (slot >= (int)cnt[level]) return;
    int out_idx = slot;
    for (int l = 0; l < level; ++l) out_idx += cnt[l];
    if (out_idx >= cap) return;

    const uint64_t kp = lvl_kps[(size_t)frame * total_cap + kslot];
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
    // ---- stage the 43x43 patch as 43 aligned 64-byte row segments. Keypoints are >= 22 px from every border, so the patch
    // is inside the level; a segment may run up to 20 bytes past the row's last pixel (into the pitch padding or the next
    // row of the same frame -- rows y-21 .. y+21 never include the frame's last row), never outside the allocation.
    const int xs = x - kPatchR;
    const int xa = xs & ~15;
    const int off = __builtin_amdgcn_readfirstlane(xs - xa);   // patch column c lives at segment byte off + c; wave-uniform
    const uint8_t* seg0 = img + (size_t)(y - kPatchR) * pitch + xa;
    if (((pitch & 15) == 0) && ((reinterpret_cast<uintptr_t>(img) & 15) == 0)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = lane + 64 * j;
            if (i < kPatch * 4) {
                const int r = i >> 2, q = i & 3;
                *reinterpret_cast<uint4*>(patch + r * kPatchPitch + 16 * q) = *reinterpret_cast<const uint4*>(seg0 + (size_t)r * pitch + 16 * q);
            }
        }
    } else {   // 4-byte aligned base / stride (the ABI's minimum)
        for (int i = lane; i < kPatch * 16; i += 64) {
            const int r = i >> 4, q = i & 15;
            *reinterpret_cast<uint32_t*>(patch + r * kPatchPitch + 4 * q) = *reinterpret_cast<const uint32_t*>(seg0 + (size_t)r * pitch + 4 * q);
        }
    }
    __syncthreads();   // single-wave workgroup: orders the LDS writes above against the reads below, no s_barrier cost

    // ---- ic_angle: m10 = sum u*I, m01 = sum v*I over the disc |u| <= u_max[|v|], |v| <= 15 (exact integers).
    // Lane 2k / 2k+1 sums the negative / non-negative u of disc row v = k - 15.
    // Sixteen pixels per lane as four words (v_alignbyte_b32 for the sub-word offset), sums as v_dot4_u32_u8 against 0/1 mask bytes and
    // (u + 16) * mask weight bytes built from u_max: 4 LDS word reads + 8 dot products instead of 16 byte reads with a compare each.
    int m10 = 0, m01 = 0;
    if (lane < 62) {
        const int v = (lane >> 1) - 15, h = lane & 1;
        // lane h = 0: u = -16 .. -1 (pixel j <-> u = j - 16, inside the disc iff j >= 16 - u_max); h = 1: u = 0 .. 15 (inside iff j <= u_max)
        const int first = off + kPatchR + (h ? 0 : -16);                     // byte offset of pixel j = 0 in the row segment (>= 5)
        const uint32_t* pw = reinterpret_cast<const uint32_t*>(patch + (kPatchR + v) * kPatchPitch + (first & ~3));
        const int shb = first & 3;
        const uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2], w3 = pw[3], w4 = pw[4];
        const uint32_t px[4] = {__builtin_amdgcn_alignbyte(w1, w0, shb), __builtin_amdgcn_alignbyte(w2, w1, shb),
                                __builtin_amdgcn_alignbyte(w3, w2, shb), __builtin_amdgcn_alignbyte(w4, w3, shb)};
        const uint32_t* tab = c_ic_tab + (((v < 0 ? -v : v) * 2 + h) << 3);
        uint32_t sum = 0, wsum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sum = __builtin_amdgcn_udot4(px[q], tab[q], sum, false);
            wsum = __builtin_amdgcn_udot4(px[q], tab[4 + q], wsum, false);
        }
        m10 = (int)wsum - 16 * (int)sum;
        m01 = v * (int)sum;
    }
    // wave totals by the DPP prefix pattern (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31: lane 63 ends with the sum of all lanes) and one
    // v_readlane each: 12 cross-lane adds instead of 12 ds_bpermute round trips + 12 adds (integer sums: any order gives the same bits)
    m10 = __builtin_amdgcn_readlane(wave_total_in_lane63(m10), 63);
    m01 = __builtin_amdgcn_readlane(wave_total_in_lane63(m01), 63);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // upstream evaluates keypt.angle * M_PI / 180.0 in double and rounds to float once (ORACLE_SPEC rule 11)
    // (one f64 multiply by RN(pi / 180): exhaustively equal to the product-then-quotient form on [0, 360], include/ovs_detmath.h)
    const float rad = ovs_det_deg2rad(angle);
    // rule 11 as a run-time variant (geo->variant bit 3, wave-uniform): libm's cosf / sinf (ovs_detmath.h: glibc's algorithm, bit for bit)
    // instead of OpenVSLAM's util::cos / util::sin polynomial
    const bool trig_libm = (geo->variant & 8) != 0;
    const float cos_a = trig_libm ? ovs_det_cosf(rad) : util_cos(rad), sin_a = trig_libm ? ovs_det_sinf(rad) : util_sin(rad);

    // ---- blur row pass (8.8 fixed point): hblur[r][c] = sum_k g[k] * patch[r][c + k], c <-> dx = c - 18.
    // One lane-step = outputs c = 4q .. 4q+3 of row r from 4 aligned words (columns 37..39 are computed and never read).
    const int sh = off & 3;
    const bool taps_indep = (geo->variant & 4) != 0;
    const uint32_t g0123 = taps_indep ? kG0123Indep : kG0123, g456 = taps_indep ? kG456Indep : kG456;
    uint32_t* hb32 = reinterpret_cast<uint32_t*>(hblur);
    // output j of a lane-step is the tap string slid j bytes along the twelve pixels n0 n1 n2: sliding the TAPS (wave-uniform words, built
    // once) instead of the pixels needs 2 + 2 + 3 + 3 dot products and no per-output byte alignment (was 8 dot products + 6 alignments)
    const unsigned long long taps = (unsigned long long)g0123 | ((unsigned long long)g456 << 32);   // g0 .. g6 in bytes 0 .. 6
    const uint32_t t1a = (uint32_t)(taps << 8), t1b = (uint32_t)(taps >> 24);
    const uint32_t t2a = (uint32_t)(taps << 16), t2b = (uint32_t)(taps >> 16), t2c = (uint32_t)(taps >> 48);
    const uint32_t t3a = (uint32_t)(taps << 24), t3b = (uint32_t)(taps >> 8), t3c = (uint32_t)(taps >> 40);
    // lane -> (row r0 = lane / 10 of a six-row band, column group q = lane % 10), lanes 60 .. 63 idle; band it covers rows 6 it + r0. Every
    // address is the lane's base plus a compile-time constant (no per-step index arithmetic: the 64-lane stride over the 430 steps cost
    // twelve instructions per step for the division by ten)
    if (lane < 60) {
        const int r0 = (lane * 0x199a) >> 16, q = lane - 10 * r0;   // lane / 10 for lane < 64
        const uint32_t* const pw0 = reinterpret_cast<const uint32_t*>(patch + r0 * kPatchPitch + ((off + 4 * q) & ~3));
        uint32_t* const hb0 = hb32 + r0 * (kHbPitch / 2) + 2 * q;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (it == 7 && r0 != 0) break;   // rows 42 .. 47: only row 42 exists
            const uint32_t* pw = pw0 + it * 6 * (kPatchPitch / 4);
            const uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2], w3 = pw[3];
            const uint32_t n0 = __builtin_amdgcn_alignbyte(w1, w0, sh), n1 = __builtin_amdgcn_alignbyte(w2, w1, sh),
                           n2 = __builtin_amdgcn_alignbyte(w3, w2, sh);
            // every sum <= 255 * 257 = 65535
            const uint32_t o0 = __builtin_amdgcn_udot4(n0, g0123, __builtin_amdgcn_udot4(n1, g456, 0u, false), false);
            const uint32_t o1 = __builtin_amdgcn_udot4(n0, t1a, __builtin_amdgcn_udot4(n1, t1b, 0u, false), false);
            const uint32_t o2 = __builtin_amdgcn_udot4(n0, t2a, __builtin_amdgcn_udot4(n1, t2b, __builtin_amdgcn_udot4(n2, t2c, 0u, false), false), false);
            const uint32_t o3 = __builtin_amdgcn_udot4(n0, t3a, __builtin_amdgcn_udot4(n1, t3b, __builtin_amdgcn_udot4(n2, t3c, 0u, false), false), false);
            uint32_t* hb = hb0 + it * 6 * (kHbPitch / 2);
            hb[0] = o0 | (o1 << 16);
            hb[1] = o2 | (o3 << 16);
        }
    }
    __syncthreads();

    // ---- steered BRIEF: column pass of the blur at the sampled points only; lane evaluates tests lane + 64*t
    auto blurred_at = [&](float fx, float fy) -> int {
        const int dy = __float2int_rn(__fadd_rn(__fmul_rn(fx, sin_a), __fmul_rn(fy, cos_a)));
        const int dx = __float2int_rn(__fsub_rn(__fmul_rn(fx, cos_a), __fmul_rn(fy, sin_a)));
        const uint16_t* hp = hblur + (dy + kBlurR) * kHbPitch + dx + kBlurR;
        // symmetric taps g[k] = g[6 - k]: three 2.3-cycle adds + four multiply-adds instead of seven multiply-adds (sums <= 2 * 65535: exact)
        const uint32_t a0 = (uint32_t)hp[0] + hp[6 * kHbPitch], a1 = (uint32_t)hp[kHbPitch] + hp[5 * kHbPitch],
                       a2 = (uint32_t)hp[2 * kHbPitch] + hp[4 * kHbPitch], a3 = hp[3 * kHbPitch];
        uint32_t acc = 32768u;   // round half up
        acc += (g0123 & 255u) * a0;
        acc += ((g0123 >> 8) & 255u) * a1;
        acc += ((g0123 >> 16) & 255u) * a2;
        acc += (g0123 >> 24) * a3;
        return (int)min(acc >> 16, 255u);   // (only the 257-sum variant can exceed 255)
    };
    unsigned long long bits[4];
    const uint32_t* pat = reinterpret_cast<const uint32_t*>(c_pattern);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t pp = pat[lane + 64 * t];
        // the four signed pattern bytes straight to float (SDWA byte select with sign extension: one instruction each, not extract + convert)
        const int t0 = blurred_at(cvt_s8<0>(pp), cvt_s8<1>(pp));
        const int t1 = blurred_at(cvt_s8<2>(pp), cvt_s8<3>(pp));
        bits[t] = __ballot(t0 < t1);
    }
    if (lane < 4) {
        const unsigned long long b = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
        reinterpret_cast<unsigned long long*>(desc + ((size_t)frame * cap + out_idx) * 32)[lane] = b;
        if (desc_m) reinterpret_cast<unsigned long long*>(desc_m + ((size_t)frame * cap + out_idx) * 32)[lane] = b;
    }
    if (lane == 0) {
        ovs_keypoint k;
        k.x = __fmul_rn((float)x, g.scale);   // correct_keypoint_scale
        k.y = __fmul_rn((float)y, g.scale);
        k.size = g.kp_size;
        k.angle = angle;
        k.response = (float)cand_score(kp);
        k.octave = level;
        k.class_id = -1;
        kps[(size_t)frame * cap + out_idx] = k;
        if (kps_m) kps_m[(size_t)frame * cap + out_idx] = k;
    }
}

hipError_t launch_describe(const FrameGeo& hgeo, const DevBuffers& d, const uint8_t* img0, size_t stride0, size_t frame_stride0,
                           ovs_keypoint* kps, uint8_t* desc, int32_t* counts, int cap, int batch, hipStream_t s, ovs_keypoint* kps_m, uint8_t* desc_m,
                           int32_t* counts_m) {
    const int xcd_map = tuning().describe_xcd ? 1 : 0;   // 0: plain frame-major order (A/B of the XCD mapping)
    const int frames = xcd_map ? ((batch + 7) & ~7) : batch;   // frame 8 g + x on XCD x: pad the last group
    dim3 grid((unsigned)hgeo.total_kp_cap * (unsigned)frames);
    hipLaunchKernelGGL(k_describe, grid, dim3(64), 0, s, d.geo, img0, stride0, frame_stride0, d.pyr, d.pyr_frame_bytes, d.lvl_kps,
                       d.lvl_count, kps, desc, counts, cap, batch, xcd_map, kps_m, desc_m, counts_m);
    return hipGetLastError();
}

}   // namespace ovs
