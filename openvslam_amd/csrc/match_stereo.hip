// match_stereo.hip -- M6: match::stereo::compute (expected: src/openvslam/match/stereo.{h,cc}; ORB-SLAM2 ComputeStereoMatches).
//
// Every left keypoint is independent upstream too (its loop is an OpenMP parallel-for), so this is a plain data-parallel path:
//   k_stereo_index             get_right_keypoint_indices_in_each_row as a CSR over image rows: count, scan and fill by ONE workgroup with the
//                              row counters in LDS (round 3: was two memsets + three launches, a quarter of a call's latency). Bucket order is
//                              irrelevant: strict `<` over ascending indices = minimum of (distance, index).
//   k_stereo_match             one lane per left keypoint: candidates of row (int)y, |octave difference| <= 1, disparity window,
//                              Hamming distance, best < (THR_HIGH + THR_LOW) / 2.
//   k_stereo_subpixel          16 lanes per left keypoint, 11 of them evaluate one shift each: L1 distance of the two 11x11
//                              windows (centre value subtracted) on the LEFT keypoint's pyramid level of both extractors --
//                              the pyramids never leave HBM -- then the parabola fit, disparity window and depth.
//   k_stereo_outliers          one workgroup: median of the accepted L1 distances by two-pass radix select (they are < 2^16),
//                              matches above 2 x median are dropped.
#include <algorithm>
#include <cstring>
#include <new>

#include "ovs_common.h"

namespace ovs {

constexpr int kStereoWin = 5, kStereoSlide = 5;

// count -> scan -> fill in one workgroup: s_row[r] first counts the right keypoints whose vertical band covers image row r, then holds the
// fill cursor of that row. Dynamic LDS: (rows0 + 1) + 1024 words.
__global__ __launch_bounds__(1024) void k_stereo_index(const ovs_keypoint* __restrict__ kps_right, const int32_t* __restrict__ n_ptr, int n_fixed,
                                                      PyrView pv, uint32_t* __restrict__ row_off, uint32_t* __restrict__ row_items, uint32_t item_cap,
                                                      uint32_t* __restrict__ overflow) {
    extern __shared__ uint32_t s_dyn[];
    const int rows0 = pv.rows[0];
    uint32_t* const s_row = s_dyn;               // [rows0 + 1]
    uint32_t* const s_part = s_dyn + rows0 + 1;   // [1024]
    const int tid = threadIdx.x;
    const int n = n_ptr ? *n_ptr : n_fixed;
    for (int r = tid; r <= rows0; r += 1024) s_row[r] = 0;
    if (tid == 0) *overflow = 0u;
    __syncthreads();
    auto band = [&](int i, int& lo, int& hi) {
        const ovs_keypoint k = kps_right[i];
        const float r = __fmul_rn(2.0f, pv.scale[k.octave]);
        hi = min((int)ceilf(__fadd_rn(k.y, r)), rows0 - 1);
        lo = max((int)floorf(__fsub_rn(k.y, r)), 0);
    };
    for (int i = tid; i < n; i += 1024) {
        int lo, hi;
        band(i, lo, hi);
        for (int row = lo; row <= hi; ++row) atomicAdd(&s_row[row], 1u);
    }
    __syncthreads();
    // exclusive scan of s_row[0 .. rows0) into row_off[0 .. rows0] (and in place: the fill cursors)
    const int per = (rows0 + 1023) / 1024;
    uint32_t local = 0;
    for (int k = 0; k < per; ++k) {
        const int r = tid * per + k;
        if (r < rows0) local += s_row[r];
    }
    s_part[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - local;
    for (int k = 0; k < per; ++k) {
        const int r = tid * per + k;
        if (r < rows0) {
            const uint32_t c = s_row[r];
            row_off[r] = run;
            s_row[r] = run;
            run += c;
        }
    }
    if (tid == 1023) row_off[rows0] = s_part[1023];
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        int lo, hi;
        band(i, lo, hi);
        for (int row = lo; row <= hi; ++row) {
            const uint32_t pos = atomicAdd(&s_row[row], 1u);
            if (pos < item_cap) row_items[pos] = (uint32_t)i;
            else *overflow = 1u;
        }
    }
}

__global__ __launch_bounds__(256) void k_stereo_match(const ovs_keypoint* __restrict__ kps_left, const uint8_t* __restrict__ desc_left,
                                                     const int32_t* __restrict__ n_ptr, int n_fixed,
                                                     const ovs_keypoint* __restrict__ kps_right, const uint8_t* __restrict__ desc_right,
                                                     PyrView pv, const uint32_t* __restrict__ row_off, const uint32_t* __restrict__ row_items,
                                                     uint32_t item_cap, float max_disp, int32_t* __restrict__ best_right) {
    const int n = n_ptr ? *n_ptr : n_fixed;
    const int il = blockIdx.x * 256 + threadIdx.x;
    if (il >= n) return;
    const ovs_keypoint kl = kps_left[il];
    int32_t result = -1;
    const int row = (int)kl.y;
    if (row >= 0 && row < pv.rows[0]) {
        const float min_x_right = __fsub_rn(kl.x, max_disp), max_x_right = kl.x;   // min_disp = 0
        uint32_t b = row_off[row];
        const uint32_t e = min(row_off[row + 1], item_cap);
        if (b < e && !(max_x_right < 0)) {
            uint32_t a[8];
            const uint32_t* src = reinterpret_cast<const uint32_t*>(desc_left + (size_t)il * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = src[i];
            const uint32_t thr = (OVS_HAMMING_DIST_THR_HIGH + OVS_HAMMING_DIST_THR_LOW) / 2;
            uint32_t best_key = (thr << 16);   // only d < thr can win
            for (; b < e; ++b) {
                const uint32_t ir = row_items[b];
                const ovs_keypoint kr = kps_right[ir];
                if (kr.octave < kl.octave - 1 || kr.octave > kl.octave + 1) continue;
                if (kr.x < min_x_right || max_x_right < kr.x) continue;
                const uint32_t* t = reinterpret_cast<const uint32_t*>(desc_right + (size_t)ir * 32);
                uint32_t d = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) d += __builtin_popcount(a[i] ^ t[i]);
                const uint32_t key = (d << 16) | ir;
                if (key < best_key) best_key = key;
            }
            if ((best_key >> 16) < thr) result = (int32_t)(best_key & 0xFFFFu);
        }
    }
    best_right[il] = result;
}

__global__ __launch_bounds__(256) void k_stereo_subpixel(const ovs_keypoint* __restrict__ kps_left, const int32_t* __restrict__ n_ptr, int n_fixed,
                                                        const ovs_keypoint* __restrict__ kps_right, const int32_t* __restrict__ best_right,
                                                        PyrView pl, PyrView pr, float focal_x_baseline, float max_disp,
                                                        float* __restrict__ stereo_x_right, float* __restrict__ depths,
                                                        int32_t* __restrict__ sad_out, int variant) {
    const int n = n_ptr ? *n_ptr : n_fixed;
    const int il = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int j = threadIdx.x & 15;                // shift index: offset = j - 5 for j < 11
    const int lane = threadIdx.x & 63, gbase = lane & ~15;
    const bool in_range = il < n;
    const int ir = in_range ? best_right[il] : -1;
    bool ok = ir >= 0;
    ovs_keypoint kl{}, kr{};
    if (ok) {
        kl = kps_left[il];
        kr = kps_right[ir];
    }
    const int level = kl.octave;
    const float isf = pl.inv_scale[level];
    const int sxl = __float2int_rn(__fmul_rn(kl.x, isf)), syl = __float2int_rn(__fmul_rn(kl.y, isf)), sxr = __float2int_rn(__fmul_rn(kr.x, isf));
    const int lc = pl.cols[level], lr = pl.rows[level];
    if (ok) {
        const int ini_x = sxr - kStereoSlide - kStereoWin, end_x = sxr + kStereoSlide + kStereoWin + 1;
        if (ini_x < 0 || lc <= end_x) ok = false;
        if (syl - kStereoWin < 0 || lr <= syl + kStereoWin || sxl - kStereoWin < 0 || lc <= sxl + kStereoWin) ok = false;
    }
    uint32_t sad = 0;
    if (ok && j < 2 * kStereoSlide + 1) {
        const int offset = j - kStereoSlide;
        const uint8_t* IL = pl.base[level];
        const uint8_t* IR = pr.base[level];
        const int sl = pl.pitch[level], sr = pr.pitch[level];
        const int cl = IL[(size_t)syl * sl + sxl];
        const int cr = IR[(size_t)syl * sr + sxr + offset];
        for (int dy = -kStereoWin; dy <= kStereoWin; ++dy) {
            const uint8_t* rl = IL + (size_t)(syl + dy) * sl + sxl - kStereoWin;
            const uint8_t* rr = IR + (size_t)(syl + dy) * sr + sxr + offset - kStereoWin;
#pragma unroll
            for (int dx = 0; dx <= 2 * kStereoWin; ++dx) {
                const int a = (int)rl[dx] - cl, b = (int)rr[dx] - cr;
                const int t = a - b;
                sad += (uint32_t)(t < 0 ? -t : t);
            }
        }
    }
    // first strict minimum in shift order = minimum of (sad, shift index)
    uint32_t key = (ok && j < 2 * kStereoSlide + 1) ? ((sad << 4) | (uint32_t)j) : 0xFFFFFFFFu;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(key, off);
        key = o < key ? o : key;
    }
    const int bj = (int)(key & 15u);
    const float c1 = (float)__shfl((int)sad, gbase + max(bj - 1, 0));
    const float c2 = (float)__shfl((int)sad, gbase + bj);
    const float c3 = (float)__shfl((int)sad, gbase + min(bj + 1, 15));
    if (j != 0 || !in_range) return;
    float out_x = -1.0f, out_d = -1.0f;
    int32_t out_sad = -1;
    if (ok && bj != 0 && bj != 2 * kStereoSlide) {
        // rule 20: float arithmetic (default) | double, rounded to float once (variant bit 1)
        const float delta = (variant & 2) ? (float)__ddiv_rn(__dsub_rn((double)c1, (double)c3),
                                                             __dmul_rn(2.0, __dsub_rn(__dadd_rn((double)c1, (double)c3), __dmul_rn(2.0, (double)c2))))
                                          : __fdiv_rn(__fsub_rn(c1, c3), __fmul_rn(2.0f, __fsub_rn(__fadd_rn(c1, c3), __fmul_rn(2.0f, c2))));
        if (!(delta < -1.0f || 1.0f < delta)) {
            float best_x_right = __fmul_rn(pl.scale[level], __fadd_rn(__fadd_rn((float)sxr, (float)(bj - kStereoSlide)), delta));
            float disp = __fsub_rn(kl.x, best_x_right);
            if (!(disp < 0.0f || max_disp <= disp)) {
                if (disp <= 0.0f) {
                    disp = 0.01f;
                    best_x_right = __fsub_rn(kl.x, 0.01f);
                }
                out_d = __fdiv_rn(focal_x_baseline, disp);
                out_x = best_x_right;
                out_sad = (int32_t)(key >> 4);
            }
        }
    }
    stereo_x_right[il] = out_x;
    depths[il] = out_d;
    sad_out[il] = out_sad;
}

// median of the accepted L1 distances (element size/2 of the ascending order), then drop everything above 2 x median
__global__ __launch_bounds__(1024) void k_stereo_outliers(const int32_t* __restrict__ n_ptr, int n_fixed, const int32_t* __restrict__ sad,
                                                         float* __restrict__ stereo_x_right, float* __restrict__ depths,
                                                         int32_t* __restrict__ n_valid, int variant) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_sel[3];   // chosen high byte, rank inside it, total
    __shared__ uint32_t s_cnt;
    const int n = n_ptr ? *n_ptr : n_fixed;
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int v = sad[i];
        if (v >= 0) atomicAdd(&hist[((uint32_t)v >> 8) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t total = 0;
        for (int b = 0; b < 256; ++b) total += hist[b];
        uint32_t rank = total / 2, b = 0;
        if (total) {
            while (rank >= hist[b]) {
                rank -= hist[b];
                ++b;
            }
        }
        s_sel[0] = b;
        s_sel[1] = rank;
        s_sel[2] = total;
    }
    __syncthreads();
    const uint32_t hi = s_sel[0], total = s_sel[2];
    __syncthreads();
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int v = sad[i];
        if (v >= 0 && (((uint32_t)v >> 8) & 255u) == hi) atomicAdd(&hist[(uint32_t)v & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t rank = s_sel[1], b = 0;
        if (total) {
            while (rank >= hist[b]) {
                rank -= hist[b];
                ++b;
            }
        }
        s_sel[1] = (hi << 8) | b;   // the median
    }
    __syncthreads();
    const float thr = __fmul_rn((variant & 1) ? 2.1f : 2.0f, (float)s_sel[1]);   // rule 20: 2.0 (default) | ORB-SLAM2's 1.5f * 1.4f
    uint32_t kept = 0;
    for (int i = tid; i < n; i += 1024) {
        const int v = sad[i];
        if (v < 0) continue;
        if (thr < (float)v) {
            stereo_x_right[i] = -1.0f;
            depths[i] = -1.0f;
        } else {
            ++kept;
        }
    }
    atomicAdd(&s_cnt, kept);
    __syncthreads();
    if (tid == 0 && n_valid) *n_valid = (int32_t)s_cnt;
}

}   // namespace ovs

using namespace ovs;

struct ovs_stereo {
    int device = 0;
    int max_rows = 0, max_kps = 0;
    int variant = 0;   // ovs_stereo_set_variant: bit 0 outlier factor 2.1, bit 1 parabola in double
    uint32_t item_cap = 0;
    hipStream_t stream = nullptr;
    uint32_t* d_row_off = nullptr;
    uint32_t* d_row_items = nullptr;
    uint32_t* d_overflow = nullptr;
    int32_t* d_best_right = nullptr;
    int32_t* d_sad = nullptr;
    int32_t* d_n_valid = nullptr;
    // host-API staging
    ovs_keypoint* d_kps_l = nullptr;
    ovs_keypoint* d_kps_r = nullptr;
    uint8_t* d_desc_l = nullptr;
    uint8_t* d_desc_r = nullptr;
    float* d_x_right = nullptr;
    float* d_depths = nullptr;
    // n_valid | overflow | pad | x_right[max_kps] | depths[max_kps] in ONE device block with a pinned mirror: one copy brings a call's
    // results back (d_n_valid, d_overflow, d_x_right, d_depths point into it)
    uint8_t* d_res = nullptr;
    uint8_t* h_res = nullptr;
};

extern "C" {

ovs_status ovs_stereo_create(int32_t max_rows, int32_t max_keypoints, int32_t device, ovs_stereo** out) {
    if (!out || max_rows < 1 || max_keypoints < 1 || max_keypoints > 65535) return OVS_ERR_INVALID;
    // k_stereo_index keeps the row CSR of the right image in LDS: (rows + 1 + 1024) words of the CU's 160 KiB
    if (sizeof(uint32_t) * ((size_t)max_rows + 1 + 1024) > kMaxLdsPerWorkgroup) return OVS_ERR_CAPACITY;
    *out = nullptr;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    ovs_stereo* s = new (std::nothrow) ovs_stereo();
    if (!s) return OVS_ERR_INVALID;
    s->device = device;
    s->max_rows = max_rows;
    s->max_kps = max_keypoints;
    // a right keypoint sits in at most ceil(y + 2s) - floor(y - 2s) + 1 <= 4 s_max + 3 rows; s_max = 1.2^15 would be absurd: size
    // for the largest scale factor of a 16-level x1.2 pyramid and refuse (OVS_ERR_CAPACITY) beyond it
    s->item_cap = (uint32_t)std::min<size_t>((size_t)max_keypoints * 72, (size_t)1 << 26);
#define CREATE_TRY(expr)                       \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) {                \
            ovs::set_last_error(#expr, _e);    \
            ovs_stereo_destroy(s);             \
            return OVS_ERR_HIP;                \
        }                                      \
    } while (0)
    CREATE_TRY(hipSetDevice(device));
    CREATE_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    const size_t R = (size_t)max_rows + 1, K = (size_t)max_keypoints;
    CREATE_TRY(hipMalloc(&s->d_row_off, sizeof(uint32_t) * R));
    CREATE_TRY(hipMalloc(&s->d_row_items, sizeof(uint32_t) * s->item_cap));
    CREATE_TRY(hipMalloc(&s->d_best_right, sizeof(int32_t) * K));
    CREATE_TRY(hipMalloc(&s->d_sad, sizeof(int32_t) * K));
    CREATE_TRY(hipMalloc(&s->d_kps_l, sizeof(ovs_keypoint) * K));
    CREATE_TRY(hipMalloc(&s->d_kps_r, sizeof(ovs_keypoint) * K));
    CREATE_TRY(hipMalloc(&s->d_desc_l, 32 * K));
    CREATE_TRY(hipMalloc(&s->d_desc_r, 32 * K));
    CREATE_TRY(hipMalloc(&s->d_res, 16 + 2 * sizeof(float) * K));
    CREATE_TRY(hipHostMalloc(&s->h_res, 16 + 2 * sizeof(float) * K, hipHostMallocDefault));
    s->d_n_valid = reinterpret_cast<int32_t*>(s->d_res);
    s->d_overflow = reinterpret_cast<uint32_t*>(s->d_res + 4);
    s->d_x_right = reinterpret_cast<float*>(s->d_res + 16);
    s->d_depths = s->d_x_right + K;
#undef CREATE_TRY
    *out = s;
    return OVS_OK;
}

ovs_status ovs_stereo_destroy(ovs_stereo* s) {
    if (!s) return OVS_OK;
    if (s->stream) hipStreamSynchronize(s->stream);
    void* ptrs[] = {s->d_row_off, s->d_row_items, s->d_best_right, s->d_sad, s->d_kps_l, s->d_kps_r, s->d_desc_l, s->d_desc_r, s->d_res};
    for (void* p : ptrs) hipFree(p);
    if (s->h_res) hipHostFree(s->h_res);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
    return OVS_OK;
}

ovs_status ovs_stereo_set_variant(ovs_stereo* s, int32_t which, int32_t value) {
    if (!s || (value != 0 && value != 1)) return OVS_ERR_INVALID;
    if (which == OVS_STEREO_VARIANT_OUTLIER_FACTOR) s->variant = (s->variant & ~1) | value;
    else if (which == OVS_STEREO_VARIANT_PARABOLA) s->variant = (s->variant & ~2) | (value << 1);
    else return OVS_ERR_INVALID;
    return OVS_OK;
}

ovs_status ovs_stereo_compute_dev(ovs_stereo* s, const ovs_orb* left, int32_t frame_left, const ovs_orb* right, int32_t frame_right,
                                  const ovs_keypoint* d_kps_left, const uint8_t* d_desc_left, const int32_t* d_n_left, int32_t cap_left,
                                  const ovs_keypoint* d_kps_right, const uint8_t* d_desc_right, const int32_t* d_n_right, int32_t cap_right,
                                  float focal_x_baseline, float true_baseline, float* d_stereo_x_right, float* d_depths,
                                  int32_t* d_n_valid, void* stream) {
    if (!s || !left || !right || !d_kps_left || !d_desc_left || !d_kps_right || !d_desc_right || !d_stereo_x_right || !d_depths ||
        cap_left < 1 || cap_right < 1 || !(true_baseline > 0))
        return OVS_ERR_INVALID;
    if (cap_left > s->max_kps || cap_right > s->max_kps) return OVS_ERR_CAPACITY;
    PyrView pl, pr;
    if (!orb_pyramid_view(left, frame_left, &pl) || !orb_pyramid_view(right, frame_right, &pr)) return OVS_ERR_INVALID;
    if (pl.num_levels != pr.num_levels || pl.rows[0] != pr.rows[0] || pl.cols[0] != pr.cols[0]) return OVS_ERR_INVALID;
    if (pl.rows[0] > s->max_rows) return OVS_ERR_CAPACITY;
    if (orb_device(left) != s->device || orb_device(right) != s->device) return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    const int rows0 = pl.rows[0];
    const float max_disp = focal_x_baseline / true_baseline;
    const dim3 gl((cap_left + 255) / 256);
    const size_t index_lds = sizeof(uint32_t) * (size_t)(rows0 + 1 + 1024);
    if (index_lds > 64 * 1024) {   // above the default limit of dynamic LDS (images taller than ~15 000 rows): raise it, per device
        static LdsAttrCache configured;
        OVS_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(k_stereo_index), index_lds, configured));
    }
    hipLaunchKernelGGL(k_stereo_index, dim3(1), dim3(1024), index_lds, st, d_kps_right, d_n_right, cap_right, pr,
                       s->d_row_off, s->d_row_items, s->item_cap, s->d_overflow);
    OVS_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_stereo_match, gl, dim3(256), 0, st, d_kps_left, d_desc_left, d_n_left, cap_left, d_kps_right, d_desc_right, pr,
                       (const uint32_t*)s->d_row_off, (const uint32_t*)s->d_row_items, s->item_cap, max_disp, s->d_best_right);
    hipLaunchKernelGGL(k_stereo_subpixel, dim3((cap_left + 15) / 16), dim3(256), 0, st, d_kps_left, d_n_left, cap_left, d_kps_right,
                       (const int32_t*)s->d_best_right, pl, pr, focal_x_baseline, max_disp, d_stereo_x_right, d_depths, s->d_sad, s->variant);
    hipLaunchKernelGGL(k_stereo_outliers, dim3(1), dim3(1024), 0, st, d_n_left, cap_left, (const int32_t*)s->d_sad, d_stereo_x_right,
                       d_depths, d_n_valid ? d_n_valid : s->d_n_valid, s->variant);
    OVS_HIP_TRY(hipGetLastError());
    return OVS_OK;
}

ovs_status ovs_stereo_compute(ovs_stereo* s, const ovs_orb* left, const ovs_orb* right, const ovs_keypoint* kps_left,
                              const uint8_t* desc_left, int32_t n_left, const ovs_keypoint* kps_right, const uint8_t* desc_right,
                              int32_t n_right, float focal_x_baseline, float true_baseline, float* stereo_x_right, float* depths,
                              int32_t* n_valid) {
    if (!s || !left || !right || n_left < 0 || n_right < 0) return OVS_ERR_INVALID;
    if (n_valid) *n_valid = 0;
    if (n_left == 0) return OVS_OK;
    if (!kps_left || !desc_left || !stereo_x_right || !depths) return OVS_ERR_INVALID;
    if (n_right == 0) {
        for (int i = 0; i < n_left; ++i) stereo_x_right[i] = depths[i] = -1.0f;
        return OVS_OK;
    }
    if (!kps_right || !desc_right) return OVS_ERR_INVALID;
    if (n_left > s->max_kps || n_right > s->max_kps) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = s->stream;
    // the extractors ran on their own streams: wait for exactly those two (never the whole device -- tracking and mapping threads
    // share it)
    OVS_HIP_TRY(hipStreamSynchronize(ovs::orb_last_stream(left)));
    OVS_HIP_TRY(hipStreamSynchronize(ovs::orb_last_stream(right)));
    // Residency (round 3): data::frame's stereo constructor calls this right after the two extract() calls, with exactly the vectors they
    // filled -- those keypoints and descriptors are still in the extractors' device output blocks. Verified byte for byte against the pinned
    // blocks the results were downloaded into (a few microseconds) before the four uploads are skipped.
    const ovs_keypoint *dkl = s->d_kps_l, *dkr = s->d_kps_r;
    const uint8_t *ddl = s->d_desc_l, *ddr = s->d_desc_r;
    if (!ovs::orb_host_outputs_equal(left, kps_left, desc_left, n_left, &dkl, &ddl)) {
        OVS_HIP_TRY(hipMemcpyAsync(s->d_kps_l, kps_left, sizeof(ovs_keypoint) * n_left, hipMemcpyHostToDevice, st));
        OVS_HIP_TRY(hipMemcpyAsync(s->d_desc_l, desc_left, (size_t)32 * n_left, hipMemcpyHostToDevice, st));
    }
    if (!ovs::orb_host_outputs_equal(right, kps_right, desc_right, n_right, &dkr, &ddr)) {
        OVS_HIP_TRY(hipMemcpyAsync(s->d_kps_r, kps_right, sizeof(ovs_keypoint) * n_right, hipMemcpyHostToDevice, st));
        OVS_HIP_TRY(hipMemcpyAsync(s->d_desc_r, desc_right, (size_t)32 * n_right, hipMemcpyHostToDevice, st));
    }
    ovs_status rc = ovs_stereo_compute_dev(s, left, 0, right, 0, dkl, ddl, nullptr, n_left, dkr, ddr, nullptr, n_right, focal_x_baseline,
                                           true_baseline, s->d_x_right, s->d_depths, s->d_n_valid, st);
    if (rc != OVS_OK) return rc;
    // one copy for [n_valid | overflow | x_right (whole region) | depths of the first n_left] into the pinned mirror, one wait
    const size_t off_depths = 16 + sizeof(float) * (size_t)s->max_kps, bytes = off_depths + sizeof(float) * (size_t)n_left;
    OVS_HIP_TRY(hipMemcpyAsync(s->h_res, s->d_res, bytes, hipMemcpyDeviceToHost, st));
    OVS_HIP_TRY(hipStreamSynchronize(st));
    std::memcpy(stereo_x_right, s->h_res + 16, sizeof(float) * (size_t)n_left);
    std::memcpy(depths, s->h_res + off_depths, sizeof(float) * (size_t)n_left);
    int32_t nv;
    uint32_t overflow;
    std::memcpy(&nv, s->h_res, sizeof(nv));
    std::memcpy(&overflow, s->h_res + 4, sizeof(overflow));
    if (n_valid) *n_valid = nv;
    return overflow ? OVS_ERR_CAPACITY : OVS_OK;
}

}   // extern "C"
