// match_hamming.hip -- M1 + M2: match::base::compute_descriptor_distance_32 and match::robust::brute_force_match
// (expected: src/openvslam/match/base.h, robust.{h,cc}).
//
// Upstream's loop is sequential: keyframe keypoints (idx_2) in order, each scanning ALL frame keypoints (idx_1) that no
// earlier idx_2 has claimed, keeping best / second-best distance with strict `<`, accepting iff
// best <= HAMMING_DIST_THR_LOW and !(lowe_ratio * second < best). Exact parallel form used here:
//   1. k_hamming_near (the O(n1*n2) part, integer VALU-bound): one lane per idx_2; the idx_1 descriptor is wave-uniform, so
//      it is fetched with SCALAR loads (s_load_dwordx8) and XORed straight from SGPRs: 8 v_xor + 8 v_bcnt_u32_b32 (with
//      accumulate) per pair, no LDS, no shuffles. Only distances that can influence the outcome are kept: d <= near_thr,
//      where near_thr = max(THR_LOW, tau-1) and tau is the smallest `second` for which the ratio test can no longer
//      reject a best <= THR_LOW. Those are rare (the true match and near-duplicates): a short per-query list.
//   2. k_bf_resolve: one workgroup per problem replays the claim order in parallel rounds. A pending query finalises in a
//      round iff it is the lowest pending query touching every frame keypoint of its near list (LDS atomicMin marks); then
//      all earlier queries that could claim one of its candidates are final, so its view of `already_matched` is exact.
//      Queries finalised in one round never share a candidate, so their claims do not race.
//   3. If a near list overflows (pathological inputs: many near-identical descriptors) the problem falls back to a
//      literal serial replay with cooperative full-distance scans -- slow, but exact.
// Pairs are emitted in upstream's order (ascending idx_2).
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "ovs_common.h"

namespace ovs {

constexpr int kNearK = 8;             // near-list capacity per query
constexpr int kResolveThreads = 1024;

__device__ __forceinline__ uint32_t hamming256(const uint32_t (&a)[8], const uint32_t* __restrict__ b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d = __builtin_popcount(a[i] ^ b[i]) + d;
    return d;
}

// ---- 1. all pairs, near lists -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hamming_near(const uint8_t* __restrict__ desc_1, size_t stride_1,
                                                     const int32_t* __restrict__ n1_arr, const uint8_t* __restrict__ desc_2,
                                                     size_t stride_2, const int32_t* __restrict__ n2_arr,
                                                     const uint8_t* __restrict__ valid_2, int max_n2, uint32_t near_thr,
                                                     uint32_t* __restrict__ near_cnt, uint32_t* __restrict__ near_list) {
    const int p = blockIdx.y;
    const int n1 = n1_arr[p], n2 = n2_arr[p];
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n2) return;
    const bool active = q < n2 && (!valid_2 || valid_2[(size_t)p * (stride_2 / 32) + q]);
    uint32_t a[8];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(desc_2 + (size_t)p * stride_2 + (size_t)(active ? q : 0) * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = src[i];
    }
    const uint32_t* __restrict__ t = reinterpret_cast<const uint32_t*>(desc_1 + (size_t)p * stride_1);
    uint32_t* my_list = near_list + ((size_t)p * max_n2 + q) * kNearK;
    uint32_t cnt = 0;
    for (int j = 0; j < n1; ++j) {
        const uint32_t d = hamming256(a, t + (size_t)j * 8);   // wave-uniform address: scalar loads
        if (d <= near_thr && active) {
            if (cnt < (uint32_t)kNearK) my_list[cnt] = (d << 16) | (uint32_t)j;
            ++cnt;
        }
    }
    if (q < n2) near_cnt[(size_t)p * max_n2 + q] = active ? cnt : 0u;
}

// ---- 2./3. resolve ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ratio_rejects(float lowe_ratio, uint32_t second, uint32_t best) {
    return __fmul_rn(lowe_ratio, (float)second) < (float)best;
}

__global__ __launch_bounds__(kResolveThreads) void k_bf_resolve(const uint8_t* __restrict__ desc_1, size_t stride_1,
                                                               const int32_t* __restrict__ n1_arr,
                                                               const uint8_t* __restrict__ desc_2, size_t stride_2,
                                                               const int32_t* __restrict__ n2_arr,
                                                               const uint8_t* __restrict__ valid_2, int max_n1, int max_n2,
                                                               float lowe_ratio, const uint32_t* __restrict__ near_cnt,
                                                               const uint32_t* __restrict__ near_list, int32_t* __restrict__ pairs,
                                                               int32_t* __restrict__ counts, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* touch = reinterpret_cast<uint32_t*>(smem);                  // [max_n1] lowest pending query touching idx_1
    int32_t* match = reinterpret_cast<int32_t*>(touch + max_n1);           // [max_n2] idx_1 matched to idx_2, -1 none
    uint8_t* claimed = reinterpret_cast<uint8_t*>(match + max_n2);         // [max_n1]
    uint8_t* pending = claimed + ((max_n1 + 15) & ~15);                    // [max_n2]
    __shared__ uint32_t s_flag[4];
    __shared__ unsigned long long s_red[kResolveThreads / 64];
    __shared__ uint32_t s_wave[kResolveThreads / 64];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p = blockIdx.x;
    const int n1 = n1_arr[p], n2 = n2_arr[p];
    const uint32_t* cnts = near_cnt + (size_t)p * max_n2;
    const uint32_t* lists = near_list + (size_t)p * max_n2 * kNearK;

    if (tid == 0) s_flag[0] = 0;
    for (int i = tid; i < n1; i += kResolveThreads) claimed[i] = 0;
    __syncthreads();
    uint32_t any_overflow = 0;
    for (int q = tid; q < n2; q += kResolveThreads) {
        const uint32_t c = cnts[q];
        match[q] = -1;
        pending[q] = c > 0;
        any_overflow |= c > (uint32_t)kNearK;
    }
    if (any_overflow) atomicOr(&s_flag[0], 1u);
    __syncthreads();

    if (s_flag[0] == 0) {
        // ---- parallel rounds (every round finalises at least the lowest pending query)
        for (int round = 0; round <= n2; ++round) {
            for (int i = tid; i < n1; i += kResolveThreads) touch[i] = 0xFFFFFFFFu;
            if (tid == 0) s_flag[1] = 0;
            __syncthreads();
            for (int q = tid; q < n2; q += kResolveThreads) {
                if (!pending[q]) continue;
                const uint32_t c = cnts[q];
                for (uint32_t k = 0; k < c; ++k) atomicMin(&touch[lists[(size_t)q * kNearK + k] & 0xFFFFu], (uint32_t)q);
            }
            __syncthreads();
            uint32_t still = 0;
            for (int q = tid; q < n2; q += kResolveThreads) {
                if (!pending[q]) continue;
                const uint32_t c = cnts[q];
                bool first = true;
                for (uint32_t k = 0; k < c; ++k) first &= touch[lists[(size_t)q * kNearK + k] & 0xFFFFu] == (uint32_t)q;
                if (!first) { still = 1; continue; }
                // every earlier query sharing a candidate is final: replay upstream's inner loop on the near list
                uint32_t best = OVS_MAX_HAMMING_DIST, second = OVS_MAX_HAMMING_DIST, best_idx = 0xFFFFFFFFu;
                for (uint32_t k = 0; k < c; ++k) {   // list is in ascending idx_1, as upstream scans
                    const uint32_t e = lists[(size_t)q * kNearK + k];
                    const uint32_t j = e & 0xFFFFu, d = e >> 16;
                    if (claimed[j]) continue;
                    if (d < best) { second = best; best = d; best_idx = j; }
                    else if (d < second) second = d;
                }
                pending[q] = 0;
                if (best_idx == 0xFFFFFFFFu || best > OVS_HAMMING_DIST_THR_LOW) continue;
                if (ratio_rejects(lowe_ratio, second, best)) continue;
                match[q] = (int32_t)best_idx;
                claimed[best_idx] = 1;
            }
            if (still) atomicOr(&s_flag[1], 1u);
            __syncthreads();
            if (s_flag[1] == 0) break;
            __syncthreads();
        }
    } else {
        // ---- literal serial replay (overflowing near lists): every query scans all unclaimed frame descriptors
        for (int q = 0; q < n2; ++q) {
            if (valid_2 && !valid_2[(size_t)p * (stride_2 / 32) + q]) continue;   // uniform
            uint32_t a[8];
            const uint32_t* src = reinterpret_cast<const uint32_t*>(desc_2 + (size_t)p * stride_2 + (size_t)q * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = src[i];
            // per-thread best/second over its strided subset, then a block reduction of the two smallest (d, idx) keys
            unsigned long long k1 = ~0ull, k2 = ~0ull;   // key = d << 32 | idx; k1 <= k2
            for (int j = tid; j < n1; j += kResolveThreads) {
                if (claimed[j]) continue;
                const uint32_t* b = reinterpret_cast<const uint32_t*>(desc_1 + (size_t)p * stride_1 + (size_t)j * 32);
                uint32_t d = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) d += __builtin_popcount(a[i] ^ b[i]);
                const unsigned long long key = ((unsigned long long)d << 32) | (uint32_t)j;
                if (key < k1) { k2 = k1; k1 = key; }
                else if (key < k2) k2 = key;
            }
            // reduce smallest key
            unsigned long long m1 = k1;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(m1, off);
                m1 = o < m1 ? o : m1;
            }
            if (lane == 0) s_red[wv] = m1;
            __syncthreads();
            unsigned long long g1 = ~0ull;
            for (int w = 0; w < kResolveThreads / 64; ++w) g1 = s_red[w] < g1 ? s_red[w] : g1;
            __syncthreads();
            // second smallest distance: smallest key different from g1
            unsigned long long m2 = (k1 == g1) ? k2 : k1;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(m2, off);
                m2 = o < m2 ? o : m2;
            }
            if (lane == 0) s_red[wv] = m2;
            __syncthreads();
            unsigned long long g2 = ~0ull;
            for (int w = 0; w < kResolveThreads / 64; ++w) g2 = s_red[w] < g2 ? s_red[w] : g2;
            __syncthreads();
            if (tid == 0 && g1 != ~0ull) {
                const uint32_t best = (uint32_t)(g1 >> 32), best_idx = (uint32_t)g1;
                const uint32_t second = g2 == ~0ull ? (uint32_t)OVS_MAX_HAMMING_DIST : (uint32_t)(g2 >> 32);
                // upstream initialises best = second = MAX_HAMMING_DIST and uses strict `<`: a distance of 256 never wins
                if (best < OVS_MAX_HAMMING_DIST && best <= OVS_HAMMING_DIST_THR_LOW &&
                    !ratio_rejects(lowe_ratio, second < OVS_MAX_HAMMING_DIST ? second : OVS_MAX_HAMMING_DIST, best)) {
                    match[q] = (int32_t)best_idx;
                    claimed[best_idx] = 1;
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- emit pairs in ascending idx_2 (upstream's emplace_back order)
    const int ipt = (n2 + kResolveThreads - 1) / kResolveThreads;
    const int b = tid * ipt, e = min(n2, b + ipt);
    uint32_t mine = 0;
    for (int q = b; q < e; ++q) mine += match[q] >= 0;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t2 = __shfl_up(incl, off);
        if (lane >= off) incl += t2;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    uint32_t pre = 0, total = 0;
    for (int w = 0; w < kResolveThreads / 64; ++w) {
        if (w < wv) pre += s_wave[w];
        total += s_wave[w];
    }
    uint32_t pos = pre + incl - mine;
    int32_t* out = pairs + (size_t)p * cap * 2;
    for (int q = b; q < e; ++q) {
        if (match[q] >= 0) {
            if (pos < (uint32_t)cap) {
                out[2 * pos] = match[q];
                out[2 * pos + 1] = q;
            }
            ++pos;
        }
    }
    if (tid == 0) counts[p] = (int32_t)(total < (uint32_t)cap ? total : (uint32_t)cap);
}

// ---- unconstrained best / second best ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hamming_best2(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt,
                                                      const uint8_t* __restrict__ t_valid, int32_t* __restrict__ best_idx,
                                                      uint16_t* __restrict__ best, uint16_t* __restrict__ second) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool active = i < nq;
    uint32_t a[8];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(q + (size_t)(active ? i : 0) * 32);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = src[k];
    const uint32_t* __restrict__ tt = reinterpret_cast<const uint32_t*>(t);
    uint32_t b = OVS_MAX_HAMMING_DIST, s = OVS_MAX_HAMMING_DIST;
    int bi = -1;
    for (int j = 0; j < nt; ++j) {
        if (t_valid && !t_valid[j]) continue;
        const uint32_t d = hamming256(a, tt + (size_t)j * 8);
        if (d < b) { s = b; b = d; bi = j; }
        else if (d < s) s = d;
    }
    if (active) {
        best_idx[i] = bi;
        best[i] = (uint16_t)b;
        second[i] = (uint16_t)s;
    }
}

}   // namespace ovs

using namespace ovs;

struct ovs_matcher {
    int device = 0;
    int max_n1 = 0, max_n2 = 0, max_batch = 0;
    hipStream_t stream = nullptr;
    uint32_t* d_near_cnt = nullptr;
    uint32_t* d_near_list = nullptr;
    // host-API staging (one problem)
    uint8_t* d_desc_1 = nullptr;
    uint8_t* d_desc_2 = nullptr;
    uint8_t* d_valid = nullptr;
    int32_t* d_n = nullptr;        // [2]
    int32_t* d_pairs = nullptr;
    int32_t* d_count = nullptr;
    int32_t* d_best_idx = nullptr;
    uint16_t* d_best = nullptr;
    uint16_t* d_second = nullptr;
    size_t resolve_lds = 0;
};

namespace {

// Largest distance that can still influence brute_force_match's accept/reject decision (see file header).
uint32_t near_threshold(float lowe_ratio) {
    uint32_t tau = 257;
    for (uint32_t s = 0; s <= 256; ++s) {
        const volatile float lhs = lowe_ratio * (float)s;   // one float multiply, as the device does (__fmul_rn)
        if (!(lhs < (float)OVS_HAMMING_DIST_THR_LOW)) { tau = s; break; }
    }
    uint32_t thr = std::max<uint32_t>(OVS_HAMMING_DIST_THR_LOW, tau - 1);
    return std::min<uint32_t>(thr, OVS_MAX_HAMMING_DIST);
}

ovs_status run_bf(ovs_matcher* m, const uint8_t* d1, size_t stride_1, const int32_t* d_n1, const uint8_t* d2, size_t stride_2,
                  const int32_t* d_n2, const uint8_t* d_valid, int batch, float lowe_ratio, int32_t* d_pairs, int32_t* d_counts, int cap,
                  hipStream_t s) {
    const uint32_t thr = near_threshold(lowe_ratio);
    dim3 grid((m->max_n2 + 255) / 256, batch);
    hipLaunchKernelGGL(k_hamming_near, grid, dim3(256), 0, s, d1, stride_1, d_n1, d2, stride_2, d_n2, d_valid, m->max_n2, thr,
                       m->d_near_cnt, m->d_near_list);
    OVS_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_bf_resolve, dim3(batch), dim3(kResolveThreads), m->resolve_lds, s, d1, stride_1, d_n1, d2, stride_2, d_n2,
                       d_valid, m->max_n1, m->max_n2, lowe_ratio, m->d_near_cnt, m->d_near_list, d_pairs, d_counts, cap);
    OVS_HIP_TRY(hipGetLastError());
    return OVS_OK;
}

}   // namespace

extern "C" {

ovs_status ovs_matcher_create(int32_t max_n1, int32_t max_n2, int32_t max_batch, int32_t device, ovs_matcher** out) {
    if (!out || max_n1 < 1 || max_n2 < 1 || max_batch < 1 || max_n1 > 65535 || max_n2 > 65535) return OVS_ERR_INVALID;
    *out = nullptr;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    ovs_matcher* m = new (std::nothrow) ovs_matcher();
    if (!m) return OVS_ERR_INVALID;
    m->device = device;
    m->max_n1 = max_n1;
    m->max_n2 = max_n2;
    m->max_batch = max_batch;
    m->resolve_lds = (size_t)max_n1 * 4 + (size_t)max_n2 * 4 + (((size_t)max_n1 + 15) & ~(size_t)15) + (((size_t)max_n2 + 15) & ~(size_t)15);
    if (m->resolve_lds > 150 * 1024) {
        delete m;
        return OVS_ERR_CAPACITY;
    }
#define CREATE_TRY(expr)                       \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) {                \
            ovs::set_last_error(#expr, _e);    \
            ovs_matcher_destroy(m);            \
            return OVS_ERR_HIP;                \
        }                                      \
    } while (0)
    CREATE_TRY(hipSetDevice(device));
    CREATE_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    const size_t B = (size_t)max_batch;
    CREATE_TRY(hipMalloc(&m->d_near_cnt, sizeof(uint32_t) * B * max_n2));
    CREATE_TRY(hipMalloc(&m->d_near_list, sizeof(uint32_t) * B * max_n2 * kNearK));
    CREATE_TRY(hipMalloc(&m->d_desc_1, (size_t)max_n1 * 32));
    CREATE_TRY(hipMalloc(&m->d_desc_2, (size_t)max_n2 * 32));
    CREATE_TRY(hipMalloc(&m->d_valid, (size_t)std::max(max_n1, max_n2)));
    CREATE_TRY(hipMalloc(&m->d_n, sizeof(int32_t) * 2));
    CREATE_TRY(hipMalloc(&m->d_pairs, sizeof(int32_t) * 2 * max_n2));
    CREATE_TRY(hipMalloc(&m->d_count, sizeof(int32_t)));
    CREATE_TRY(hipMalloc(&m->d_best_idx, sizeof(int32_t) * max_n2));
    CREATE_TRY(hipMalloc(&m->d_best, sizeof(uint16_t) * max_n2));
    CREATE_TRY(hipMalloc(&m->d_second, sizeof(uint16_t) * max_n2));
    if (m->resolve_lds > 64 * 1024)
        CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bf_resolve), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)m->resolve_lds));
#undef CREATE_TRY
    *out = m;
    return OVS_OK;
}

ovs_status ovs_matcher_destroy(ovs_matcher* m) {
    if (!m) return OVS_OK;
    if (m->stream) hipStreamSynchronize(m->stream);
    hipFree(m->d_near_cnt);
    hipFree(m->d_near_list);
    hipFree(m->d_desc_1);
    hipFree(m->d_desc_2);
    hipFree(m->d_valid);
    hipFree(m->d_n);
    hipFree(m->d_pairs);
    hipFree(m->d_count);
    hipFree(m->d_best_idx);
    hipFree(m->d_best);
    hipFree(m->d_second);
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
    return OVS_OK;
}

ovs_status ovs_robust_brute_force_match_batch_dev(ovs_matcher* m, const uint8_t* d_desc_1, size_t stride_1, const int32_t* d_n1,
                                                  const uint8_t* d_desc_2, size_t stride_2, const int32_t* d_n2,
                                                  const uint8_t* d_valid_2, int32_t batch, float lowe_ratio, int32_t* d_pairs,
                                                  int32_t* d_counts, int32_t cap, void* stream) {
    if (!m || !d_desc_1 || !d_desc_2 || !d_n1 || !d_n2 || !d_pairs || !d_counts || batch < 1 || cap < 1) return OVS_ERR_INVALID;
    if (batch > m->max_batch || stride_1 > (size_t)m->max_n1 * 32 || stride_2 > (size_t)m->max_n2 * 32) return OVS_ERR_CAPACITY;
    if (((uintptr_t)d_desc_1 & 3) || ((uintptr_t)d_desc_2 & 3) || (stride_1 & 31) || (stride_2 & 31)) return OVS_ERR_ALIGN;
    OVS_HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = stream ? (hipStream_t)stream : m->stream;
    return run_bf(m, d_desc_1, stride_1, d_n1, d_desc_2, stride_2, d_n2, d_valid_2, batch, lowe_ratio, d_pairs, d_counts, cap, s);
}

ovs_status ovs_robust_brute_force_match(ovs_matcher* m, const uint8_t* desc_1, int32_t n1, const uint8_t* desc_2, int32_t n2,
                                        const uint8_t* valid_2, float lowe_ratio, int32_t* pairs, int32_t cap, int32_t* n_out) {
    if (!m || !n_out || n1 < 0 || n2 < 0 || cap < 0) return OVS_ERR_INVALID;
    *n_out = 0;
    if (n1 == 0 || n2 == 0) return OVS_OK;
    if (!desc_1 || !desc_2 || (cap > 0 && !pairs)) return OVS_ERR_INVALID;
    if (n1 > m->max_n1 || n2 > m->max_n2) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const int32_t nn[2] = {n1, n2};
    OVS_HIP_TRY(hipMemcpyAsync(m->d_desc_1, desc_1, (size_t)n1 * 32, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(m->d_desc_2, desc_2, (size_t)n2 * 32, hipMemcpyHostToDevice, s));
    if (valid_2) OVS_HIP_TRY(hipMemcpyAsync(m->d_valid, valid_2, (size_t)n2, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(m->d_n, nn, sizeof(nn), hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));   // nn is a stack array
    ovs_status st = run_bf(m, m->d_desc_1, (size_t)m->max_n1 * 32, m->d_n, m->d_desc_2, (size_t)m->max_n2 * 32, m->d_n + 1,
                           valid_2 ? m->d_valid : nullptr, 1, lowe_ratio, m->d_pairs, m->d_count, m->max_n2, s);
    if (st != OVS_OK) return st;
    int32_t n = 0;
    OVS_HIP_TRY(hipMemcpyAsync(&n, m->d_count, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    const int32_t k = std::min(n, cap);
    if (k > 0) {
        OVS_HIP_TRY(hipMemcpyAsync(pairs, m->d_pairs, sizeof(int32_t) * 2 * k, hipMemcpyDeviceToHost, s));
        OVS_HIP_TRY(hipStreamSynchronize(s));
    }
    *n_out = k;
    return n > cap ? OVS_ERR_CAPACITY : OVS_OK;
}

ovs_status ovs_hamming_best2(ovs_matcher* m, const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt, const uint8_t* t_valid,
                             int32_t* best_idx, uint16_t* best, uint16_t* second) {
    if (!m || nq < 0 || nt < 0) return OVS_ERR_INVALID;
    if (nq == 0) return OVS_OK;
    if (!q || (nt > 0 && !t) || !best_idx || !best || !second) return OVS_ERR_INVALID;
    // queries ride in the idx_2 buffers, targets in the idx_1 buffers
    if (nq > m->max_n2 || nt > m->max_n1) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    OVS_HIP_TRY(hipMemcpyAsync(m->d_desc_2, q, (size_t)nq * 32, hipMemcpyHostToDevice, s));
    if (nt > 0) OVS_HIP_TRY(hipMemcpyAsync(m->d_desc_1, t, (size_t)nt * 32, hipMemcpyHostToDevice, s));
    if (t_valid && nt > 0) OVS_HIP_TRY(hipMemcpyAsync(m->d_valid, t_valid, (size_t)nt, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_hamming_best2, dim3((nq + 255) / 256), dim3(256), 0, s, m->d_desc_2, nq, m->d_desc_1, nt,
                       (t_valid && nt > 0) ? m->d_valid : nullptr, m->d_best_idx, m->d_best, m->d_second);
    OVS_HIP_TRY(hipGetLastError());
    OVS_HIP_TRY(hipMemcpyAsync(best_idx, m->d_best_idx, sizeof(int32_t) * nq, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(best, m->d_best, sizeof(uint16_t) * nq, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(second, m->d_second, sizeof(uint16_t) * nq, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    return OVS_OK;
}

}   // extern "C"
