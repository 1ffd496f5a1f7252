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

constexpr int kNearSplit = 4;         // waves per workgroup; each scans a quarter of the frame descriptors for the same 64 queries
constexpr int kNearSeg = 64;          // near-list capacity per (query, wave); beyond it: cooperative full scan, still exact
constexpr int kTopK = 8;              // sorted smallest keys kept per query for the resolver's fast path

__device__ __forceinline__ uint32_t hamming256(const uint32_t (&a)[8], const uint32_t* __restrict__ b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d = __builtin_popcount(a[i] ^ b[i]) + d;
    return d;
}

// keep the kTopK smallest keys (ascending) -- key = d << 16 | idx_1, i.e. (distance, first-seen) order
__device__ __forceinline__ void topk_insert(uint32_t e, uint32_t (&t)[kTopK]) {
    if (e < t[kTopK - 1]) {
        t[kTopK - 1] = e;
#pragma unroll
        for (int k = kTopK - 1; k > 0; --k) {
            const uint32_t lo = min(t[k - 1], t[k]), hi = max(t[k - 1], t[k]);
            t[k - 1] = lo;
            t[k] = hi;
        }
    }
}

// ---- 1. all pairs, near lists -------------------------------------------------------------------------------------
// The O(n1 * n2) part runs on the matrix cores. A 256-bit Hamming distance is an exact integer dot product: with the frame
// descriptor a expanded to {0, 1} bytes and the keyframe descriptor b to {+1, -1} bytes (bit set -> +1),
//     sum_k a_k * b_k = |a & b| - |a & ~b| = 2 |a & b| - |a|,   d(a, b) = |a| + |b| - 2 |a & b| = |b| - sum_k a_k * b_k,
// so one v_mfma_i32_32x32x32_i8 chain of 8 steps (K = 256) yields the 32 x 32 distances of a tile, bit-exact in int32. The kernel
// is compute-bound (n1 pairs per 32 bytes of keyframe descriptor), which is what the matrix core is for; nothing is approximated.
//   * Workgroup = 4 waves x 64 queries (idx_2) each = 256 queries; every wave scans ALL frame descriptors (idx_1) in tiles of 32.
//   * The wave's 64 queries live in registers as two B operands (2 x 8 steps x 4 VGPRs of +-1 bytes), expanded once.
//   * Per 32-descriptor tile a lane loads 16 raw bytes (row = lane & 31, half = lane >> 5), expands them step by step into the
//     A operand (4 VGPRs of 0/1 bytes: shift + and per VGPR -- the order of the 256 bit positions along K is free as long as A
//     and B use the same one, so byte v of a VGPR takes bit (base + v) of each of the word's four bytes) and feeds TWO MFMAs
//     (one per query tile). 64 + ~20 VALU and 16 MFMA per 2048 pairs, against 16 VALU per PAIR on the vector path.
//   * C layout (guide 3, fragment layout): lane holds column (query) lane & 31, rows (frame keypoints) (r & 3) + 8 (r >> 2) +
//     4 (lane >> 5). A pair is near iff acc >= |b| - near_thr: one v_max3 per two pairs and one branch per tile on the common path.
//   * Near pairs (rare) go to the query's list; the list keeps its four segments so the resolver's layout is unchanged (now:
//     C-tile half x first / second half of the tiles -- one writer lane per segment, so the fill counts are registers). At the end
//     each thread builds one query's sorted top-8 from the entries its own wave wrote.
// History (round 2, config 2, 128 problems of 2000 x 2000): scalar-cache + v_bcnt vector path 0.303 ms (0.57 of its VALU issue
// floor); LDS-broadcast vector path 0.394 ms; see DESIGN.md for this kernel's numbers.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kNearQueries = 256;   // queries per workgroup (4 waves x 2 tiles of 32)
constexpr int kQueueSlots = 128;    // per-wave ring of queued C-tile columns (drained 64 at a time)
static_assert(kNearSplit == 4, "segment counts are stored as one uint4 per query");

// step s of the K loop covers bits (s & 1) * 4 + v (v = 0..3: the VGPR) of every byte of word s >> 1 of the lane's 16-byte half
__device__ __forceinline__ v4i expand01(uint32_t w, int base) {
    v4i r;
    r[0] = (int)((w >> base) & 0x01010101u);
    r[1] = (int)((w >> (base + 1)) & 0x01010101u);
    r[2] = (int)((w >> (base + 2)) & 0x01010101u);
    r[3] = (int)((w >> (base + 3)) & 0x01010101u);
    return r;
}
// 0/1 bytes -> +1 / -1 bytes: 0xFF - 0xFE * z per byte (no carries: every byte product is <= 0xFE)
__device__ __forceinline__ v4i to_pm1(v4i z) {
    v4i r;
#pragma unroll
    for (int v = 0; v < 4; ++v) r[v] = (int)~((uint32_t)z[v] * 0xFEu);
    return r;
}
__device__ __forceinline__ int max16(const v16i& c) {
    return max(max(max(max(c[0], c[1]), max(c[2], c[3])), max(max(c[4], c[5]), max(c[6], c[7]))),
               max(max(max(c[8], c[9]), max(c[10], c[11])), max(max(c[12], c[13]), max(c[14], c[15]))));
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_hamming_near(const uint8_t* __restrict__ desc_1, size_t stride_1,
                                                     const int32_t* __restrict__ n1_arr, const uint8_t* __restrict__ desc_2,
                                                     size_t stride_2, const int32_t* __restrict__ n2_arr,
                                                     const uint8_t* __restrict__ valid_2, int max_n2, uint32_t near_thr,
                                                     uint32_t* __restrict__ near_cnt, uint32_t* __restrict__ near_list,
                                                     uint32_t* __restrict__ near_top, int chunks, int total_wg) {
    __shared__ uint32_t s_segcnt[kNearSplit][kNearQueries];
    __shared__ int4 s_queue[4][4][kQueueSlots];    // per wave: queued C-tile columns, [quarter of the 16 accumulators][slot]
    __shared__ uint32_t s_qmeta[4][kQueueSlots];   // tile << 8 | half << 6 | query of the wave
    __shared__ int2 s_ctx[4][64];                  // per query of the wave: acceptance bound, |b|
    // XCD-major work order (workgroup b runs on XCD b % 8): XCD k takes the k-th contiguous eighth of the (problem, chunk) sequence,
    // so the chunks of one problem share one L2
    const int per_xcd = gridDim.x >> 3;
    const int wg = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (wg >= total_wg) return;
    const int p = wg / chunks, chunk_id = wg - p * chunks;
    const int n1 = n1_arr[p], n2 = n2_arr[p];
    if (chunk_id * kNearQueries >= n2) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, half = lane >> 5;
#pragma unroll
    for (int w = 0; w < kNearSplit; ++w) s_segcnt[w][tid] = 0;
    __syncthreads();

    // ---- this wave's 64 queries -> two B operands (+-1 bytes), |b| and the per-lane acceptance bound
    const int q_wave = chunk_id * kNearQueries + wv * 64;
    v4i qb[2][8];
    int need[2], pcq[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = q_wave + u * 32 + col;
        const bool act = q < n2 && (!valid_2 || valid_2[(size_t)p * (stride_2 / 32) + q]);
        const uint4* src = reinterpret_cast<const uint4*>(desc_2 + (size_t)p * stride_2 + (size_t)(q < n2 ? q : 0) * 32);
        const uint4 lo = src[0], hi = src[1];
        pcq[u] = __builtin_popcount(lo.x) + __builtin_popcount(lo.y) + __builtin_popcount(lo.z) + __builtin_popcount(lo.w) +
                 __builtin_popcount(hi.x) + __builtin_popcount(hi.y) + __builtin_popcount(hi.z) + __builtin_popcount(hi.w);
        const uint4 mine = half ? hi : lo;
        const uint32_t w4[4] = {mine.x, mine.y, mine.z, mine.w};
#pragma unroll
        for (int s = 0; s < 8; ++s) qb[u][s] = to_pm1(expand01(w4[s >> 1], (s & 1) * 4));
        need[u] = act ? pcq[u] - (int)near_thr : 0x7FFFFFFF;   // inactive query: no accumulator value reaches the bound
    }
    if (half == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) s_ctx[wv][u * 32 + col] = make_int2(need[u], pcq[u]);
    }
    // s_ctx / s_queue / s_qmeta are written by some lanes of a wave and read by OTHER lanes of the same wave: a wavefront-scope release /
    // acquire pair plus a wave barrier states that hand-over to the compiler (no instruction is emitted: a wave's DS operations execute
    // in order), instead of leaving it to the fact that it cannot prove the dynamic indices distinct.
    auto wave_handover = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    wave_handover();
    // Near pairs are rare per PAIR (about 3 per 1000 on consecutive video frames) but nearly every 32 x 32 tile holds one, so a
    // per-tile "any lane has one" branch into row-by-row tests would run on every tile with one or two useful lanes. Instead a lane
    // whose tile column holds a near pair appends its 16 accumulators to a wave-level queue in LDS (slots compacted with the ballot,
    // no row tests on this path), and when 64 columns are queued the wave drains them with every lane busy: lane k takes entry k,
    // counts its near rows, reserves that many slots of the query's list segment with ONE LDS atomic and stores the entries.
    // The list of a query keeps four segments (the resolver's layout): segment = 2 * (C-tile half the column came from) + (0 / 1 for
    // the first / second half of the tiles).
    const int n_tiles = (n1 + 31) >> 5, phase_tiles = (n_tiles + 1) >> 1;
    int q_head = 0, q_tail = 0;   // wave-uniform ring indices (entries q_head .. q_tail - 1, slots taken mod kQueueSlots)
    auto drain = [&](int n_take) __attribute__((always_inline)) {
        wave_handover();   // the queue entries stored by the producing lanes -> the draining lanes
        if (lane < n_take) {
            const int e = (q_head + lane) & (kQueueSlots - 1);
            const uint32_t meta = s_qmeta[wv][e];
            const int tile_e = (int)(meta >> 8), half_e = (int)(meta >> 6) & 1, qi = (int)(meta & 63u);
            const int2 ctx = s_ctx[wv][qi];
            int v[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int4 x = s_queue[wv][k][e];
                v[4 * k] = x.x; v[4 * k + 1] = x.y; v[4 * k + 2] = x.z; v[4 * k + 3] = x.w;
            }
            uint32_t nh = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) nh += v[r] >= ctx.x ? 1u : 0u;
            const int seg = 2 * half_e + (tile_e >= phase_tiles ? 1 : 0);
            uint32_t slot = atomicAdd(&s_segcnt[seg][wv * 64 + qi], nh);
            uint32_t* dst = near_list + (((size_t)p * max_n2 + (q_wave + qi)) * kNearSplit + seg) * kNearSeg;
            const int j_base = tile_e * 32 + 4 * half_e;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (v[r] >= ctx.x) {
                    // past kNearSeg the segment counts as overflowed (the resolver then rescans that query): the clamp only keeps the store in bounds
                    dst[min(slot, (uint32_t)kNearSeg - 1u)] = ((uint32_t)(ctx.y - v[r]) << 16) | (uint32_t)(j_base + (r & 3) + 8 * (r >> 2));
                    ++slot;
                }
            }
        }
        wave_handover();   // the drained slots are free for the next column sets only after every lane has read its entry
        q_head += n_take;
    };

    const uint8_t* __restrict__ t = desc_1 + (size_t)p * stride_1;
    auto load_rows = [&](int j0) __attribute__((always_inline)) -> uint4 {
        const int row = min(j0 + col, n1 - 1);   // rows past n1 re-read the last descriptor; their accumulators are voided below
        return *reinterpret_cast<const uint4*>(t + (size_t)row * 32 + half * 16);
    };
    if (n1 > 0) {
        // The workgroups of a problem run side by side on one XCD and would all ask L2 for the same descriptor lines at the same moment
        // (each such miss goes out to the fabric: 5x the algorithmic reads were measured): every chunk starts its sweep at a different
        // tile, so a line is first touched by one workgroup and found in L2 by the others.
        const int tile_first = (chunk_id * n_tiles) / chunks;
        uint4 raw = load_rows(tile_first << 5);
        for (int it = 0; it < n_tiles; ++it) {
            const int tile = it + tile_first - (it + tile_first >= n_tiles ? n_tiles : 0);
            const int tile_next = tile + 1 == n_tiles ? 0 : tile + 1;
            const int j0 = tile << 5;
            const uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w};
            if (it + 1 < n_tiles) raw = load_rows(tile_next << 5);   // next tile's bytes are in flight under this tile's MFMAs
            v16i acc[2];
            acc[0] = (v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            acc[1] = acc[0];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const v4i a = expand01(w4[s >> 1], (s & 1) * 4);
                acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qb[0][s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qb[1][s], acc[1], 0, 0, 0);
            }
            if (j0 + 32 > n1) {   // last, partial tile (wave-uniform): rows past n1 can never be near
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool in = j0 + (r & 3) + 8 * (r >> 2) + 4 * half < n1;
                    acc[0][r] = in ? acc[0][r] : (int)0x80000000;
                    acc[1][r] = in ? acc[1][r] : (int)0x80000000;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool has = max16(acc[u]) >= need[u];
                const unsigned long long m = __builtin_amdgcn_ballot_w64(has);
                if (m) {
                    if (has) {
                        const int e = (q_tail + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) &
                                      (kQueueSlots - 1);
                        s_qmeta[wv][e] = ((uint32_t)tile << 8) | (uint32_t)(half << 6) | (uint32_t)(u * 32 + col);
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            s_queue[wv][k][e] = make_int4(acc[u][4 * k], acc[u][4 * k + 1], acc[u][4 * k + 2], acc[u][4 * k + 3]);
                    }
                    q_tail += __builtin_popcountll(m);
                    if (q_tail - q_head >= 64) drain(64);   // at most 63 were queued before this column set: never more than 127 entries
                }
            }
        }
        if (q_tail > q_head) drain(q_tail - q_head);
    }
    // Workgroup scope is enough (and an agent-scope fence would cost an L2 write-back + invalidate per workgroup: buffer_wbl2 sc1 /
    // buffer_inv sc1, which also evicts the descriptors every other workgroup of the XCD is streaming): the entries read back below
    // were written by lanes of the reader's OWN wave, through the CU's write-through L1.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    // ---- per query: segment counts and the sorted top-8 of its list. Thread tid <-> query tid of the workgroup; eight entries per
    // segment in flight at a time.
    const int q = chunk_id * kNearQueries + tid;
    if (q < n2) {
        uint32_t top[kTopK];
#pragma unroll
        for (int k = 0; k < kTopK; ++k) top[k] = ~0u;
        uint32_t c[kNearSplit];
#pragma unroll
        for (int w = 0; w < kNearSplit; ++w) c[w] = s_segcnt[w][tid];
        const uint32_t* lst = near_list + ((size_t)p * max_n2 + q) * kNearSplit * kNearSeg;
        const uint32_t longest = min((uint32_t)kNearSeg, max(max(c[0], c[1]), max(c[2], c[3])));
        for (uint32_t k0 = 0; k0 < longest; k0 += 8) {
            uint32_t e[kNearSplit][8];
#pragma unroll
            for (int w = 0; w < kNearSplit; ++w)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    e[w][i] = k0 + i < min(c[w], (uint32_t)kNearSeg)
                                  ? __hip_atomic_load(lst + w * kNearSeg + k0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                  : ~0u;
#pragma unroll
            for (int w = 0; w < kNearSplit; ++w)
#pragma unroll
                for (int i = 0; i < 8; ++i) topk_insert(e[w][i], top);
        }
        *reinterpret_cast<uint4*>(near_cnt + ((size_t)p * max_n2 + q) * kNearSplit) = make_uint4(c[0], c[1], c[2], c[3]);
        uint4* dst = reinterpret_cast<uint4*>(near_top + ((size_t)p * max_n2 + q) * kTopK);
        dst[0] = make_uint4(top[0], top[1], top[2], top[3]);
        dst[1] = make_uint4(top[4], top[5], top[6], top[7]);
    }
}

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// Four consecutive 32-byte descriptors through the scalar data cache under ONE wait (hipcc otherwise waits after each load).
// The loads, their count and the wait are all inside the statement (guide 5.7); outputs are early-clobber SGPR tuples.
__device__ __forceinline__ void sload_desc4(const uint32_t* __restrict__ p, u32x8& b0, u32x8& b1, u32x8& b2, u32x8& b3) {
    asm volatile(
        "s_load_dwordx8 %0, %4, 0x0\n\t"
        "s_load_dwordx8 %1, %4, 0x20\n\t"
        "s_load_dwordx8 %2, %4, 0x40\n\t"
        "s_load_dwordx8 %3, %4, 0x60\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(b0), "=&s"(b1), "=&s"(b2), "=&s"(b3)
        : "s"(p)
        : "memory");
}

// eight descriptors under one wait: half as many stalls per pair as sload_desc4 (64 SGPRs of payload)
__device__ __forceinline__ void sload_desc8(const uint32_t* __restrict__ p, u32x8& b0, u32x8& b1, u32x8& b2, u32x8& b3, u32x8& b4, u32x8& b5,
                                            u32x8& b6, u32x8& b7) {
    asm volatile(
        "s_load_dwordx8 %0, %8, 0x0\n\t"
        "s_load_dwordx8 %1, %8, 0x20\n\t"
        "s_load_dwordx8 %2, %8, 0x40\n\t"
        "s_load_dwordx8 %3, %8, 0x60\n\t"
        "s_load_dwordx8 %4, %8, 0x80\n\t"
        "s_load_dwordx8 %5, %8, 0xa0\n\t"
        "s_load_dwordx8 %6, %8, 0xc0\n\t"
        "s_load_dwordx8 %7, %8, 0xe0\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(b0), "=&s"(b1), "=&s"(b2), "=&s"(b3), "=&s"(b4), "=&s"(b5), "=&s"(b6), "=&s"(b7)
        : "s"(p)
        : "memory");
}

__device__ __forceinline__ uint32_t hamming256v(const uint32_t (&a)[8], const u32x8& b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d = __builtin_popcount(a[i] ^ b[i]) + d;
    return d;
}

// ---- 1b. all pairs, near lists: the popcount path (OVS_NEAR_PATH_POPCOUNT) ---------------------------------------------
// The vector-ALU form of the same stage (BASELINE north star: "Hamming distance uses popcount"), kept selectable and under the same
// parity tests; it is the kernel `roofline_valu` is quoted on. Same outputs as k_hamming_near (a query's four list segments are the
// four waves' quarters of idx_1 here), 2.3x slower on config 2.
// Workgroup = 4 waves x the same 64 queries (idx_2); wave w scans the w-th quarter of the frame descriptors (idx_1). The descriptors
// of the quarter stream through the scalar cache in groups of FOUR into two SGPR sets that are double-buffered: the loads of the next
// group are in flight while the current one is XORed / popcounted (SMEM returns out of order, so only lgkmcnt(0) is a usable wait --
// the overlap has to come from issuing early, not from partial waits). Entries with d <= near_thr go to the wave's own segment of the
// query's list (count in a register: no atomics, no waits in the loop) and into a sorted top-8 kept in registers; the four partial
// top-8s are merged through LDS at the end.
//
// v2 (round 2): (i) the popcount accumulate is pinned to the 8 x (v_xor, v_bcnt acc) chain -- hipcc split it into 6 independent
// v_bcnt + 3 v_add3_u32 per descriptor (19 instead of 16 VALU per pair); (ii) 1-D grid in XCD-major order, so all chunks of a problem
// run on ONE XCD and its 64 KB of frame descriptors are fetched into one L2 instead of eight (fabric traffic was 9.7x algorithmic);
// (iii) the double-buffered scalar loads above. 0.303 ms per 128 problems of 2000 x 2000 = 0.57 of the VALU issue floor.
//
// v3 (round 2, measured and NOT kept): frame descriptors staged in LDS and read back with broadcast ds_read_b128 (256 queries per
// workgroup, one self-contained asm block per descriptor: two LDS reads of the next descriptor, 8 x (v_xor, v_bcnt), lgkmcnt(0)).
// Bit-identical results, but 0.394 ms (0.44 of the floor): the LDS return has to be waited for inside every 16-instruction block
// (a wait placed in the NEXT statement lets hipcc copy VGPRs whose data has not landed), and a 16-VALU block is shorter than the
// ds_read_b128 round trip. The scalar-cache form stays.

// One word of four descriptors in ONE asm volatile statement: four v_xor_b32 (SGPR operand) followed by the four v_bcnt_u32_b32 that
// accumulate them, so no v_bcnt issues directly behind the v_xor it depends on (hipcc pairs them back to back), the popcount stays an
// 8-deep accumulate chain per descriptor (hipcc splits it into 6 independent v_bcnt + 3 v_add3_u32: 19 instead of 16 VALU per pair),
// and -- volatile statements keep their program order -- the machine scheduler cannot sink the NEXT group's scalar loads below this
// group's arithmetic.
__device__ __forceinline__ void xor_bcnt4_first(uint32_t a, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3, uint32_t& d0, uint32_t& d1,
                                                uint32_t& d2, uint32_t& d3) {
    uint32_t t0, t1, t2, t3;
    asm volatile(
        "v_xor_b32 %4, %9, %8\n\tv_xor_b32 %5, %10, %8\n\tv_xor_b32 %6, %11, %8\n\tv_xor_b32 %7, %12, %8\n\t"
        "v_bcnt_u32_b32 %0, %4, 0\n\tv_bcnt_u32_b32 %1, %5, 0\n\tv_bcnt_u32_b32 %2, %6, 0\n\tv_bcnt_u32_b32 %3, %7, 0"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(a), "s"(s0), "s"(s1), "s"(s2), "s"(s3));
}
__device__ __forceinline__ void xor_bcnt4_acc(uint32_t a, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3, uint32_t& d0, uint32_t& d1,
                                              uint32_t& d2, uint32_t& d3) {
    uint32_t t0, t1, t2, t3;
    asm volatile(
        "v_xor_b32 %4, %9, %8\n\tv_xor_b32 %5, %10, %8\n\tv_xor_b32 %6, %11, %8\n\tv_xor_b32 %7, %12, %8\n\t"
        "v_bcnt_u32_b32 %0, %4, %0\n\tv_bcnt_u32_b32 %1, %5, %1\n\tv_bcnt_u32_b32 %2, %6, %2\n\tv_bcnt_u32_b32 %3, %7, %3"
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(a), "s"(s0), "s"(s1), "s"(s2), "s"(s3));
}

// SGPR sets pinned to fixed registers: the load statement and the wait statement must name the SAME physical registers (the loads land
// asynchronously; a compiler-inserted copy between the two statements would copy stale values)
#define OVS_SET_A0 "s[36:43]"
#define OVS_SET_A1 "s[44:51]"
#define OVS_SET_A2 "s[52:59]"
#define OVS_SET_A3 "s[60:67]"
#define OVS_SET_B0 "s[68:75]"
#define OVS_SET_B1 "s[76:83]"
#define OVS_SET_B2 "s[84:91]"
#define OVS_SET_B3 "s[92:99]"

#define OVS_ISSUE4(p, r0, r1, r2, r3, R0, R1, R2, R3)                                                                       \
    asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x20\n\ts_load_dwordx8 %2, %4, 0x40\n\ts_load_dwordx8 %3, %4, 0x60" \
                 : "={" R0 "}"(r0), "={" R1 "}"(r1), "={" R2 "}"(r2), "={" R3 "}"(r3)                                        \
                 : "s"(p)                                                                                                    \
                 : "memory")
#define OVS_WAIT4(r0, r1, r2, r3, R0, R1, R2, R3) \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+{" R0 "}"(r0), "+{" R1 "}"(r1), "+{" R2 "}"(r2), "+{" R3 "}"(r3)::"memory")

__device__ __forceinline__ void dist4(const uint32_t (&a)[8], const u32x8& b0, const u32x8& b1, const u32x8& b2, const u32x8& b3, uint32_t& d0,
                                      uint32_t& d1, uint32_t& d2, uint32_t& d3) {
    xor_bcnt4_first(a[0], b0[0], b1[0], b2[0], b3[0], d0, d1, d2, d3);
#pragma unroll
    for (int i = 1; i < 8; ++i) xor_bcnt4_acc(a[i], b0[i], b1[i], b2[i], b3[i], d0, d1, d2, d3);
}

__global__ __launch_bounds__(256) void k_hamming_near_popc(const uint8_t* __restrict__ desc_1, size_t stride_1,
                                                     const int32_t* __restrict__ n1_arr, const uint8_t* __restrict__ desc_2,
                                                     size_t stride_2, const int32_t* __restrict__ n2_arr,
                                                     const uint8_t* __restrict__ valid_2, int max_n2, uint32_t near_thr,
                                                     uint32_t* __restrict__ near_cnt, uint32_t* __restrict__ near_list,
                                                     uint32_t* __restrict__ near_top, int chunks, int total_wg) {
    __shared__ uint32_t s_top[kNearSplit][kTopK][64];
    // XCD-major work order (workgroup b runs on XCD b % 8): XCD k takes the k-th contiguous eighth of the (problem, chunk) sequence
    const int per_xcd = gridDim.x >> 3;
    const int wg = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (wg >= total_wg) return;
    const int p = wg / chunks, chunk_id = wg - p * chunks;
    const int n1 = n1_arr[p], n2 = n2_arr[p];
    if (chunk_id * 64 >= n2) return;
    // wave index made provably uniform so the descriptor addresses below stay scalar (s_load_dwordx8)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = chunk_id * 64 + lane;
    const bool active = q < n2 && (!valid_2 || valid_2[(size_t)p * (stride_2 / 32) + q]);
    uint32_t a[8];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(desc_2 + (size_t)p * stride_2 + (size_t)(q < n2 ? q : 0) * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = src[i];
    }
    const int chunk = (((n1 + kNearSplit - 1) / kNearSplit) + 7) & ~7;
    const int jb = min(n1, wv * chunk), je = min(n1, jb + chunk);
    const uint32_t* __restrict__ t = reinterpret_cast<const uint32_t*>(desc_1 + (size_t)p * stride_1);
    uint32_t* my_seg = near_list + (((size_t)p * max_n2 + q) * kNearSplit + wv) * kNearSeg;
    uint32_t top[kTopK];
#pragma unroll
    for (int k = 0; k < kTopK; ++k) top[k] = ~0u;
    uint32_t cnt = 0;
    auto hit = [&](uint32_t d, int j) {
        const uint32_t e = (d << 16) | (uint32_t)j;
        if (cnt < (uint32_t)kNearSeg) my_seg[cnt] = e;
        ++cnt;
        topk_insert(e, top);
    };
    auto group = [&](uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, int j) {
        // near distances are rare (the true match and near-duplicates): one test per four pairs on the common path
        if (min(min(d0, d1), min(d2, d3)) <= near_thr) {
            if (d0 <= near_thr) hit(d0, j);
            if (d1 <= near_thr) hit(d1, j + 1);
            if (d2 <= near_thr) hit(d2, j + 2);
            if (d3 <= near_thr) hit(d3, j + 3);
        }
    };
    int j = jb;
    if (active) {
        const int n8 = (je - jb) >> 3;   // iterations of 8 descriptors = one A group + one B group
        if (n8 > 0) {
            u32x8 a0, a1, a2, a3, b0, b1, b2, b3;
            const uint32_t* pa = t + (size_t)j * 8;
            OVS_ISSUE4(pa, a0, a1, a2, a3, OVS_SET_A0, OVS_SET_A1, OVS_SET_A2, OVS_SET_A3);
            for (int it = 0; it < n8; ++it, j += 8) {
                const uint32_t* pb = t + (size_t)(j + 4) * 8;
                OVS_WAIT4(a0, a1, a2, a3, OVS_SET_A0, OVS_SET_A1, OVS_SET_A2, OVS_SET_A3);
                OVS_ISSUE4(pb, b0, b1, b2, b3, OVS_SET_B0, OVS_SET_B1, OVS_SET_B2, OVS_SET_B3);
                uint32_t d0, d1, d2, d3;
                dist4(a, a0, a1, a2, a3, d0, d1, d2, d3);
                group(d0, d1, d2, d3, j);
                OVS_WAIT4(b0, b1, b2, b3, OVS_SET_B0, OVS_SET_B1, OVS_SET_B2, OVS_SET_B3);
                if (it + 1 < n8) {   // wave-uniform: the next A group (the last iteration has none to fetch)
                    const uint32_t* pn = t + (size_t)(j + 8) * 8;
                    OVS_ISSUE4(pn, a0, a1, a2, a3, OVS_SET_A0, OVS_SET_A1, OVS_SET_A2, OVS_SET_A3);
                }
                dist4(a, b0, b1, b2, b3, d0, d1, d2, d3);
                group(d0, d1, d2, d3, j + 4);
            }
        }
        for (; j + 4 <= je; j += 4) {   // remainder: four at a time
            u32x8 b0, b1, b2, b3;
            sload_desc4(t + (size_t)j * 8, b0, b1, b2, b3);
            uint32_t d0, d1, d2, d3;
            dist4(a, b0, b1, b2, b3, d0, d1, d2, d3);
            group(d0, d1, d2, d3, j);
        }
        for (; j < je; ++j) {
            const uint32_t d = hamming256(a, t + (size_t)j * 8);
            if (d <= near_thr) hit(d, j);
        }
    }
#pragma unroll
    for (int k = 0; k < kTopK; ++k) s_top[wv][k][lane] = top[k];
    if (q < n2) near_cnt[((size_t)p * max_n2 + q) * kNearSplit + wv] = cnt;
    __syncthreads();
    if (wv == 0 && q < n2) {
#pragma unroll
        for (int w = 1; w < kNearSplit; ++w)
#pragma unroll
            for (int k = 0; k < kTopK; ++k) topk_insert(s_top[w][k][lane], top);
        uint4* dst = reinterpret_cast<uint4*>(near_top + ((size_t)p * max_n2 + q) * kTopK);
        dst[0] = make_uint4(top[0], top[1], top[2], top[3]);
        dst[1] = make_uint4(top[4], top[5], top[6], top[7]);
    }
}

// ---- 2./3. resolve ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ratio_rejects(float lowe_ratio, uint32_t second, uint32_t best) {
    return __fmul_rn(lowe_ratio, (float)second) < (float)best;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

// One wave per problem replays upstream's idx_2 loop, 64 queries (one per lane) at a time:
//   * every unresolved lane evaluates its query against the current `already_matched` set: first two unclaimed keys of its
//     sorted top-8 (rarely: a rescan of its full near list) -> best, second, accept?, target;
//   * accepting lanes stamp mark[target] = lane (LDS atomicMin, epoch-tagged so the table is never cleared);
//   * a lane is AFFECTED if a LOWER lane of this round wants one of the two frame keypoints its decision rests on (anything
//     before `best` is already claimed, anything after `second` cannot matter; a best > THR_LOW is a final reject);
//   * all lanes below the first affected lane commit at once -- by induction their inputs were exact -- the rest go round again.
// (Round 3 gave the windowed matchers' resolver, match_window.hip k_list_resolve, a wider commit rule -- every thread no lower thread
// can interfere with commits, not just the prefix. Here it was measured and reverted: it needs every unclaimed register candidate stamped,
// and these near lists overlap so much that more queries end up waiting: 0.129 -> 0.213 ms per 256 problems.)
// A chunk whose near list overflowed kNearSeg falls back to a literal one-query-at-a-time replay with cooperative scans.
// NW = waves per problem (1 or 4): a round replays 64 * NW queries at once; with NW = 4 the four waves meet at workgroup barriers
// (three per round) and the number of rounds per problem drops ~2.5x -- the kernel is pure dependent latency on otherwise idle CUs.
template <bool STAGED, int NW>
__global__ __launch_bounds__(64 * NW) void k_bf_resolve(const uint8_t* __restrict__ desc_1, size_t stride_1,
                                                       const int32_t* __restrict__ n1_arr, const uint8_t* __restrict__ valid_1,
                                                       const uint8_t* __restrict__ desc_2, size_t stride_2,
                                                       const int32_t* __restrict__ n2_arr, int max_n1, int max_n2, float lowe_ratio, const uint32_t* __restrict__ near_cnt,
                                                       const uint32_t* __restrict__ near_list, const uint32_t* __restrict__ near_top,
                                                       int32_t* __restrict__ pairs, int32_t* __restrict__ counts, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // explicit LDS address space: a volatile GENERIC pointer would be lowered to flat, system-scope accesses
    typedef __attribute__((address_space(3))) volatile uint32_t lds_u32;
    typedef __attribute__((address_space(3))) volatile uint8_t lds_u8;
    __shared__ uint32_t s_wave[NW];   // per-wave first affected thread / per-wave match count
    constexpr int T = 64 * NW;
    lds_u32* mark = (lds_u32*)(smem);                                  // [max_n1]
    lds_u8* claimed = (lds_u8*)(smem + (size_t)max_n1 * 4);            // [max_n1]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p = blockIdx.x;
    const int n1 = n1_arr[p], n2 = n2_arr[p];
    const uint32_t* cnts = near_cnt + (size_t)p * max_n2 * kNearSplit;
    const uint32_t* lists = near_list + (size_t)p * max_n2 * kNearSplit * kNearSeg;
    const uint32_t* tops = near_top + (size_t)p * max_n2 * kTopK;
    int32_t* out = pairs + (size_t)p * cap * 2;
    auto wg_barrier = [&]() __attribute__((always_inline)) {
        if (NW == 1) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
    };
    auto wg_any = [&](bool v) __attribute__((always_inline)) -> bool {
        if (NW == 1) return __ballot(v) != 0ull;
        return __syncthreads_or(v ? 1 : 0) != 0;
    };
    // valid_1 (optional): frame keypoints the caller excludes from matching start out "already matched" -- exactly upstream's
    // `continue` for them in the inner loop; the near lists may still name them, every consumer below skips claimed entries
    const uint8_t* v1 = valid_1 ? valid_1 + (size_t)p * (stride_1 / 32) : nullptr;
    for (int i = tid; i < n1; i += T) {
        claimed[i] = v1 ? (v1[i] ? 0 : 1) : 0;
        mark[i] = ~0u;
    }
    uint32_t n_out = 0;
    uint32_t epoch = 0;

    // STAGED: every query's segment counts and sorted top-8 are copied into LDS up front (independent 16-byte loads in flight), so
    // the replay loop below never waits on HBM: nothing else hides that latency, and hipcc cannot keep a software prefetch in flight
    // across the loop's back edge (it drains vmcnt to 0 at the first use).
    lds_u32* s_tops = (lds_u32*)(smem + (((size_t)max_n1 * 5 + 15) & ~(size_t)15));   // [n2][8]
    lds_u32* s_cnts = s_tops + (size_t)max_n2 * kTopK;                                // [n2] four saturated u8 counts
    if (STAGED) {
        for (int q = tid; q < n2; q += 4 * T) {
            uint4 cc[4], lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qq = q + T * u;
                if (qq < n2) {
                    cc[u] = *reinterpret_cast<const uint4*>(cnts + (size_t)qq * kNearSplit);
                    const uint4* src = reinterpret_cast<const uint4*>(tops + (size_t)qq * kTopK);
                    lo[u] = src[0];
                    hi[u] = src[1];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qq = q + T * u;
                if (qq < n2) {
                    s_cnts[qq] = min(cc[u].x, 255u) | (min(cc[u].y, 255u) << 8) | (min(cc[u].z, 255u) << 16) | (min(cc[u].w, 255u) << 24);
                    lds_u32* d = s_tops + (size_t)qq * kTopK;
                    d[0] = lo[u].x; d[1] = lo[u].y; d[2] = lo[u].z; d[3] = lo[u].w;
                    d[4] = hi[u].x; d[5] = hi[u].y; d[6] = hi[u].z; d[7] = hi[u].w;
                }
            }
        }
    }
    wg_barrier();
    for (int q0 = 0; q0 < n2; q0 += T) {
        const int q = q0 + tid;
        uint4 cc = make_uint4(0u, 0u, 0u, 0u), lo = make_uint4(~0u, ~0u, ~0u, ~0u), hi = lo;
        if (q < n2) {
            if (STAGED) {
                const uint32_t pk = s_cnts[q];
                cc = make_uint4(pk & 255u, (pk >> 8) & 255u, (pk >> 16) & 255u, pk >> 24);
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const __attribute__((address_space(3))) u32x4* src = (const __attribute__((address_space(3))) u32x4*)(s_tops + (size_t)q * kTopK);
                const u32x4 l4 = src[0], h4 = src[1];
                lo = make_uint4(l4.x, l4.y, l4.z, l4.w);
                hi = make_uint4(h4.x, h4.y, h4.z, h4.w);
            } else {
                cc = *reinterpret_cast<const uint4*>(cnts + (size_t)q * kNearSplit);
                const uint4* src = reinterpret_cast<const uint4*>(tops + (size_t)q * kTopK);
                lo = src[0];
                hi = src[1];
            }
        }
        const uint32_t c4x = cc.x, c4y = cc.y, c4z = cc.z, c4w = cc.w;   // per-wave segment counts (named: a runtime-indexed array would live in scratch)
        const uint32_t c = c4x + c4y + c4z + c4w;
        const bool over = c4x > (uint32_t)kNearSeg || c4y > (uint32_t)kNearSeg || c4z > (uint32_t)kNearSeg || c4w > (uint32_t)kNearSeg;
        uint32_t top[kTopK];
#pragma unroll
        for (int k = 0; k < kTopK; ++k) top[k] = ~0u;
        if (c) {
            top[0] = lo.x; top[1] = lo.y; top[2] = lo.z; top[3] = lo.w;
            top[4] = hi.x; top[5] = hi.y; top[6] = hi.z; top[7] = hi.w;
        }
        int32_t my_match = -1;
        bool more = c > (uint32_t)kTopK;   // the near list holds entries beyond the register top-8

        // evaluation of this lane's query against the current claims (lists are complete: !over)
        auto evaluate = [&]() __attribute__((always_inline)) -> uint2 {
            uint32_t best = ~0u, second = ~0u;
            // the eight claim flags are fetched as one batch of independent LDS reads (through the volatile pointer every read
            // was followed by its own wait: ~8 serialized LDS round trips per evaluation); the compiler barrier at the top of each
            // round keeps them from being cached across rounds
            const __attribute__((address_space(3))) uint8_t* cl = (const __attribute__((address_space(3))) uint8_t*)claimed;
            uint32_t dead[kTopK];
#pragma unroll
            for (int k = 0; k < kTopK; ++k) dead[k] = top[k] != ~0u ? (uint32_t)cl[top[k] & 0xFFFFu] : 1u;
#pragma unroll
            for (int k = 0; k < kTopK; ++k) {
                const uint32_t e = top[k];
                if (!dead[k]) {   // keys ascend: the first two unclaimed are best and second
                    if (best == ~0u) best = e;
                    else if (second == ~0u) second = e;
                }
            }
            if (second == ~0u && more) {
                // the top-8 has run dry but the near list is longer (dense clusters of near-duplicates): rescan the whole list, eight
                // independent loads at a time (a one-entry-per-trip loop pays one HBM round trip per entry on an otherwise idle CU),
                // and REFILL the register top-8 with the eight smallest keys that are still unclaimed, so the query does not rescan
                // again until those are gone too.
                uint32_t nt[kTopK];
#pragma unroll
                for (int k = 0; k < kTopK; ++k) nt[k] = ~0u;
                const uint32_t* seg = lists + (size_t)q * kNearSplit * kNearSeg;
                uint32_t alive_total = 0;
#pragma unroll 1   // compact code: this path runs on a lone wave and its footprint matters more than its trip count
                for (int w = 0; w < kNearSplit; ++w) {
                    const uint32_t nw = w == 0 ? c4x : w == 1 ? c4y : w == 2 ? c4z : c4w;
#pragma unroll 1
                    for (uint32_t k0 = 0; k0 < nw; k0 += 8) {
                        uint32_t e8[8], d8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) e8[u] = k0 + u < nw ? seg[w * kNearSeg + k0 + u] : ~0u;
#pragma unroll
                        for (int u = 0; u < 8; ++u) d8[u] = e8[u] != ~0u ? (uint32_t)cl[e8[u] & 0xFFFFu] : 1u;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (!d8[u]) {
                                ++alive_total;
                                topk_insert(e8[u], nt);
                            }
                    }
                }
#pragma unroll
                for (int k = 0; k < kTopK; ++k) top[k] = nt[k];
                more = alive_total > (uint32_t)kTopK;   // everything alive is now in registers unless there were more than eight
                best = nt[0];
                second = nt[1];
            }
            return make_uint2(best, second);
        };
        auto accepts = [&](uint32_t best, uint32_t second) __attribute__((always_inline)) -> bool {
            if (best == ~0u) return false;
            const uint32_t bd = best >> 16;
            const uint32_t sd = second == ~0u ? (uint32_t)OVS_MAX_HAMMING_DIST : (second >> 16);
            return bd <= (uint32_t)OVS_HAMMING_DIST_THR_LOW && !ratio_rejects(lowe_ratio, sd, bd);
        };

        if (!wg_any(over)) {
            bool mine = c > 0;
            while (wg_any(mine)) {   // (NW > 1: this barrier also publishes the previous round's commits)
                uint32_t best = ~0u, second = ~0u;
                bool acc = false;
                asm volatile("" ::: "memory");   // claims committed in the previous round must be re-read
                ++epoch;
                const uint32_t tag = (0xFFFFFFu - epoch) << 8;   // newer rounds carry smaller tags: atomicMin overrides stale marks
                if (mine) {
                    const uint2 r = evaluate();
                    best = r.x;
                    second = r.y;
                    acc = accepts(best, second);
                    if (acc) __hip_atomic_fetch_min((__attribute__((address_space(3))) uint32_t*)&mark[best & 0xFFFFu], tag | (uint32_t)tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                wg_barrier();
                bool affected = false;
                if (mine && best != ~0u && (best >> 16) <= (uint32_t)OVS_HAMMING_DIST_THR_LOW) {
                    const uint32_t m1 = mark[best & 0xFFFFu];
                    affected = (m1 & ~0xFFu) == tag && (m1 & 0xFFu) < (uint32_t)tid;
                    // losing `second` to a lower lane can only RAISE it, which never turns an accept (ratio * second >= best) into a
                    // reject and never changes its target: only a lane the ratio test currently rejects has to look again
                    if (second != ~0u && !acc) {
                        const uint32_t m2 = mark[second & 0xFFFFu];
                        affected |= (m2 & ~0xFFu) == tag && (m2 & 0xFFu) < (uint32_t)tid;
                    }
                }
                const unsigned long long aff = __ballot(affected && mine);
                int f = aff ? (wv * 64 + __ffsll((long long)aff) - 1) : T;   // first affected thread of this wave
                if (NW > 1) {
                    if (lane == 0) s_wave[wv] = (uint32_t)f;
                    __syncthreads();
#pragma unroll
                    for (int w = 0; w < NW; ++w) f = min(f, (int)s_wave[w]);
                }
                if (mine && tid < f) {
                    if (acc) {
                        my_match = (int32_t)(best & 0xFFFFu);
                        claimed[best & 0xFFFFu] = 1;
                    }
                    mine = false;
                }
                if (NW == 1) __builtin_amdgcn_wave_barrier();
            }
        } else {
            // ---- pathological chunk: literal replay, one query at a time, wave after wave; overflowed queries scan every frame descriptor
            for (int w = 0; w < NW; ++w) {
                if (wv == w) {
                    const int qw0 = q0 + 64 * w;
                    unsigned long long todo = __ballot(c > 0);
                    while (todo) {
                        const int i = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        uint32_t best = ~0u, second = ~0u;
                        const bool over_i = __shfl((int)over, i) != 0;
                        if (!over_i) {
                            if (lane == i) {
                                const uint2 r = evaluate();
                                best = r.x;
                                second = r.y;
                            }
                        } else {
                            const int qi = qw0 + i;
                            uint32_t a[8];
                            const uint32_t* src = reinterpret_cast<const uint32_t*>(desc_2 + (size_t)p * stride_2 + (size_t)qi * 32);
#pragma unroll
                            for (int k = 0; k < 8; ++k) a[k] = src[k];
                            uint32_t k1 = ~0u, k2 = ~0u;
                            for (int j = lane; j < n1; j += 64) {
                                if (claimed[j]) continue;
                                const uint32_t* b = reinterpret_cast<const uint32_t*>(desc_1 + (size_t)p * stride_1 + (size_t)j * 32);
                                uint32_t d = 0;
#pragma unroll
                                for (int k = 0; k < 8; ++k) d += __builtin_popcount(a[k] ^ b[k]);
                                if (d >= (uint32_t)OVS_MAX_HAMMING_DIST) continue;   // strict '<' against MAX_HAMMING_DIST upstream
                                const uint32_t e = (d << 16) | (uint32_t)j;
                                if (e < k1) { k2 = k1; k1 = e; }
                                else if (e < k2) k2 = e;
                            }
                            const uint32_t g1 = wave_min_u32(k1);
                            const uint32_t g2 = wave_min_u32(k1 == g1 ? k2 : k1);
                            if (lane == i) { best = g1; second = g2; }
                        }
                        if (lane == i && accepts(best, second)) {
                            my_match = (int32_t)(best & 0xFFFFu);
                            claimed[best & 0xFFFFu] = 1;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                wg_barrier();
            }
        }
        // ---- emit this chunk's pairs in ascending idx_2
        const unsigned long long m = __ballot(my_match >= 0);
        uint32_t base = n_out, total = (uint32_t)__popcll(m);
        if (NW > 1) {
            if (lane == 0) s_wave[wv] = total;
            __syncthreads();
            total = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if (w < wv) base += s_wave[w];
                total += s_wave[w];
            }
            __syncthreads();   // s_wave is reused by the next chunk's rounds
        }
        if (my_match >= 0) {
            const uint32_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < (uint32_t)cap) {
                out[2 * pos] = my_match;
                out[2 * pos + 1] = q;
            }
        }
        n_out += total;
    }
    if (tid == 0) counts[p] = (int32_t)(n_out < (uint32_t)cap ? n_out : (uint32_t)cap);
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
    int near_path = OVS_NEAR_PATH_MATRIX;
    hipStream_t stream = nullptr;
    uint32_t* d_near_cnt = nullptr;
    uint32_t* d_near_list = nullptr;
    uint32_t* d_near_top = nullptr;
    // host-API staging (one problem)
    uint8_t* d_desc_1 = nullptr;
    uint8_t* d_desc_2 = nullptr;
    uint8_t* d_valid = nullptr;
    uint8_t* d_valid_1 = nullptr;
    int32_t* d_n = nullptr;        // [2]
    int32_t* d_pairs = nullptr;
    int32_t* d_count = nullptr;
    int32_t* d_best_idx = nullptr;
    uint16_t* d_best = nullptr;
    uint16_t* d_second = nullptr;
    size_t resolve_lds = 0;        // mark + claimed
    size_t resolve_lds_staged = 0; // + per-query top-8 and counts (0 = does not fit: the resolver reads them from HBM)
    StageProfiler<2> prof;
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
    // a distance of 256 can never become best (strict '<' against the initial MAX_HAMMING_DIST) and as `second` equals "none"
    return std::min<uint32_t>(thr, OVS_MAX_HAMMING_DIST - 1);
}

ovs_status run_bf(ovs_matcher* m, const uint8_t* d1, size_t stride_1, const int32_t* d_n1, const uint8_t* d_valid_1, const uint8_t* d2,
                  size_t stride_2, const int32_t* d_n2, const uint8_t* d_valid, int batch, float lowe_ratio, int32_t* d_pairs, int32_t* d_counts, int cap,
                  hipStream_t s) {
    const uint32_t thr = near_threshold(lowe_ratio);
    OVS_HIP_TRY(m->prof.begin(s));
    if (m->near_path == OVS_NEAR_PATH_POPCOUNT) {
        const int chunks = (m->max_n2 + 63) / 64, total_wg = chunks * batch;
        hipLaunchKernelGGL(k_hamming_near_popc, dim3(((total_wg + 7) / 8) * 8), dim3(256), 0, s, d1, stride_1, d_n1, d2, stride_2, d_n2, d_valid,
                           m->max_n2, thr, m->d_near_cnt, m->d_near_list, m->d_near_top, chunks, total_wg);
    } else {
        const int chunks = (m->max_n2 + kNearQueries - 1) / kNearQueries, total_wg = chunks * batch;
        hipLaunchKernelGGL(k_hamming_near, dim3(((total_wg + 7) / 8) * 8), dim3(256), 0, s, d1, stride_1, d_n1, d2, stride_2, d_n2, d_valid,
                           m->max_n2, thr, m->d_near_cnt, m->d_near_list, m->d_near_top, chunks, total_wg);
    }
    OVS_HIP_TRY(hipGetLastError());
    OVS_HIP_TRY(m->prof.mark(1, s));
    if (m->resolve_lds_staged)
        hipLaunchKernelGGL((k_bf_resolve<true, 4>), dim3(batch), dim3(256), m->resolve_lds_staged, s, d1, stride_1, d_n1, d_valid_1, d2, stride_2,
                           d_n2, m->max_n1, m->max_n2, lowe_ratio, m->d_near_cnt, m->d_near_list, m->d_near_top, d_pairs, d_counts, cap);
    else
        hipLaunchKernelGGL((k_bf_resolve<false, 1>), dim3(batch), dim3(64), m->resolve_lds, s, d1, stride_1, d_n1, d_valid_1, d2, stride_2, d_n2,
                           m->max_n1, m->max_n2, lowe_ratio, m->d_near_cnt, m->d_near_list, m->d_near_top, d_pairs, d_counts, cap);
    OVS_HIP_TRY(hipGetLastError());
    OVS_HIP_TRY(m->prof.mark(2, s));
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
    m->resolve_lds = (size_t)max_n1 * 4 + (((size_t)max_n1 + 15) & ~(size_t)15);
    if (m->resolve_lds > 150 * 1024) {
        delete m;
        return OVS_ERR_CAPACITY;
    }
    {
        const size_t staged = (((size_t)max_n1 * 5 + 15) & ~(size_t)15) + (size_t)max_n2 * (kTopK + 1) * 4;
        m->resolve_lds_staged = staged <= 150 * 1024 ? staged : 0;
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
    CREATE_TRY(hipMalloc(&m->d_near_cnt, sizeof(uint32_t) * B * max_n2 * kNearSplit));
    CREATE_TRY(hipMalloc(&m->d_near_list, sizeof(uint32_t) * B * max_n2 * kNearSplit * kNearSeg));
    CREATE_TRY(hipMalloc(&m->d_near_top, sizeof(uint32_t) * B * max_n2 * kTopK));
    CREATE_TRY(hipMalloc(&m->d_desc_1, (size_t)max_n1 * 32));
    CREATE_TRY(hipMalloc(&m->d_desc_2, (size_t)max_n2 * 32));
    CREATE_TRY(hipMalloc(&m->d_valid, (size_t)std::max(max_n1, max_n2)));
    CREATE_TRY(hipMalloc(&m->d_valid_1, (size_t)max_n1));
    CREATE_TRY(hipMalloc(&m->d_n, sizeof(int32_t) * 2));
    CREATE_TRY(hipMalloc(&m->d_pairs, sizeof(int32_t) * 2 * max_n2));
    CREATE_TRY(hipMalloc(&m->d_count, sizeof(int32_t)));
    CREATE_TRY(hipMalloc(&m->d_best_idx, sizeof(int32_t) * max_n2));
    CREATE_TRY(hipMalloc(&m->d_best, sizeof(uint16_t) * max_n2));
    CREATE_TRY(hipMalloc(&m->d_second, sizeof(uint16_t) * max_n2));
    if (m->resolve_lds > 64 * 1024)
        CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bf_resolve<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)m->resolve_lds));
    if (m->resolve_lds_staged > 64 * 1024)
        CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bf_resolve<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)m->resolve_lds_staged));
#undef CREATE_TRY
    *out = m;
    return OVS_OK;
}

ovs_status ovs_matcher_destroy(ovs_matcher* m) {
    if (!m) return OVS_OK;
    if (m->stream) hipStreamSynchronize(m->stream);
    hipFree(m->d_near_cnt);
    hipFree(m->d_near_list);
    hipFree(m->d_near_top);
    hipFree(m->d_desc_1);
    hipFree(m->d_desc_2);
    hipFree(m->d_valid);
    hipFree(m->d_valid_1);
    hipFree(m->d_n);
    hipFree(m->d_pairs);
    hipFree(m->d_count);
    hipFree(m->d_best_idx);
    hipFree(m->d_best);
    hipFree(m->d_second);
    m->prof.destroy();
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
    return OVS_OK;
}

ovs_status ovs_matcher_set_near_path(ovs_matcher* m, int32_t path) {
    if (!m || (path != OVS_NEAR_PATH_MATRIX && path != OVS_NEAR_PATH_POPCOUNT)) return OVS_ERR_INVALID;
    m->near_path = path;
    return OVS_OK;
}

ovs_status ovs_matcher_profile_enable(ovs_matcher* m, int32_t enable) {
    if (!m) return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(m->device));
    m->prof.enabled = enable != 0;
    if (enable) OVS_HIP_TRY(m->prof.ensure());
    return OVS_OK;
}

ovs_status ovs_matcher_profile_read(ovs_matcher* m, float* stage_ms, int32_t* ncalls) {
    if (!m || !stage_ms || !ncalls) return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(m->device));
    OVS_HIP_TRY(m->prof.read(stage_ms, ncalls));
    return OVS_OK;
}

ovs_status ovs_robust_brute_force_match_batch_dev(ovs_matcher* m, const uint8_t* d_desc_1, size_t stride_1, const int32_t* d_n1,
                                                  const uint8_t* d_valid_1, const uint8_t* d_desc_2, size_t stride_2, const int32_t* d_n2,
                                                  const uint8_t* d_valid_2, int32_t batch, float lowe_ratio, int32_t* d_pairs,
                                                  int32_t* d_counts, int32_t cap, void* stream) {
    if (!m || !d_desc_1 || !d_desc_2 || !d_n1 || !d_n2 || !d_pairs || !d_counts || batch < 1 || cap < 1) return OVS_ERR_INVALID;
    if (batch > m->max_batch || stride_1 > (size_t)m->max_n1 * 32 || stride_2 > (size_t)m->max_n2 * 32) return OVS_ERR_CAPACITY;
    if (((uintptr_t)d_desc_1 & 3) || ((uintptr_t)d_desc_2 & 3) || (stride_1 & 31) || (stride_2 & 31)) return OVS_ERR_ALIGN;
    OVS_HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;   // verbatim: NULL is HIP's default stream (torch's default stream handle is 0)
    return run_bf(m, d_desc_1, stride_1, d_n1, d_valid_1, d_desc_2, stride_2, d_n2, d_valid_2, batch, lowe_ratio, d_pairs, d_counts, cap, s);
}

ovs_status ovs_robust_brute_force_match(ovs_matcher* m, const uint8_t* desc_1, int32_t n1, const uint8_t* valid_1, const uint8_t* desc_2,
                                        int32_t n2, const uint8_t* valid_2, float lowe_ratio, int32_t* pairs, int32_t cap, int32_t* n_out) {
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
    if (valid_1) OVS_HIP_TRY(hipMemcpyAsync(m->d_valid_1, valid_1, (size_t)n1, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(m->d_n, nn, sizeof(nn), hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));   // nn is a stack array
    ovs_status st = run_bf(m, m->d_desc_1, (size_t)m->max_n1 * 32, m->d_n, valid_1 ? m->d_valid_1 : nullptr, m->d_desc_2, (size_t)m->max_n2 * 32, m->d_n + 1,
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
