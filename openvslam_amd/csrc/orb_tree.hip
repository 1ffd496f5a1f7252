// orb_tree.hip -- A4: orb_extractor::distribute_keypoints_via_tree / initialize_nodes / assign_child_nodes /
// find_keypoints_with_max_response and orb_extractor_node::divide_node (expected: src/openvslam/feature/
// orb_extractor.cc, orb_extractor_node.cc; lineage ORB-SLAM2 DistributeOctTree).
//
// Upstream mutates a std::list (push_front children, erase parent) and std::sort's (count, node*) pairs; the result --
// including the ORDER of the returned keypoints -- is reproduced here without a list (tools/tree_model.py is the
// executable statement of this algorithm and is tested against the oracle's literal std::list restatement):
//   * a pass splits a set S of nodes in a processing order: phase 1 = every non-leaf node in list order; phase 2 = nodes
//     sorted by (count desc, list position asc), cut at the first split that makes #nodes >= N;
//   * children are pushed to the front in creation order, so new list = reverse(creation sequence) ++ (old list \ S):
//     child with creation index c sits at total_new-1-c, a surviving node keeps its relative order behind them;
//   * a candidate only needs its current node id (16 bits inside its list entry); child counts are LDS atomics;
//   * "max response, first wins" = 64-bit LDS atomic max over (score, ~emission_order).
// Tie rule for equal counts (implementation-defined upstream: pointer order): later-created node first = list position
// ascending, exactly as the oracle defines it.
//
// One 512-thread workgroup per (level, frame) problem (1024 threads finish one problem sooner, 0.076 vs 0.087 ms per 32 frames, but only two
// such workgroups fit a CU; at 128 frames per launch 512 threads win, 0.247 vs 0.278 ms); node lists (< 4N+64 records) ping-pong in an L2-resident global
// scratch, everything else lives in LDS. The kernel is latency-bound by design (a handful of passes over ~10^4
// candidates); throughput comes from running levels x frames problems concurrently.
#include "ovs_common.h"

namespace ovs {

constexpr int kTreeThreads = 512;
constexpr uint32_t kNotInS = 0xFFFFFFFFu;

struct NodeRec {
    uint32_t xb;      // bx | ex << 16
    uint32_t yb;      // by | ey << 16
    uint32_t count;
    uint32_t pad;
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// Exclusive scan of one value per thread over the whole block (two barriers). s_wave: 16 words of LDS.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_wave, uint32_t& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kTreeThreads / 64; ++w) {
        const uint32_t t = s_wave[w];
        if (w < wv) pre += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return pre + incl - v;
}

// out[i] = sum_{j<i} in[j] for i < n (in/out in LDS, may alias); returns the total. Contains barriers.
__device__ __forceinline__ uint32_t array_excl_scan(const uint32_t* in, uint32_t* out, int n, uint32_t* s_wave) {
    const int ipt = (n + kTreeThreads - 1) / kTreeThreads;
    const int b = threadIdx.x * ipt, e = min(n, b + ipt);
    uint32_t sum = 0;
    for (int i = b; i < e; ++i) sum += in[i];
    uint32_t total;
    uint32_t run = block_excl_scan(sum, s_wave, total);
    for (int i = b; i < e; ++i) {
        const uint32_t v = in[i];
        out[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(kTreeThreads) void k_tree(const FrameGeo* __restrict__ geo, uint64_t* __restrict__ cand,
                                                      size_t cand_frame_entries, const uint32_t* __restrict__ cand_count,
                                                      NodeRec* __restrict__ nodes, size_t node_frame_entries,
                                                      uint64_t* __restrict__ lvl_kps, uint32_t* __restrict__ lvl_count, int NCmax,
                                                      int P2max, int level_lo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // ---- LDS carve (all offsets multiples of 16)
    const int NN = 4 * NCmax;
    uint32_t* child_cnt = reinterpret_cast<uint32_t*>(smem);             // [NN] count of child k of current node j at 4j+k
    uint32_t* cmap = child_cnt + NN;                                      // [NN] new list position of that child
    unsigned long long* best = reinterpret_cast<unsigned long long*>(smem);   // [NN] aliases child_cnt+cmap (final step)
    uint32_t* nch = cmap + NN;                                            // [NCmax] non-empty children (0 for leaves)
    uint32_t* keep_pos = nch + NCmax;                                     // [NCmax] new position if the node survives
    uint32_t* base = keep_pos + NCmax;                                    // [NCmax] creation index of the node's first child
    uint32_t* rank = base + NCmax;                                        // [NCmax] processing rank, kNotInS if not split
    uint32_t* tmp = rank + NCmax;                                         // [NCmax] scan scratch
    uint32_t* sidx = tmp + NCmax;                                         // [P2max] node at processing rank r
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(sidx + P2max);   // [P2max]
    uint32_t* splitxy = reinterpret_cast<uint32_t*>(keys + P2max);        // [NCmax] cx | cy << 16 (valid where nch/rank say)
    uint32_t* s_wave = splitxy + NCmax;                                   // [16]
    uint32_t* s_misc = s_wave + 16;                                       // [80]: [0..63] root counts / root pos, [64..] scalars

    const int tid = threadIdx.x;
    const int level = level_lo + (int)blockIdx.x, frame = blockIdx.y;   // the launch covers levels [level_lo, level_lo + gridDim.x)
    const int L = geo->num_levels;
    const LevelGeo& g = geo->lv[level];
    const uint32_t N = (uint32_t)g.n_keypts;
    uint32_t n = cand_count[frame * L + level];
    if (n > (uint32_t)g.cand_cap) n = g.cand_cap;
    uint64_t* list = cand + (size_t)frame * cand_frame_entries + g.cand_off;
    NodeRec* cur = nodes + (size_t)frame * node_frame_entries + g.node_off;
    NodeRec* nxt = cur + g.max_nodes;
    uint64_t* out = lvl_kps + (size_t)frame * geo->total_kp_cap + g.kp_base;
    if (n == 0) {
        if (tid == 0) lvl_count[frame * L + level] = 0;
        return;
    }

    // ---- initialize_nodes: root patches, candidates to roots (double division, as upstream's keypt.pt.x / delta_x)
    const int gx = g.gx, gy = g.gy, nroot = gx * gy;
    const double dx = g.dx, dy = g.dy;
    if (tid < 64) s_misc[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kTreeThreads) {
        const uint64_t c = list[i];
        const float fx = (float)((int)cand_x(c) - kOrbPatchRadius), fy = (float)((int)cand_y(c) - kOrbPatchRadius);
        uint32_t ix = (uint32_t)((double)fx / dx), iy = (uint32_t)((double)fy / dy);
        ix = min(ix, (uint32_t)gx - 1u);
        iy = min(iy, (uint32_t)gy - 1u);
        const uint32_t r = ix + iy * gx;
        atomicAdd(&s_misc[r], 1u);
        list[i] = (c & ~0xFFFFull) | r;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t pos = 0;
        for (int r = 0; r < nroot; ++r) {
            const uint32_t cnt = s_misc[r];
            if (cnt) {
                const int ix = r % gx, iy = r / gx;
                NodeRec rec;
                rec.xb = (uint32_t)(int)(dx * ix) | ((uint32_t)(int)(dx * (ix + 1)) << 16);
                rec.yb = (uint32_t)(int)(dy * iy) | ((uint32_t)(int)(dy * (iy + 1)) << 16);
                rec.count = cnt;
                rec.pad = 0;
                cur[pos] = rec;
                s_misc[r] = pos++;
            } else
                s_misc[r] = 0xFFFFu;
        }
        s_misc[64] = pos;   // list size
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kTreeThreads) {
        const uint64_t c = list[i];
        list[i] = (c & ~0xFFFFull) | s_misc[(uint32_t)c & 0xFFFFu];
    }
    uint32_t size = s_misc[64];
    __syncthreads();

    int phase = 1;
    for (int guard = 0; guard < 64; ++guard) {   // upstream's while(true); depth is bounded by log2(image size) << 64
        const uint32_t prev = size;
        // ---- 1. split coordinates of every current node; clear child counters
        for (uint32_t j = tid; j < size; j += kTreeThreads) {
            const NodeRec r = cur[j];
            const uint32_t bx = r.xb & 0xFFFFu, ex = r.xb >> 16, by = r.yb & 0xFFFFu, ey = r.yb >> 16;
            // divide_node: half = ceil((end - begin) / 2.0)
            const uint32_t cx = bx + ((ex - bx + 1u) >> 1), cy = by + ((ey - by + 1u) >> 1);
            splitxy[j] = cx | (cy << 16);
            tmp[j] = r.count > 1 ? 1u : 0u;   // non-leaf flag
            child_cnt[4 * j + 0] = 0;
            child_cnt[4 * j + 1] = 0;
            child_cnt[4 * j + 2] = 0;
            child_cnt[4 * j + 3] = 0;
        }
        __syncthreads();
        // ---- 2. count candidates per child of every non-leaf node
        for (uint32_t i = tid; i < n; i += kTreeThreads) {
            const uint64_t c = list[i];
            const uint32_t nd = (uint32_t)c & 0xFFFFu;
            if (tmp[nd]) {
                const uint32_t s = splitxy[nd];
                const uint32_t x = cand_x(c) - kOrbPatchRadius, y = cand_y(c) - kOrbPatchRadius;
                const uint32_t k = (x >= (s & 0xFFFFu) ? 1u : 0u) + (y >= (s >> 16) ? 2u : 0u);
                atomicAdd(&child_cnt[4 * nd + k], 1u);
            }
        }
        __syncthreads();
        for (uint32_t j = tid; j < size; j += kTreeThreads)
            nch[j] = (child_cnt[4 * j] != 0) + (child_cnt[4 * j + 1] != 0) + (child_cnt[4 * j + 2] != 0) + (child_cnt[4 * j + 3] != 0);
        __syncthreads();

        // ---- 3. processing order: sidx[r] = node processed r-th, M nodes are split
        uint32_t M;
        if (phase == 1) {
            // all non-leaf nodes in list order
            M = array_excl_scan(tmp, rank, (int)size, s_wave);   // rank[j] = #non-leaf before j
            for (uint32_t j = tid; j < size; j += kTreeThreads) {
                if (tmp[j]) sidx[rank[j]] = j;
                else rank[j] = kNotInS;
            }
            __syncthreads();
        } else {
            // sort the pool by (count desc, list position asc); leaves sort to the end
            uint32_t P2 = 1;
            while (P2 < size) P2 <<= 1;
            for (uint32_t j = tid; j < P2; j += kTreeThreads) {
                unsigned long long key = ~0ull;
                if (j < size && tmp[j]) key = ((unsigned long long)(0xFFFFFFu - cur[j].count) << 16) | j;
                keys[j] = key;
            }
            __syncthreads();
            for (uint32_t k2 = 2; k2 <= P2; k2 <<= 1) {
                for (uint32_t jj = k2 >> 1; jj > 0; jj >>= 1) {
                    for (uint32_t i = tid; i < P2; i += kTreeThreads) {
                        const uint32_t ixj = i ^ jj;
                        if (ixj > i) {
                            const unsigned long long a = keys[i], b = keys[ixj];
                            const bool up = (i & k2) == 0;
                            if (up ? (a > b) : (a < b)) {
                                keys[i] = b;
                                keys[ixj] = a;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            // pool size, gains in processing order, first rank at which #nodes reaches N
            uint32_t npool_cur = array_excl_scan(tmp, rank, (int)size, s_wave);   // rank[] reused below
            for (uint32_t r = tid; r < npool_cur; r += kTreeThreads) {
                const uint32_t j = (uint32_t)keys[r] & 0xFFFFu;
                sidx[r] = j;
                base[r] = nch[j] - 1u;   // gain of splitting the r-th pool node (nch >= 1 for a non-leaf)
            }
            if (tid == 0) s_misc[65] = npool_cur;   // M candidate (atomicMin below)
            __syncthreads();
            array_excl_scan(base, base, (int)npool_cur, s_wave);   // base[r] = gain of ranks < r
            for (uint32_t r = tid; r < npool_cur; r += kTreeThreads) {
                const uint32_t after = size + base[r] + (nch[sidx[r]] - 1u);
                if (after >= N) atomicMin(&s_misc[65], r + 1u);
            }
            __syncthreads();
            M = s_misc[65];
            for (uint32_t j = tid; j < size; j += kTreeThreads) rank[j] = kNotInS;
            __syncthreads();
            for (uint32_t r = tid; r < M; r += kTreeThreads) rank[sidx[r]] = r;
            __syncthreads();
        }

        // ---- 4. creation index of each split node's first child (processing order), positions of survivors (list order)
        for (uint32_t r = tid; r < M; r += kTreeThreads) tmp[r] = nch[sidx[r]];
        __syncthreads();
        const uint32_t total_new = array_excl_scan(tmp, tmp, (int)M, s_wave);
        for (uint32_t r = tid; r < M; r += kTreeThreads) base[sidx[r]] = tmp[r];
        __syncthreads();
        for (uint32_t j = tid; j < size; j += kTreeThreads) tmp[j] = (rank[j] == kNotInS) ? 1u : 0u;
        __syncthreads();
        array_excl_scan(tmp, keep_pos, (int)size, s_wave);
        const uint32_t new_size = total_new + (size - M);

        // ---- 5. write the new list; count splittable children (the next pool)
        uint32_t my_pool = 0;
        for (uint32_t j = tid; j < size; j += kTreeThreads) {
            const NodeRec r = cur[j];
            if (rank[j] == kNotInS) {
                const uint32_t p = total_new + keep_pos[j];
                keep_pos[j] = p;
                nxt[p] = r;
            } else {
                const uint32_t bx = r.xb & 0xFFFFu, ex = r.xb >> 16, by = r.yb & 0xFFFFu, ey = r.yb >> 16;
                const uint32_t s = splitxy[j];
                const uint32_t cx = s & 0xFFFFu, cy = s >> 16;
                uint32_t c = base[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t cnt = child_cnt[4 * j + k];
                    if (cnt) {
                        const uint32_t p = total_new - 1u - c;
                        ++c;
                        NodeRec ch;
                        ch.xb = (k & 1) ? (cx | (ex << 16)) : (bx | (cx << 16));
                        ch.yb = (k & 2) ? (cy | (ey << 16)) : (by | (cy << 16));
                        ch.count = cnt;
                        ch.pad = 0;
                        nxt[p] = ch;
                        cmap[4 * j + k] = p;
                        my_pool += cnt > 1 ? 1u : 0u;
                    }
                }
            }
        }
        uint32_t npool_new;
        block_excl_scan(my_pool, s_wave, npool_new);   // barriers inside: nxt / cmap / keep_pos now visible

        // ---- 6. move every candidate to its new node
        for (uint32_t i = tid; i < n; i += kTreeThreads) {
            const uint64_t c = list[i];
            const uint32_t nd = (uint32_t)c & 0xFFFFu;
            uint32_t p;
            if (rank[nd] == kNotInS) p = keep_pos[nd];
            else {
                const uint32_t s = splitxy[nd];
                const uint32_t x = cand_x(c) - kOrbPatchRadius, y = cand_y(c) - kOrbPatchRadius;
                const uint32_t k = (x >= (s & 0xFFFFu) ? 1u : 0u) + (y >= (s >> 16) ? 2u : 0u);
                p = cmap[4 * nd + k];
            }
            list[i] = (c & ~0xFFFFull) | p;
        }
        __syncthreads();
        NodeRec* t = cur;
        cur = nxt;
        nxt = t;
        size = new_size;
        // ---- 7. upstream's stop rules
        if (N <= size || size == prev) break;
        if (phase == 1 && N < size + 3u * npool_new) phase = 2;
    }

    // ---- find_keypoints_with_max_response: per node max (score, first in emission order); output in list order
    for (uint32_t j = tid; j < size; j += kTreeThreads) best[j] = 0ull;
    __syncthreads();
    const uint32_t ncx = (uint32_t)g.ncx;
    for (uint32_t i = tid; i < n; i += kTreeThreads) {
        const uint64_t c = list[i];
        const uint32_t x = cand_x(c), y = cand_y(c);
        const unsigned long long key = ((unsigned long long)(cand_score(c) + 1u) << 32) | (0xFFFFFFFFu - cand_order(x, y, ncx));
        __hip_atomic_fetch_max(&best[(uint32_t)c & 0xFFFFu], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    const uint32_t kp_cap = (uint32_t)g.kp_cap;
    for (uint32_t j = tid; j < size && j < kp_cap; j += kTreeThreads) {
        const unsigned long long key = best[j];
        const uint32_t score = (uint32_t)(key >> 32) - 1u;
        const uint32_t ord = 0xFFFFFFFFu - (uint32_t)key;
        const uint32_t cell = ord >> 12, ci = cell / ncx, cj = cell - ci * ncx;
        const uint32_t x = kDetOrigin + 64u * cj + (ord & 63u), y = kDetOrigin + 64u * ci + ((ord >> 6) & 63u);
        out[j] = cand_pack(x, y, score, 0);
    }
    if (tid == 0) lvl_count[frame * L + level] = size < kp_cap ? size : kp_cap;
}

static size_t tree_lds_bytes(int NCmax, int P2max) {
    return (size_t)(8 * NCmax) * 4 + (size_t)NCmax * 5 * 4 + (size_t)P2max * 4 + (size_t)P2max * 8 + (size_t)NCmax * 4 + 16 * 4 + 80 * 4;
}

hipError_t launch_tree(const FrameGeo& hgeo, const DevBuffers& d, int batch, hipStream_t s, int level_lo, int n_levels) {
    if (n_levels < 0) n_levels = hgeo.num_levels - level_lo;
    if (n_levels <= 0 || batch <= 0) return hipSuccess;
    int NCmax = 0;
    for (int l = 0; l < hgeo.num_levels; ++l) NCmax = std::max(NCmax, hgeo.lv[l].max_nodes / 4);
    NCmax = (NCmax + 3) & ~3;
    int P2max = 1;
    while (P2max < NCmax) P2max <<= 1;
    const size_t lds = tree_lds_bytes(NCmax, P2max);
    static thread_local size_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tree), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    dim3 grid(n_levels, batch);
    hipLaunchKernelGGL(k_tree, grid, dim3(kTreeThreads), lds, s, d.geo, d.cand, d.cand_frame_entries, d.cand_count,
                       reinterpret_cast<NodeRec*>(d.nodes), d.node_frame_entries, d.lvl_kps, d.lvl_count, NCmax, P2max, level_lo);
    return hipGetLastError();
}

}   // namespace ovs
