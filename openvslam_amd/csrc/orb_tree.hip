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
// Candidate sweeps (round 2): a node's split point is a function of its bounds alone, so the sweep that MOVES a candidate into a new
// child can already COUNT it into that child's own children (the next pass's child counts), and the sweep of the last pass can feed
// "max response per node" directly. One sweep over the candidates per pass plus one at the start (root and root-child counts), instead
// of two per pass plus four: 1 + P against 2 P + 4 for P passes (P ~ 6-8).
//
// Grid form (round 6, batches): the candidates are counted ONCE into a separable grid of the finest depth, every count a pass needs is a
// look-up in the pyramid of sums above it, and one last pass finds each candidate's final node through the nodes' cell marks: two passes over
// the candidates instead of 1 + P. Same node machinery, same results; the sweep form stays as the path for single frames and for problems
// the grid cannot finish (see tree_problem).
//
// One 512-thread workgroup per (level, frame) problem (1024 threads finish one problem sooner, 0.076 vs 0.087 ms per 32 frames, but only two
// such workgroups fit a CU; at 128 frames per launch 512 threads win, 0.247 vs 0.278 ms); node lists (< 4N+64 records) ping-pong in an L2-resident global
// scratch, everything else lives in LDS. The kernel is latency-bound by design (a handful of passes over ~10^4
// candidates); throughput comes from running levels x frames problems concurrently.
#include "ovs_common.h"

#include <algorithm>
#include <cstdlib>

namespace ovs {

constexpr int kTreeThreadsBatch = 512, kTreeThreadsFew = 1024;   // threads per (level, frame) problem: launches of many / of few problems

// -DOVS_TREE_TIMING: thread 0 of the (level 0, frame 0) workgroup records wall_clock64() (10 ns ticks) at the step boundaries of every
// pass and prints the intervals at the end (one printf: the intervals themselves stay clean). Measurement aid, off in the product build.
#ifdef OVS_TREE_TIMING
#define OVS_TT_DECL long long tt_[64]; int ntt_ = 0;
#define OVS_TT_MARK() do { if (ntt_ < 62) tt_[ntt_++] = wall_clock64(); } while (0)
#define OVS_TT_PRINT(n_) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && level_lo == 0) { printf("k_tree n=%u:", (unsigned)(n_)); for (int i_ = 1; i_ < ntt_; ++i_) printf(" %lld", tt_[i_] - tt_[i_ - 1]); printf("\n"); } } while (0)
#else
#define OVS_TT_DECL
#define OVS_TT_MARK() do {} while (0)
#define OVS_TT_PRINT(n_) do {} while (0)
#endif
constexpr uint32_t kNotInS = 0xFFFFFFFFu;
// candidate loads in flight per thread in a sweep. Each one holds ~15 registers across the three look-up rounds: with four the 512-thread
// kernel needs 109 VGPRs (two workgroups per CU), with two 75 (three per CU) -- for this latency-bound kernel the third workgroup is worth
// more than the deeper software pipeline: 0.346 -> 0.311 ms per 256 frames, 63 -> 60.7 us for a single frame (one: 0.327 ms)
constexpr int kSweepLoads = 2;
constexpr int kRankDirect = 1024;   // pool ordering: direct rank counting up to this many nodes, bitonic sort beyond

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
template <int kTreeThreads>
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
template <int kTreeThreads>
__device__ __forceinline__ uint32_t array_excl_scan(const uint32_t* in, uint32_t* out, int n, uint32_t* s_wave) {
    const int ipt = (n + kTreeThreads - 1) / kTreeThreads;
    const int b = threadIdx.x * ipt, e = min(n, b + ipt);
    uint32_t sum = 0;
    for (int i = b; i < e; ++i) sum += in[i];
    uint32_t total;
    uint32_t run = block_excl_scan<kTreeThreads>(sum, s_wave, total);
    for (int i = b; i < e; ++i) {
        const uint32_t v = in[i];
        out[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

// ---- the count pyramid of the grid form (round 6) ------------------------------------------------------------------------------------
// A node's split point is a function of its bounds alone and the bounds are the root patch's, halved (ceil) once per depth and axis: the nodes of
// depth d are the cells of a SEPARABLE grid of (gx << d) x (gy << d) intervals. So the candidates are counted ONCE into the grid of depth D
// (one LDS atomic each), the coarser grids are sums of four, and every count the passes need -- a node's, its children's -- is a look-up:
// the passes touch nodes only (rounds 1-5 swept all candidates once per pass: 39 of a single 1080p frame's 56 us). At the end the final nodes
// mark their cells and one more pass over the candidates feeds "max response per node". Entries are 16-bit (two per LDS word).
// depth of the finest grid for a level that wants N keypoints from `roots` root patches: one more than the depth at which a UNIFORM tree would
// reach N nodes (clustered corners go deeper; a level whose tree would go deeper still takes the sweep form), between 4 and 7. Measured at
// 1080p / N = 434 per 256 frames: depth 4 / 5 / 6 / 7 = 0.212 / 0.216 / 0.270 / 0.489 ms (sweep form 0.307), depth 3 falls back (0.429).
__host__ __device__ inline int tree_grid_depth(int N, int roots) {
    int k = 0;
    while (k < 7 && ((long long)roots << (2 * k)) < (long long)N) ++k;
    return k + 1 < 4 ? 4 : (k + 1 > 7 ? 7 : k + 1);
}
struct GridDims {
    int D;                 // depth of the finest grid
    int r0, nroot;         // root patches (r0: rounded up to an even number)
    int gx, gy;
    // first entry of depth d's grid: the depths' blocks follow each other, each an even number of entries (depth 0: r0, depth t >= 1: nroot * 4^t)
    __device__ __forceinline__ int off(int d) const { return d == 0 ? 0 : r0 + nroot * ((((1 << (2 * d)) - 4)) / 3); }
};
__device__ __forceinline__ uint32_t grid_get(const uint32_t* w, int idx) { return (w[idx >> 1] >> (16 * (idx & 1))) & 0xFFFFu; }
// One (level, frame) problem by one workgroup. kGrid: the grid form; returns false (workgroup-uniform) when it cannot finish -- the grid does
// not fit, or a node of the finest depth would have to be split -- and has then written nothing but LDS: the caller runs the sweep form.
template <int kTreeThreads, bool kGrid>
__device__ __forceinline__ bool tree_problem(unsigned char* smem, const FrameGeo* __restrict__ geo, uint64_t* __restrict__ cand,
                                             size_t cand_frame_entries, const uint32_t* __restrict__ cand_count,
                                             NodeRec* __restrict__ nodes, size_t node_frame_entries,
                                             uint64_t* __restrict__ lvl_kps, uint32_t* __restrict__ lvl_count, int NCmax,
                                             int P2max, int Rmax, int grid_words, int grid_depth, int level, int frame, int level_lo) {
    // ---- LDS carve (all offsets multiples of 16)
    const int NN = 4 * NCmax;
    uint32_t* child_cnt = reinterpret_cast<uint32_t*>(smem);             // [NN] count of child k of current node j at 4j+k
    uint32_t* child_nxt = child_cnt + NN;                                 // [NN] the same for the list being built (ping-pong)
    uint32_t* cmap = child_nxt + NN;                                      // [NN] new list position of that child
    uint32_t* nch = cmap + NN;                                            // [NCmax] non-empty children (0 for leaves)
    uint32_t* keep_pos = nch + NCmax;                                     // [NCmax] new position if the node survives
    uint32_t* base = keep_pos + NCmax;                                    // [NCmax] creation index of the node's first child
    uint32_t* rank = base + NCmax;                                        // [NCmax] processing rank, kNotInS if not split
    uint32_t* tmp = rank + NCmax;                                         // [NCmax] scan scratch
    uint32_t* sidx = tmp + NCmax;                                         // [P2max] node at processing rank r
    // the sort keys of the sorted phase live in child_nxt: 8 * P2max < 16 * NCmax bytes, and child_nxt is cleared only after step 3
    uint32_t* nxb = sidx + P2max;                                         // [NCmax] bx | ex << 16 of the current nodes
    uint32_t* nyb = nxb + NCmax;                                          // [NCmax] by | ey << 16
    uint32_t* s_wave = nyb + NCmax;                                       // [16]
    uint32_t* s_misc = s_wave + 16;                                       // [16] scalars: [0] list size, [1] split count of the sorted phase
    uint32_t* s_root = s_misc + 16;                                       // [Rmax] list position of a root patch (0xFFFF: empty)
    uint32_t* s_rcc = s_root + Rmax;                                      // [4 Rmax] the root patches' child counts
    uint32_t* gridw = s_rcc + 4 * Rmax;                                   // [grid_words] the count pyramid (grid form), two entries per word

    OVS_TT_DECL
    OVS_TT_MARK();
    const int tid = threadIdx.x;
    const int L = geo->num_levels;
    const LevelGeo& g = geo->lv[level];
    const uint32_t switch_factor = (geo->variant & 1) ? 1u : 3u;   // ORACLE_SPEC rule 6
    const bool tie_earlier = (geo->variant & 2) != 0;                // rule 7: equal counts -> earlier-created node (higher list position) first
    const uint32_t N = (uint32_t)g.n_keypts;
    uint32_t n = cand_count[frame * L + level];
    if (n > (uint32_t)g.cand_cap) n = g.cand_cap;
    uint64_t* list = cand + (size_t)frame * cand_frame_entries + g.cand_off;
    NodeRec* cur = nodes + (size_t)frame * node_frame_entries + g.node_off;
    NodeRec* nxt = cur + g.max_nodes;
    uint64_t* out = lvl_kps + (size_t)frame * geo->total_kp_cap + g.kp_base;
    if (n == 0) {
        if (tid == 0) lvl_count[frame * L + level] = 0;
        return true;
    }

    // ---- initialize_nodes: root patches, candidates to roots (double division, as upstream's keypt.pt.x / delta_x). The same sweep
    //      counts every candidate into its root AND into the root's child it falls in (the first pass's child counts).
    const int gx = g.gx, gy = g.gy, nroot = gx * gy;
    const double dx = g.dx, dy = g.dy;
    GridDims gd;
    gd.D = 0;
    gd.gx = gx;
    gd.gy = gy;
    gd.nroot = nroot;
    gd.r0 = (nroot + 1) & ~1;
    uint32_t* rc = gridw;    // [4 (gx + gy)] per root column / row: first pixel b, length, ceil(2^32 / length), first pixel of the patch index
    uint32_t* tabw = gridw;  // [(gx + gy) (2^D + 1)] 16-bit: the cells' first pixels relative to b (entry 2^D: the length), per root column / row
    bool cache_cells = false;
    if (kGrid) {
        // the deepest grid that fits: entries of depths 0 .. D (each depth's block padded to an even count), 12-bit cell indices; the
        // root patches' tables behind it. Root grids of more than 8 patches along an axis (strips) take the sweep form.
        if (n > 65535u || gx > 8 || gy > 8) return false;
        int D = -1;
        const int Dwant = grid_depth > 0 ? grid_depth : tree_grid_depth((int)N, nroot);
        for (int t = 1; t <= Dwant; ++t) {
            int tot = 0;
            for (int d = 0; d <= t; ++d) tot += ((nroot << (2 * d)) + 1) & ~1;
            if (tot / 2 + 4 * (gx + gy) + ((gx + gy) * ((1 << t) + 1) + 1) / 2 <= grid_words && (gx << t) <= 4096 && (gy << t) <= 4096) D = t;
        }
        if (D < 2) return false;
        if (tid < 16) s_misc[tid] = tid == 3 ? 0xFFFFFFFFu : 0u;
        for (int i = tid; i < grid_words / 4; i += kTreeThreads) reinterpret_cast<uint4*>(gridw)[i] = uint4{0u, 0u, 0u, 0u};   // (grid_words is a multiple of 4)
        __syncthreads();
        // per root column / row (thread i): its pixel interval as initialize_nodes computes it, the reciprocal of its length, and the first
        // pixel upstream's (int)(x / delta) sends to patch i -- found with that very expression, so the patch of a pixel is a comparison
        uint32_t* rc0 = gridw + grid_words - 4 * (gx + gy);   // (the end of the region: its place does not depend on D)
        if (tid < gx + gy) {
            const bool ax = tid < gx;
            const int r = ax ? tid : tid - gx;
            const double dd = ax ? dx : dy;
            const uint32_t b = (uint32_t)(int)(dd * r), e = (uint32_t)(int)(dd * (r + 1));
            const uint32_t len = e > b ? e - b : 0u;
            uint32_t thr = 0;
            if (r > 0) {
                int v = max(0, (int)(dd * r) - 3);
                for (int guard = 0; guard < 16 && (uint32_t)((double)(float)v / dd) < (uint32_t)r; ++guard) ++v;
                thr = (uint32_t)v;
            }
            rc0[4 * tid] = b;
            rc0[4 * tid + 1] = len;
            rc0[4 * tid + 2] = len ? (uint32_t)(((1ull << 32) + len - 1) / len) : 0u;
            rc0[4 * tid + 3] = thr;
            atomicMin(&s_misc[3], len);
            atomicMax(&s_misc[4], len);
        }
        __syncthreads();
        // cells of at least a pixel (the estimate below is then off by at most two) and an exact reciprocal division
        const uint32_t lmin = s_misc[3], lmax = s_misc[4];
        while (D >= 2 && ((1u << D) > lmin || ((unsigned long long)lmax * lmax << D) >= (1ull << 32))) --D;
        if (D < 2) return false;
        gd.D = D;
        rc = rc0;
        tabw = gridw + (gd.off(D + 1) + 1) / 2;
        cache_cells = (gx << D) <= 256 && (gy << D) <= 256;
        // the cells' first pixels: entry k of a root column / row by following k's bits down the D halvings (two entries per thread and word)
        const int n1 = (1 << D) + 1, NT = (gx + gy) * n1;
        for (int p2 = tid; 2 * p2 < NT; p2 += kTreeThreads) {
            uint32_t word = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * p2 + h;
                if (e < NT) {
                    const int i = e / n1, k = e - i * n1;
                    uint32_t b = 0, en = rc[4 * i + 1];
                    if (k == n1 - 1) b = en;
                    else
                        for (int d = 0; d < D; ++d) {
                            const uint32_t c = b + ((en - b + 1u) >> 1);
                            if ((k >> (D - 1 - d)) & 1) b = c;
                            else en = c;
                        }
                    word |= (b & 0xFFFFu) << (16 * h);
                }
            }
            tabw[p2] = word;
        }
    }
    // a candidate's cell of the finest grid: root patch as upstream's double division decides it (through the first-pixel thresholds), then
    // the D halvings per axis -- the cell is the estimate floor(u 2^D / length) or one of the two cells before it, settled by their first pixels
    auto axis_cell = [&](uint32_t v, int i) -> uint32_t {
        const int D = gd.D, n1 = (1 << D) + 1;
        const uint32_t b = rc[4 * i], len = rc[4 * i + 1], mg = rc[4 * i + 2];
        const uint32_t u = v > b ? min(v - b, len - 1u) : 0u;
        const uint32_t est = min((uint32_t)(1 << D) - 1u, __umulhi(u << D, mg));
        const uint32_t e1 = est > 0u ? est - 1u : 0u;
        const uint32_t B0 = grid_get(tabw, i * n1 + (int)est), B1 = grid_get(tabw, i * n1 + (int)e1);
        return est - (u < B0 ? 1u : 0u) - ((est > 0u && u < B1) ? 1u : 0u);
    };
    auto finest_cell = [&](uint64_t c, uint32_t& cx_, uint32_t& cy_) {
        const uint32_t x = cand_x(c) - kOrbPatchRadius, y = cand_y(c) - kOrbPatchRadius;
        int ix = 0, iy = 0;
        for (int r = 1; r < gx; ++r) ix += x >= rc[4 * r + 3] ? 1 : 0;
        for (int r = 1; r < gy; ++r) iy += y >= rc[4 * (gx + r) + 3] ? 1 : 0;
        cx_ = ((uint32_t)ix << gd.D) + axis_cell(x, ix);
        cy_ = ((uint32_t)iy << gd.D) + axis_cell(y, gx + iy);
    };
    if (!kGrid && tid < 16) s_misc[tid] = 0;   // (the grid form has initialised the scalars above)
    for (int i = tid; i < 5 * Rmax; i += kTreeThreads) s_root[i] = 0;   // (s_rcc follows s_root)
    for (int i = tid; i < NN; i += kTreeThreads) child_cnt[i] = 0;
    __syncthreads();   // (grid form: the pyramid was cleared above, the tables are complete)
    if (kGrid) {
        OVS_TT_MARK();   // tables, cleared pyramid
        // ---- the ONE counting pass: every candidate into its cell of the finest grid, then the coarser grids as sums of four
        const int D = gd.D, WD = gx << D;
        for (uint32_t i0 = tid; i0 < n; i0 += kTreeThreads * 8) {
            uint64_t cc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cc[u] = i0 + u * kTreeThreads < n ? list[i0 + u * kTreeThreads] : 0ull;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * kTreeThreads < n) {
                    uint32_t cx_, cy_;
                    finest_cell(cc[u], cx_, cy_);
                    const int idx = gd.off(D) + (int)cy_ * WD + (int)cx_;
                    atomicAdd(&gridw[idx >> 1], 1u << (16 * (idx & 1)));
                    // the cell travels with the candidate (the node field of its list entry is free in this form) to the last pass
                    if (cache_cells) list[i0 + u * kTreeThreads] = (cc[u] & ~0xFFFFull) | (cy_ << 8) | cx_;
                }
        }
        __syncthreads();
        OVS_TT_MARK();   // counting pass
        // two depths per step (a thread adds a 4 x 2 block of depth d + 2 into two entries of depth d + 1 and one of depth d ... per pair of
        // depth-d entries): half the barriers of one depth per step
        for (int d = D - 1; d >= 0; d -= 2) {
            const bool two = d >= 1;               // this step also produces depth d - 1
            const int Wc = gx << (d + 1), Wd = gx << d, cnt_d = nroot << (2 * d);
            // unit of work: a 2 x 2 block of depth-d entries (= one entry of depth d - 1): rows 2 y, 2 y + 1, columns 2 x, 2 x + 1
            const int Wp = two ? (gx << (d - 1)) : 0, cnt_p = two ? (nroot << (2 * (d - 1))) : 0;
            if (two) {
                for (int e = tid; e < cnt_p; e += kTreeThreads) {
                    const int y = e / Wp, x = e - y * Wp;
                    uint32_t tot = 0;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        // depth-d entries (2 x, 2 y + r) and (2 x + 1, 2 y + r): each the sum of a 2 x 2 block of depth d + 1
                        const int c0 = gd.off(d + 1) + (2 * (2 * y + r)) * Wc + 4 * x;
                        const uint32_t t0 = gridw[c0 >> 1], t1 = gridw[(c0 >> 1) + 1], b0 = gridw[(c0 + Wc) >> 1], b1 = gridw[((c0 + Wc) >> 1) + 1];
                        const uint32_t lo = (t0 & 0xFFFFu) + (t0 >> 16) + (b0 & 0xFFFFu) + (b0 >> 16), hi = (t1 & 0xFFFFu) + (t1 >> 16) + (b1 & 0xFFFFu) + (b1 >> 16);
                        gridw[(gd.off(d) + (2 * y + r) * Wd + 2 * x) >> 1] = lo | (hi << 16);   // (2 x even, off even, Wd even for d >= 1)
                        tot += lo + hi;
                    }
                    // depth d - 1: 16-bit halves of shared words are written by different threads -> an atomic OR into the cleared word
                    const int ip = gd.off(d - 1) + e;
                    atomicOr(&gridw[ip >> 1], tot << (16 * (ip & 1)));
                }
            } else {
                for (int p2 = tid; 2 * p2 < cnt_d; p2 += kTreeThreads) {   // entries 2 p2 and 2 p2 + 1 of depth 0: one word
                    uint32_t word = 0;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int e = 2 * p2 + h;
                        if (e < cnt_d) {
                            const int y = e / Wd, x = e - y * Wd;
                            const int c0 = gd.off(d + 1) + (2 * y) * Wc + 2 * x;
                            const uint32_t top = gridw[c0 >> 1], bot = gridw[(c0 + Wc) >> 1];
                            word |= ((top & 0xFFFFu) + (top >> 16) + (bot & 0xFFFFu) + (bot >> 16)) << (16 * h);
                        }
                    }
                    gridw[(gd.off(d) >> 1) + p2] = word;
                }
            }
            __syncthreads();
        }
    }
    // Every candidate sweep works on kSweepLoads candidates per thread at a time, in three separate steps -- all loads, then all LDS
    // look-ups, then all counter updates and stores -- so that the look-ups of different candidates overlap: written as one loop
    // they form a chain of dependent LDS round trips per candidate, because no LDS read may move across an LDS atomic.
    for (uint32_t i0 = tid; !kGrid && i0 < n; i0 += kTreeThreads * kSweepLoads) {
        uint64_t cc[kSweepLoads];
        uint32_t key[kSweepLoads];
#pragma unroll
        for (int u = 0; u < kSweepLoads; ++u) cc[u] = i0 + u * kTreeThreads < n ? list[i0 + u * kTreeThreads] : 0ull;
#pragma unroll
        for (int u = 0; u < kSweepLoads; ++u) {
            const uint64_t c = cc[u];
            const uint32_t x = cand_x(c) - kOrbPatchRadius, y = cand_y(c) - kOrbPatchRadius;
            const float fx = (float)(int)x, fy = (float)(int)y;
            uint32_t ix = (uint32_t)((double)fx / dx), iy = (uint32_t)((double)fy / dy);
            ix = min(ix, (uint32_t)gx - 1u);
            iy = min(iy, (uint32_t)gy - 1u);
            const uint32_t bx = (uint32_t)(int)(dx * ix), ex = (uint32_t)(int)(dx * (ix + 1)), by = (uint32_t)(int)(dy * iy), ey = (uint32_t)(int)(dy * (iy + 1));
            const uint32_t cx = bx + ((ex - bx + 1u) >> 1), cy = by + ((ey - by + 1u) >> 1);
            key[u] = 4 * (ix + iy * gx) + (x >= cx ? 1u : 0u) + (y >= cy ? 2u : 0u);   // root * 4 + child of the root
        }
#pragma unroll
        for (int u = 0; u < kSweepLoads; ++u) {
            const uint32_t i = i0 + u * kTreeThreads;
            if (i < n) {
                atomicAdd(&s_rcc[key[u]], 1u);   // the root's own count is the sum of its four child counts
                list[i] = (cc[u] & ~0xFFFFull) | (key[u] >> 2);   // the root's raw index; the first pass maps it to the list position through s_root
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t pos = 0;
        for (int r = 0; r < nroot; ++r) {
            const int ix = r % gx, iy = r / gx;
            if (kGrid) {   // the root's children are the depth-1 cells (2 ix + (k & 1), 2 iy + (k >> 1))
#pragma unroll
                for (int k = 0; k < 4; ++k) s_rcc[4 * r + k] = grid_get(gridw, gd.off(1) + (2 * iy + (k >> 1)) * (2 * gx) + 2 * ix + (k & 1));
            }
            const uint32_t cnt = s_rcc[4 * r] + s_rcc[4 * r + 1] + s_rcc[4 * r + 2] + s_rcc[4 * r + 3];
            if (cnt) {
                NodeRec rec;
                rec.xb = (uint32_t)(int)(dx * ix) | ((uint32_t)(int)(dx * (ix + 1)) << 16);
                rec.yb = (uint32_t)(int)(dy * iy) | ((uint32_t)(int)(dy * (iy + 1)) << 16);
                rec.count = cnt;
                rec.pad = ((uint32_t)iy << 12) | (uint32_t)ix;   // depth << 24 | cell row << 12 | cell column (grid form)
                cur[pos] = rec;
#pragma unroll
                for (int k = 0; k < 4; ++k) child_cnt[4 * pos + k] = cnt > 1 ? s_rcc[4 * r + k] : 0u;   // leaves carry no child counts
                s_root[r] = pos++;
            } else
                s_root[r] = 0xFFFFu;
        }
        s_misc[0] = pos;   // list size
    }
    __syncthreads();
    uint32_t size = s_misc[0];
    bool first_pass = true, done_final = false;
    unsigned long long* best = nullptr;

    int phase = 1;
    OVS_TT_MARK();   // kernel entry -> roots built
    for (int guard = 0; guard < 64; ++guard) {   // upstream's while(true); depth is bounded by log2(image size) << 64
        // ---- 1. split coordinates and bounds of every current node (its child counts were produced by the previous sweep)
        for (uint32_t j = tid; j < size; j += kTreeThreads) {
            const NodeRec r = cur[j];
            nxb[j] = r.xb;
            nyb[j] = r.yb;
            tmp[j] = r.count > 1 ? 1u : 0u;   // non-leaf flag
            nch[j] = (child_cnt[4 * j] != 0) + (child_cnt[4 * j + 1] != 0) + (child_cnt[4 * j + 2] != 0) + (child_cnt[4 * j + 3] != 0);
        }
        __syncthreads();
        unsigned long long* const keys = reinterpret_cast<unsigned long long*>(child_nxt);

        OVS_TT_MARK();   // step 1
        // ---- 3. processing order: sidx[r] = node processed r-th, M nodes are split
        uint32_t M;
        if (phase == 1) {
            // all non-leaf nodes in list order
            M = array_excl_scan<kTreeThreads>(tmp, rank, (int)size, s_wave);   // rank[j] = #non-leaf before j
            for (uint32_t j = tid; j < size; j += kTreeThreads) {
                if (tmp[j]) sidx[rank[j]] = j;
                else rank[j] = kNotInS;
            }
            __syncthreads();
        } else {
            // order the pool by (count desc, list position asc); leaves sort to the end. Keys are unique, so a node's place is the number
            // of smaller keys. Up to kRankDirect nodes that number is counted directly (every thread streams the key array from LDS,
            // all lanes reading the same address: two barriers), beyond it a bitonic sort runs (log^2 stages, one barrier each:
            // 55 barriers at 1024 keys, which was a third of a single frame's quad-tree time).
            uint32_t npool_cur;
            if (size <= (uint32_t)kRankDirect) {
                for (uint32_t j = tid; j < size; j += kTreeThreads)
                    keys[j] = tmp[j] ? (((unsigned long long)(0xFFFFFFu - cur[j].count) << 16) | (tie_earlier ? 0xFFFFu - j : j)) : ~0ull;
                npool_cur = array_excl_scan<kTreeThreads>(tmp, rank, (int)size, s_wave);   // rank[] is scratch here; barriers inside publish keys[]
                for (uint32_t j = tid; j < size; j += kTreeThreads) rank[j] = 0;
                __syncthreads();
                // the key array is cut into `parts` slices so that all threads count (size is usually well below the thread count)
                uint32_t S2 = 1, lg = 0;
                while (S2 < size) { S2 <<= 1; ++lg; }
                const uint32_t parts = max(1u, (uint32_t)kTreeThreads >> lg);
                for (uint32_t w = tid; w < S2 * parts; w += kTreeThreads) {
                    const uint32_t j = w & (S2 - 1u), part = w >> lg;
                    if (j >= size || !tmp[j]) continue;
                    const unsigned long long kj = keys[j];
                    const uint32_t lo = part * size / parts, hi = (part + 1u) * size / parts;
                    uint32_t r = 0, i = lo;
                    for (; i + 8 <= hi; i += 8) {
                        unsigned long long kk[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) kk[u] = keys[i + u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) r += kk[u] < kj ? 1u : 0u;
                    }
                    for (; i < hi; ++i) r += keys[i] < kj ? 1u : 0u;
                    if (r) atomicAdd(&rank[j], r);
                }
                __syncthreads();
                for (uint32_t j = tid; j < size; j += kTreeThreads) {
                    if (!tmp[j]) continue;
                    const uint32_t r = rank[j];
                    sidx[r] = j;
                    base[r] = nch[j] - 1u;   // gain of splitting the r-th pool node (nch >= 1 for a non-leaf)
                }
            } else {
                uint32_t P2 = 1;
                while (P2 < size) P2 <<= 1;
                for (uint32_t j = tid; j < P2; j += kTreeThreads) {
                    unsigned long long key = ~0ull;
                    if (j < size && tmp[j]) key = ((unsigned long long)(0xFFFFFFu - cur[j].count) << 16) | (tie_earlier ? 0xFFFFu - j : j);
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
                // pool size, gains in processing order
                npool_cur = array_excl_scan<kTreeThreads>(tmp, rank, (int)size, s_wave);   // rank[] reused below
                for (uint32_t r = tid; r < npool_cur; r += kTreeThreads) {
                    const uint32_t jl = (uint32_t)keys[r] & 0xFFFFu, j = tie_earlier ? 0xFFFFu - jl : jl;
                    sidx[r] = j;
                    base[r] = nch[j] - 1u;   // gain of splitting the r-th pool node (nch >= 1 for a non-leaf)
                }
            }
            // first rank at which #nodes reaches N
            if (tid == 0) s_misc[1] = npool_cur;   // M candidate (atomicMin below)
            __syncthreads();
            array_excl_scan<kTreeThreads>(base, base, (int)npool_cur, s_wave);   // base[r] = gain of ranks < r
            for (uint32_t r = tid; r < npool_cur; r += kTreeThreads) {
                const uint32_t after = size + base[r] + (nch[sidx[r]] - 1u);
                if (after >= N) atomicMin(&s_misc[1], r + 1u);
            }
            __syncthreads();
            M = s_misc[1];
            for (uint32_t j = tid; j < size; j += kTreeThreads) rank[j] = kNotInS;
            __syncthreads();
            for (uint32_t r = tid; r < M; r += kTreeThreads) rank[sidx[r]] = r;
            __syncthreads();
        }

        OVS_TT_MARK();   // step 3
        // ---- 4. creation index of each split node's first child (processing order), positions of survivors (list order)
        for (uint32_t r = tid; r < M; r += kTreeThreads) tmp[r] = nch[sidx[r]];
        for (int i = tid; i < NN; i += kTreeThreads) child_nxt[i] = 0;   // the sort keys are dead: next pass's child counts / `best`
        __syncthreads();
        const uint32_t total_new = array_excl_scan<kTreeThreads>(tmp, tmp, (int)M, s_wave);
        for (uint32_t r = tid; r < M; r += kTreeThreads) base[sidx[r]] = tmp[r];
        __syncthreads();
        for (uint32_t j = tid; j < size; j += kTreeThreads) tmp[j] = (rank[j] == kNotInS) ? 1u : 0u;
        __syncthreads();
        array_excl_scan<kTreeThreads>(tmp, keep_pos, (int)size, s_wave);
        const uint32_t new_size = total_new + (size - M);
        // upstream's stop rules, known before the candidates move: the last pass's sweep feeds find_keypoints_with_max_response directly
        const bool final_pass = N <= new_size || new_size == size;

        OVS_TT_MARK();   // step 4
        // ---- 5. write the new list; count splittable children (the next pool); a surviving node keeps its child counts
        uint32_t my_pool = 0;
        for (uint32_t j = tid; j < size; j += kTreeThreads) {
            const NodeRec r = cur[j];
            if (rank[j] == kNotInS) {
                const uint32_t p = total_new + keep_pos[j];
                keep_pos[j] = p;
                nxt[p] = r;
                if (!final_pass) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) child_nxt[4 * p + k] = child_cnt[4 * j + k];
                }
            } else {
                const uint32_t bx = r.xb & 0xFFFFu, ex = r.xb >> 16, by = r.yb & 0xFFFFu, ey = r.yb >> 16;
                // divide_node: half = ceil((end - begin) / 2.0)
                const uint32_t cx = bx + ((ex - bx + 1u) >> 1), cy = by + ((ey - by + 1u) >> 1);
                uint32_t c = base[j];
                const uint32_t dep = r.pad >> 24, gix = r.pad & 0xFFFu, giy = (r.pad >> 12) & 0xFFFu;
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
                        const uint32_t cgx = 2u * gix + (uint32_t)(k & 1), cgy = 2u * giy + (uint32_t)(k >> 1);
                        ch.pad = ((dep + 1u) << 24) | (cgy << 12) | cgx;
                        nxt[p] = ch;
                        cmap[4 * j + k] = p;
                        my_pool += cnt > 1 ? 1u : 0u;
                        if (kGrid && !final_pass && cnt > 1u) {   // the child's own child counts: cells of depth dep + 2
                            if ((int)dep + 2 > gd.D) {
                                s_misc[2] = 1u;   // a node of the finest depth that may have to be split: the sweep form takes over
                            } else {
                                const int Wq = gx << (dep + 2);
#pragma unroll
                                for (int kk = 0; kk < 4; ++kk)
                                    child_nxt[4 * p + kk] = grid_get(gridw, gd.off(dep + 2) + (int)(2u * cgy + (uint32_t)(kk >> 1)) * Wq + (int)(2u * cgx + (uint32_t)(kk & 1)));
                            }
                        }
                    }
                }
            }
        }
        uint32_t npool_new;
        block_excl_scan<kTreeThreads>(my_pool, s_wave, npool_new);   // barriers inside: nxt / cmap / keep_pos / copied child counts now visible
        if (kGrid && s_misc[2] != 0u) return false;                  // (workgroup-uniform: written before the scan's barriers)

        OVS_TT_MARK();   // step 5
        // ---- 6. one sweep over the candidates: move each to its new node and count it into that node's own children (the next pass's
        //         child counts); in the last pass: per-node max (score, first in emission order) instead
        if (final_pass) {
            // The last pass may end with up to max_nodes = 4 * NCmax nodes (e.g. 36 root patches all split when the level wants 7
            // keypoints): `best` takes BOTH child-count arrays (2 * NN words = max_nodes 64-bit entries); the current counts were last
            // read in step 5, behind the barriers of the scan above. (Found by tools/fuzz_parity.py on a 1860 x 136 image.)
            best = reinterpret_cast<unsigned long long*>(smem);
            for (uint32_t j = tid; j < new_size; j += kTreeThreads) best[j] = 0ull;
            __syncthreads();
        }
        const uint32_t ncx_cells = (uint32_t)g.ncx;
        if (kGrid && final_pass) {
            // the final nodes mark their cells (node + 1 at the node's own depth; the leaves partition the plane, so every candidate has exactly
            // one marked ancestor cell), then ONE pass over the candidates: all D + 1 ancestor cells are read, the marked one names the node
            {
                const int words = gd.off(gd.D + 1) / 2;   // the pyramid only: the tables behind it may still be read (finest_cell)
                for (int i = tid; i < words / 4; i += kTreeThreads) reinterpret_cast<uint4*>(gridw)[i] = uint4{0u, 0u, 0u, 0u};
                if (tid < (words & 3)) gridw[(words & ~3) + tid] = 0u;
            }
            __syncthreads();
            for (uint32_t j = tid; j < new_size; j += kTreeThreads) {
                const uint32_t pad = nxt[j].pad, dep = pad >> 24;
                const int idx = gd.off(dep) + (int)((pad >> 12) & 0xFFFu) * (gx << dep) + (int)(pad & 0xFFFu);
                atomicOr(&gridw[idx >> 1], (j + 1u) << (16 * (idx & 1)));
            }
            __syncthreads();
            OVS_TT_MARK();   // final nodes' marks
            const int D = gd.D;
            for (uint32_t i0 = tid; i0 < n; i0 += kTreeThreads * 4) {
                uint64_t cc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) cc[u] = i0 + u * kTreeThreads < n ? list[i0 + u * kTreeThreads] : 0ull;
                uint32_t node[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint32_t cx_, cy_;
                    if (cache_cells) {
                        cx_ = (uint32_t)cc[u] & 0xFFu;
                        cy_ = ((uint32_t)cc[u] >> 8) & 0xFFu;
                    } else {
                        finest_cell(cc[u], cx_, cy_);
                    }
                    uint32_t m = 0;
#pragma unroll
                    for (int d = 0; d <= 7; ++d)   // all ancestor cells are read at once (independent look-ups); depths beyond D do not exist
                        if (d <= D) m |= grid_get(gridw, gd.off(d) + (int)(cy_ >> (D - d)) * (gx << d) + (int)(cx_ >> (D - d)));
                    node[u] = m - 1u;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u * kTreeThreads < n && node[u] < new_size) {
                        const uint64_t c = cc[u];
                        const unsigned long long key = ((unsigned long long)(cand_score(c) + 1u) << 32) | (0xFFFFFFFFu - cand_order(cand_x(c), cand_y(c), ncx_cells));
                        __hip_atomic_fetch_max(&best[node[u]], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
            }
        }
        for (uint32_t i0 = tid; !kGrid && i0 < n; i0 += kTreeThreads * kSweepLoads) {
            uint64_t cc[kSweepLoads];
            uint32_t pp[kSweepLoads], gc[kSweepLoads];   // new node; counter of the new node's child the candidate falls in (or ~0)
#pragma unroll
            for (int u = 0; u < kSweepLoads; ++u) cc[u] = i0 + u * kTreeThreads < n ? list[i0 + u * kTreeThreads] : 0ull;
            // look-ups in three rounds of independent LDS reads (node -> rank / survivor position / bounds -> child position / child
            // count); everything is read unconditionally (in bounds for any node) and selected afterwards
            uint32_t nd[kSweepLoads], rk[kSweepLoads], kp[kSweepLoads], xb[kSweepLoads], yb[kSweepLoads], ck[kSweepLoads];
#pragma unroll
            for (int u = 0; u < kSweepLoads; ++u) {
                const uint32_t raw = (uint32_t)cc[u] & 0xFFFFu;
                nd[u] = first_pass ? s_root[min(raw, (uint32_t)Rmax - 1u)] : raw;
                if (!(i0 + u * kTreeThreads < n)) nd[u] = 0;
            }
#pragma unroll
            for (int u = 0; u < kSweepLoads; ++u) {
                rk[u] = rank[nd[u]];
                kp[u] = keep_pos[nd[u]];
                xb[u] = nxb[nd[u]];
                yb[u] = nyb[nd[u]];
            }
            uint32_t xx[kSweepLoads], yy[kSweepLoads], ccx[kSweepLoads], ccy[kSweepLoads];
#pragma unroll
            for (int u = 0; u < kSweepLoads; ++u) {
                const uint32_t cx = (xb[u] & 0xFFFFu) + (((xb[u] >> 16) - (xb[u] & 0xFFFFu) + 1u) >> 1);
                const uint32_t cy = (yb[u] & 0xFFFFu) + (((yb[u] >> 16) - (yb[u] & 0xFFFFu) + 1u) >> 1);
                xx[u] = cand_x(cc[u]) - kOrbPatchRadius;
                yy[u] = cand_y(cc[u]) - kOrbPatchRadius;
                const uint32_t k = (xx[u] >= cx ? 1u : 0u) + (yy[u] >= cy ? 2u : 0u);
                ck[u] = 4 * nd[u] + k;
                // the chosen child's own split point
                const uint32_t cbx = (k & 1u) ? cx : (xb[u] & 0xFFFFu), cex = (k & 1u) ? (xb[u] >> 16) : cx;
                const uint32_t cby = (k & 2u) ? cy : (yb[u] & 0xFFFFu), cey = (k & 2u) ? (yb[u] >> 16) : cy;
                ccx[u] = cbx + ((cex - cbx + 1u) >> 1);
                ccy[u] = cby + ((cey - cby + 1u) >> 1);
            }
            uint32_t cm[kSweepLoads], cn[kSweepLoads];
#pragma unroll
            for (int u = 0; u < kSweepLoads; ++u) {
                cm[u] = cmap[ck[u]];
                cn[u] = child_cnt[ck[u]];
            }
#pragma unroll
            for (int u = 0; u < kSweepLoads; ++u) {
                const bool split = rk[u] != kNotInS && i0 + u * kTreeThreads < n;
                pp[u] = split ? cm[u] : kp[u];
                gc[u] = (split && !final_pass && cn[u] > 1u) ? 4 * pp[u] + (xx[u] >= ccx[u] ? 1u : 0u) + (yy[u] >= ccy[u] ? 2u : 0u) : ~0u;
            }
#pragma unroll
            for (int u = 0; u < kSweepLoads; ++u) {
                const uint32_t i = i0 + u * kTreeThreads;
                if (i < n) {
                    const uint64_t c = cc[u];
                    if (final_pass) {
                        const unsigned long long key = ((unsigned long long)(cand_score(c) + 1u) << 32) | (0xFFFFFFFFu - cand_order(cand_x(c), cand_y(c), ncx_cells));
                        __hip_atomic_fetch_max(&best[pp[u]], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        if (gc[u] != ~0u) atomicAdd(&child_nxt[gc[u]], 1u);
                        list[i] = (c & ~0xFFFFull) | pp[u];
                    }
                }
            }
        }
        __syncthreads();
        {
            NodeRec* t = cur;
            cur = nxt;
            nxt = t;
            uint32_t* tc = child_cnt;
            child_cnt = child_nxt;
            child_nxt = tc;
        }
        OVS_TT_MARK();   // step 6 (the sweep)
        size = new_size;
        first_pass = false;
        if (final_pass) {
            done_final = true;
            break;
        }
        if (phase == 1 && N < size + switch_factor * npool_new) phase = 2;
    }

    // ---- find_keypoints_with_max_response; output in list order
    const uint32_t ncx = (uint32_t)g.ncx;
    if (kGrid && !done_final) return false;   // (the grid form keeps no node ids in the candidate list)
    if (!done_final) {   // not reachable (the loop ends through its last pass); kept so that an exhausted guard still yields keypoints
        best = reinterpret_cast<unsigned long long*>(smem);   // both child-count arrays: max_nodes entries
        for (uint32_t j = tid; j < size; j += kTreeThreads) best[j] = 0ull;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kTreeThreads) {
            const uint64_t c = list[i];
            const uint32_t x = cand_x(c), y = cand_y(c);
            const unsigned long long key = ((unsigned long long)(cand_score(c) + 1u) << 32) | (0xFFFFFFFFu - cand_order(x, y, ncx));
            __hip_atomic_fetch_max(&best[(uint32_t)c & 0xFFFFu], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
    }
    const uint32_t kp_cap = (uint32_t)g.kp_cap;
    for (uint32_t j = tid; j < size && j < kp_cap; j += kTreeThreads) {
        const unsigned long long key = best[j];
        const uint32_t score = (uint32_t)(key >> 32) - 1u;
        const uint32_t ord = 0xFFFFFFFFu - (uint32_t)key;
        const uint32_t cell = ord >> 12, ci = cell / ncx, cj = cell - ci * ncx;
        const uint32_t x = kDetOrigin + 64u * cj + (ord & 63u), y = kDetOrigin + 64u * ci + ((ord >> 6) & 63u);
        out[j] = cand_pack(x, y, score, 0);
    }
    OVS_TT_MARK();
    OVS_TT_PRINT(n);
    if (tid == 0) lvl_count[frame * L + level] = size < kp_cap ? size : kp_cap;
    return true;
}

template <int kTreeThreads>
__global__ __launch_bounds__(kTreeThreads) void k_tree(const FrameGeo* __restrict__ geo, uint64_t* __restrict__ cand,
                                                      size_t cand_frame_entries, const uint32_t* __restrict__ cand_count,
                                                      NodeRec* __restrict__ nodes, size_t node_frame_entries,
                                                      uint64_t* __restrict__ lvl_kps, uint32_t* __restrict__ lvl_count, int NCmax,
                                                      int P2max, int Rmax, int grid_words, int grid_depth, int level_lo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int level = level_lo + (int)blockIdx.x, frame = blockIdx.y;   // the launch covers levels [level_lo, level_lo + gridDim.x)
    // the grid form first (round 6); the sweep form of rounds 1-5 when the grid cannot finish the problem (rare: see tree_problem)
    if (grid_words > 0 && tree_problem<kTreeThreads, true>(smem, geo, cand, cand_frame_entries, cand_count, nodes, node_frame_entries, lvl_kps, lvl_count,
                                                           NCmax, P2max, Rmax, grid_words, grid_depth, level, frame, level_lo))
        return;
    __syncthreads();
    tree_problem<kTreeThreads, false>(smem, geo, cand, cand_frame_entries, cand_count, nodes, node_frame_entries, lvl_kps, lvl_count, NCmax, P2max, Rmax,
                                      grid_words, grid_depth, level, frame, level_lo);
}

static size_t tree_lds_bytes(int NCmax, int P2max, int Rmax) {
    return (size_t)(12 * NCmax) * 4 + (size_t)NCmax * 5 * 4 + (size_t)P2max * 4 + (size_t)NCmax * 2 * 4 + 16 * 4 + (16 + 5 * (size_t)Rmax) * 4;
}
// root patches of the widest root grid, at least 64 (a multiple of 4)
static int tree_root_cap(const FrameGeo& hgeo) {
    int r = 64;
    for (int l = 0; l < hgeo.num_levels; ++l) r = std::max(r, hgeo.lv[l].gx * hgeo.lv[l].gy);
    return (r + 3) & ~3;
}

// LDS the quad-tree kernel needs for this geometry (its per-node arrays are sized by the largest level); build_geometry refuses
// parameter sets that exceed the 160 KB a workgroup can have (about 2100 keypoints on ONE level, e.g. 9 600 features over 8 levels)
size_t tree_lds_bytes_for(const FrameGeo& hgeo) {
    int NCmax = 0;
    for (int l = 0; l < hgeo.num_levels; ++l) NCmax = std::max(NCmax, hgeo.lv[l].max_nodes / 4);
    NCmax = (NCmax + 3) & ~3;
    int P2max = 1;
    while (P2max < NCmax) P2max <<= 1;
    return tree_lds_bytes(NCmax, P2max, tree_root_cap(hgeo));
}

hipError_t launch_tree(const FrameGeo& hgeo, const DevBuffers& d, int batch, hipStream_t s, int level_lo, int n_levels) {
    if (n_levels < 0) n_levels = hgeo.num_levels - level_lo;
    if (n_levels <= 0 || batch <= 0) return hipSuccess;
    int NCmax = 0;
    for (int l = 0; l < hgeo.num_levels; ++l) NCmax = std::max(NCmax, hgeo.lv[l].max_nodes / 4);
    NCmax = (NCmax + 3) & ~3;
    int P2max = 1;
    while (P2max < NCmax) P2max <<= 1;
    const int Rmax = tree_root_cap(hgeo);
    const size_t lds_base = tree_lds_bytes(NCmax, P2max, Rmax);
    // The kernel is bound by one workgroup's instruction latency, not by memory: with few problems in the launch (a tracker's single frame)
    // 1024 threads halve it; with many, 512 threads let more problems share a CU and win (0.247 vs 0.278 ms per 128 frames).
    const bool few = (long long)n_levels * batch <= 64;
    // the count pyramid of the grid form (tree_grid_depth per level): whatever the launch's levels need, and what the 160 KB leave
    size_t grid_words = 0;
    static const int forced = [] {
        const char* e = std::getenv("OVS_TREE_GRID");   // A/B switch: 0 = the sweep form of rounds 1-5 only, d = the finest grid's depth
        return e ? std::atoi(e) : -1;
    }();
    const int grid_depth = forced > 0 ? std::min(forced, 7) : 0;
    // A tracker's single frame keeps the sweep form: one workgroup has a CU to itself there and both forms take the same ~56 us (measured
    // 56.9 against 55.5: its passes over the candidates are replaced by tables, a cleared pyramid and a second pass of the same latency), while
    // in a batch, where the problems share the CUs' vector ALUs, the grid form's fewer instructions count (0.307 -> 0.218 ms per 256 frames).
    if (forced != 0 && (!few || forced > 0)) {
        for (int l = level_lo; l < level_lo + n_levels; ++l) {
            const size_t roots = (size_t)hgeo.lv[l].gx * hgeo.lv[l].gy, axes = (size_t)hgeo.lv[l].gx + hgeo.lv[l].gy;
            const int D = grid_depth > 0 ? grid_depth : tree_grid_depth(hgeo.lv[l].n_keypts, (int)roots);
            size_t entries = 0;
            for (int d = 0; d <= D; ++d) entries += ((roots << (2 * d)) + 1) & ~(size_t)1;
            grid_words = std::max(grid_words, entries / 2 + 4 * axes + (axes * ((1u << D) + 1) + 1) / 2 + 4);   // + the root columns' / rows' tables
        }
        const size_t room = lds_base + 1024 < kMaxLdsPerWorkgroup ? (kMaxLdsPerWorkgroup - lds_base - 1024) / 4 : 0;
        grid_words = std::min(std::min(grid_words, room), (size_t)(24 * 1024 / 4)) & ~(size_t)3;   // (at most 24 KB: three problems still share a CU in a batch)
    }
    const size_t lds = lds_base + 4 * grid_words;
    static LdsAttrCache configured[2];   // per device
    const void* fn = few ? reinterpret_cast<const void*>(k_tree<kTreeThreadsFew>) : reinterpret_cast<const void*>(k_tree<kTreeThreadsBatch>);
    {
        hipError_t e = ensure_dynamic_lds(fn, lds, configured[few ? 1 : 0]);
        if (e != hipSuccess) return e;
    }
    dim3 grid(n_levels, batch);
    if (few)
        hipLaunchKernelGGL(k_tree<kTreeThreadsFew>, grid, dim3(kTreeThreadsFew), lds, s, d.geo, d.cand, d.cand_frame_entries, d.cand_count,
                           reinterpret_cast<NodeRec*>(d.nodes), d.node_frame_entries, d.lvl_kps, d.lvl_count, NCmax, P2max, Rmax, (int)grid_words, grid_depth, level_lo);
    else
        hipLaunchKernelGGL(k_tree<kTreeThreadsBatch>, grid, dim3(kTreeThreadsBatch), lds, s, d.geo, d.cand, d.cand_frame_entries, d.cand_count,
                           reinterpret_cast<NodeRec*>(d.nodes), d.node_frame_entries, d.lvl_kps, d.lvl_count, NCmax, P2max, Rmax, (int)grid_words, grid_depth, level_lo);
    return hipGetLastError();
}

}   // namespace ovs
