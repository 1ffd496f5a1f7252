// orb_fast.hip -- A2 + A3: orb_extractor::compute_fast_keypoints' cell loop and cv::FAST(TYPE_9_16, nonmax=true)
// (expected: src/openvslam/feature/orb_extractor.cc; OpenCV features2d fast.cpp / fast_score.cpp).
//
// Reformulation (proved in DESIGN.md, checked bit-for-bit against the oracle's literal OpenCV restatement):
//   S(p)      = max( max_arcs min_{9 contiguous}(v - ring), max_arcs min_{9 contiguous}(ring - v) )
//   corner(t) <=> S(p) > t,  cornerScore = S(p) - 1 (independent of t for detected corners),
//   NMS keeps p <=> S(p) > t and S(p) > S(q) for the 8 neighbours q inside the SAME cell's testable area
//   (neighbours with S(q) <= t score 0 upstream, but then S(q) <= t < S(p) anyway).
// So the cell uses ini_fast_thr if any NMS survivor exceeds it, else min_fast_thr ("if keypts_in_cell.empty() retry").
// The 64x64 testable areas of neighbouring cells tile the level exactly (cell stride 64, overlap 6 = two 3-px dead
// frames), so every pixel is scored once.
//
// S network (v2 ran it on every pixel as 75 packed ops per PAIR of pixels; v3 runs it in 32-bit ops on the pre-test survivors only).
// With r[0..15] the raw ring values and j even:
//   the two 9-arcs {j-1..j+7} and {j..j+8} share the 8-window W_j = {j..j+7}, so
//   min_arcs max_9 r = min_j max( max W_j, min(r[j-1], r[j+8]) ),   max_arcs min_9 r = max_j min( min W_j, max(r[j-1], r[j+8]) )
//   S = max( c - min_arcs max_9 r,  max_arcs min_9 r - c ).
//   The eight even windows come from 8 pair + 8 quad min/max, the oct step folds into the arc step.
//   Checked against the 16-arc definition on random rings (tests/test_oracle_kat.py).
//
// Mapping: one 256-thread workgroup per cell; the <=70x70 u8 tile is staged in LDS with aligned 16-byte loads (cell
// origin x = 19+64j => tile origin 16+64j). Two-stage evaluation (v3; v2 scored every pixel with the 75-op network above, which
// is still the definition of S): thread (run, rp) runs the 16-op diameter test on pixels [8*run, 8*run+8) of rows 2*rp and
// 2*rp+1 as packed pairs; the 2.4 % (level 0) to 26 % (level 7) that pass are compacted into an LDS list and scored exactly, one pixel per lane, with the
// same network in 32-bit ops; NMS reads the sparse score map (unscored pixels hold 0, and indeed have S <= t). The cell is
// processed with ini_fast_thr and, only if no NMS survivor came out, again with min_fast_thr. Survivors are appended to the
// (frame, level) candidate list with ONE global atomic per workgroup. List order is irrelevant downstream (the quad-tree
// kernel uses counts and an explicit emission-order key).
#include "ovs_common.h"

namespace ovs {

typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int kTileRowsMax = kCellSize + kCellOverlap;   // 70
constexpr int kTileWords = 20;                           // 80-byte LDS pitch
constexpr int kSmapWords = 18;                           // 72-byte pitch: 4 + 64 + 4
constexpr int kSmapRows = 66;                            // 1 + 64 + 1

__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s16x2 vmin(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 vmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }

// bytes M and M+1 (M in 0..6) of the 8-byte window {hi,lo} zero-extended into the two 16-bit halves
template <int M>
__device__ __forceinline__ uint32_t pick2(uint32_t hi, uint32_t lo) {
    constexpr uint32_t sel = (uint32_t)M | (0x0cu << 8) | ((uint32_t)(M + 1) << 16) | (0x0cu << 24);
    return __builtin_amdgcn_perm(hi, lo, sel);
}

// pixels at byte offsets Q, Q+1 of window row R
template <int Q, int R>
__device__ __forceinline__ s16x2 window_pair(const uint32_t (&w)[8][5]) {
    return as_s16x2(pick2<(Q & 3)>(w[R][(Q >> 2) + ((Q & 3) == 3 ? 1 : 0)], w[R][Q >> 2]));
}

// Packed 2 x u16 min / max of values 0..255 through the f16 pipe: such bit patterns are positive f16 denormals, ordered like
// the integers and returned unflushed (checked exhaustively on gfx950 by tools/ubench/valu_rate.hip). (v2 needed this pipe for
// its three-input packed min / max, v_pk_minimum3_f16 / v_pk_maximum3_f16, which the integer pipe lacks.)
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h16x2 as_h(s16x2 v) { return __builtin_bit_cast(h16x2, v); }
__device__ __forceinline__ s16x2 as_s(h16x2 v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ s16x2 umin2(s16x2 a, s16x2 b) { return as_s(__builtin_elementwise_minimum(as_h(a), as_h(b))); }
__device__ __forceinline__ s16x2 umax2(s16x2 a, s16x2 b) { return as_s(__builtin_elementwise_maximum(as_h(a), as_h(b))); }

// ---- sparse evaluation -------------------------------------------------------------------------------------------------------
// S(p) > t needs 9 contiguous ring pixels all brighter than c + t (or all darker than c - t). A 9-arc of the 16-ring contains at least
// one end of EVERY diameter (i, i + 8), so for the four even diameters (cardinals and diagonals; OpenCV's own pre-test walks all eight)
//   bright: min_i max(r[i], r[i+8]) > c + t      dark: max_i min(r[i], r[i+8]) < c - t      for i in {0, 2, 4, 6}
// is necessary. 12 packed min/max + 4 more ops and 9 byte extractions per pixel pair instead of 75 + 17, and on BASELINE-like frames it
// rejects 97.6 % of the pixels at level 0 and 74 % at level 7 for ini_fast_thr = 20 (92 % overall; true corners: 4.4 %). The two
// cardinal diameters alone (v3.0) let 13 % through: the extra 10 ops here save 42 % of the exact evaluations. Only the survivors get
// the exact S (one pixel per lane, 32-bit ops).
template <int P, int R0>
__device__ __forceinline__ uint32_t diameter_test_pair(const uint32_t (&w)[8][5], s16x2 thrv) {
    const s16x2 c = window_pair<6 + P, R0 + 3>(w);
    const s16x2 a0 = window_pair<6 + P, R0 + 6>(w), b0 = window_pair<6 + P, R0 + 0>(w);           // ring 0 / 8:  (0, +3) / (0, -3)
    const s16x2 a2 = window_pair<6 + P + 2, R0 + 5>(w), b2 = window_pair<6 + P - 2, R0 + 1>(w);   // ring 2 / 10: (+2, +2) / (-2, -2)
    const s16x2 a4 = window_pair<6 + P + 3, R0 + 3>(w), b4 = window_pair<6 + P - 3, R0 + 3>(w);   // ring 4 / 12: (+3, 0) / (-3, 0)
    const s16x2 a6 = window_pair<6 + P + 2, R0 + 1>(w), b6 = window_pair<6 + P - 2, R0 + 5>(w);   // ring 6 / 14: (+2, -2) / (-2, +2)
    const s16x2 bc = umin2(umin2(umax2(a0, b0), umax2(a2, b2)), umin2(umax2(a4, b4), umax2(a6, b6)));
    const s16x2 dc = umax2(umax2(umin2(a0, b0), umin2(a2, b2)), umax2(umin2(a4, b4), umin2(a6, b6)));
    const s16x2 m = vmax(bc - c, c - dc);
    return as_u32((thrv - m) >> 15);   // each half all-ones iff m > thr
}

// 16-bit VOP2 min / max: one wave-instruction per ~2.3 cycles on gfx950 against ~4.1 for v_min_u32 / v_max_u32 (tools/ubench/valu_rate.hip);
// operands here are u8 values, and gfx9 16-bit ops zero the destination's upper half, so results mix freely with 32-bit arithmetic
__device__ __forceinline__ uint32_t mn16(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t mx16(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// exact S of one pixel; p = LDS address of the top-left byte of its 7x7 neighbourhood in the staged tile (pitch kTileWords * 4)
__device__ __forceinline__ uint32_t fast_strength_one(const uint8_t* p) {
    constexpr int kP = kTileWords * 4;
    const uint32_t c = p[3 * kP + 3];
    uint32_t r[16];
    r[0] = p[6 * kP + 3];
    r[1] = p[6 * kP + 4];
    r[2] = p[5 * kP + 5];
    r[3] = p[4 * kP + 6];
    r[4] = p[3 * kP + 6];
    r[5] = p[2 * kP + 6];
    r[6] = p[1 * kP + 5];
    r[7] = p[0 * kP + 4];
    r[8] = p[0 * kP + 3];
    r[9] = p[0 * kP + 2];
    r[10] = p[1 * kP + 1];
    r[11] = p[2 * kP + 0];
    r[12] = p[3 * kP + 0];
    r[13] = p[4 * kP + 0];
    r[14] = p[5 * kP + 1];
    r[15] = p[6 * kP + 2];
    uint32_t pmx[8], pmn[8], qmx[8], qmn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        pmx[i] = mx16(r[2 * i], r[2 * i + 1]);
        pmn[i] = mn16(r[2 * i], r[2 * i + 1]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        qmx[i] = mx16(pmx[i], pmx[(i + 1) & 7]);
        qmn[i] = mn16(pmn[i], pmn[(i + 1) & 7]);
    }
    uint32_t min_a = 255u, max_b = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t ea = r[(2 * i + 15) & 15], eb = r[(2 * i + 8) & 15];
        min_a = mn16(min_a, mx16(mx16(qmx[i], qmx[(i + 2) & 7]), mn16(ea, eb)));
        max_b = mx16(max_b, mn16(mn16(qmn[i], qmn[(i + 2) & 7]), mx16(ea, eb)));
    }
    const int s = max((int)c - (int)min_a, (int)max_b - (int)c);
    return (uint32_t)max(s, 0);
}

constexpr int kMaxSurvivors = 1024;   // NMS survivors are pairwise non-adjacent: at most 32 x 32 per 64 x 64 cell

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fast_cells(const FrameGeo* __restrict__ geo, const uint8_t* __restrict__ img0, size_t stride0,
                                                   size_t frame_stride0, const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                   uint64_t* __restrict__ cand, size_t cand_frame_entries,
                                                   uint32_t* __restrict__ cand_count, const uint8_t* __restrict__ mask,
                                                   int batch, int cell_lo, int n_cells) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[kTileRowsMax][kTileWords];
    __shared__ __attribute__((aligned(16))) uint32_t smap[kSmapRows][kSmapWords];
    __shared__ uint16_t clist[4][kCellSize * kCellSize / 4];   // per wave: its pixels that passed the diameter test, (y << 8) | x
    __shared__ uint32_t wave_cnt[4];
    __shared__ uint32_t n_out, list_base;

    // NMS survivors, (S << 16) | (y << 8) | x, are collected over the tile: its last reader (the exact scoring) is a barrier behind by
    // then, and the pass that re-reads the tile (min_fast_thr) only runs when no survivor was written. 18.9 KB of LDS -> 8 workgroups/CU.
    static_assert(sizeof(tile) >= kMaxSurvivors * sizeof(uint32_t), "survivor list aliases the tile");
    uint32_t* const olist = &tile[0][0];
    const int tid = threadIdx.x;
    const int L = geo->num_levels;
    // XCD-aware work order. Workgroup b runs on XCD b % 8 (each XCD has its own L2), so XCD k takes the k-th CONTIGUOUS eighth of every
    // frame's cells: neighbouring cells, which share tile halo lines, then hit the same L2 instead of fetching the line once per XCD
    // (fabric traffic 3.0x -> see profiles/). Inside an XCD's share the FRAME index runs fastest: the workgroups in flight at any moment
    // then reserve list space on `batch` times as many (frame, level) counters -- with frame-major order all of them hammered the 8
    // counters of one frame, and same-address atomics at the fabric side cost ~0.05 ms per launch. The grid is 8 * per_xcd * batch.
    // The launch covers cells [cell_lo, cell_lo + n_cells): all of them, or one level range (level 0 runs beside the pyramid, see orb_api.hip).
    const int per_xcd = (n_cells + 7) >> 3;
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    const int slot = idx / batch, frame = idx - slot * batch;
    if (xcd * per_xcd + slot >= n_cells) return;
    const int cell_id = cell_lo + xcd * per_xcd + slot;
    // The prologue is a chain of dependent scalar loads in front of the tile fetch, and nothing else can run in this workgroup until the
    // tile is in LDS: keep the chain at three round trips (kernel arguments -> header + level table -> the level's geometry).
    int level = 0;
#pragma unroll
    for (int l = 1; l < OVS_MAX_LEVELS; ++l) level += (cell_id >= geo->cell_base_tab[l]) ? 1 : 0;
    const LevelGeo& g = geo->lv[level];
    const int g_ncx = g.ncx, g_cell_base = g.cell_base, g_max_bx = g.max_bx, g_max_by = g.max_by, g_pitch = g.pitch;
    const float g_inv_ncx = g.inv_ncx, scale = g.scale;
    const int64_t g_plane_off = g.plane_off;
    const int cell = cell_id - g_cell_base;
    const int ci = (int)(((float)cell + 0.5f) * g_inv_ncx), cj = cell - ci * g_ncx;   // exact: cell < 2^22 (see orb_pyramid.hip)

    const int min_x = kOrbPatchRadius + cj * kCellSize, min_y = kOrbPatchRadius + ci * kCellSize;
    int max_x = min_x + kCellSize + kCellOverlap, max_y = min_y + kCellSize + kCellOverlap;
    if (g_max_bx < max_x) max_x = g_max_bx;
    if (g_max_by < max_y) max_y = g_max_by;
    const int cw = max_x - min_x, ch = max_y - min_y;
    const int iw = cw - 6, ih = ch - 6;   // testable area of this cell (> 0 for every valid cell)

    const uint8_t* img;
    int pitch;
    if (level == 0) {
        img = img0 + (size_t)frame * frame_stride0;
        pitch = (int)stride0;
    } else {
        img = pyr + (size_t)frame * pyr_frame_bytes + g_plane_off;
        pitch = g_pitch;
    }

    // ---- fetch the tile: tile byte u of row r <-> image (min_x - 3 + u, min_y + r); min_x - 3 = 16 + 64*cj is 16-byte aligned. The 70 x 5
    //      16-byte chunks go two per thread (the second for tid < 94), BOTH requested before either is waited for.
    const int ax0 = min_x - 3;
    const bool vec16 = ((pitch & 15) == 0) && ((reinterpret_cast<uintptr_t>(img) & 15) == 0);
    auto fetch_chunk = [&](int idx) -> uint4 {
        const int r = idx / 5, q = idx - r * 5;
        uint4 v = {0u, 0u, 0u, 0u};
        const int gx = ax0 + 16 * q;
        if (r < ch) {
            const uint8_t* p = img + (size_t)(min_y + r) * pitch + gx;
            if (vec16 && gx + 16 <= pitch) {
                v = *reinterpret_cast<const uint4*>(p);
            } else {   // 4-byte aligned base/stride (enforced by the ABI), row tail
                const uint32_t* p4 = reinterpret_cast<const uint32_t*>(p);
                if (gx + 4 <= pitch) v.x = p4[0];
                if (gx + 8 <= pitch) v.y = p4[1];
                if (gx + 12 <= pitch) v.z = p4[2];
                if (gx + 16 <= pitch) v.w = p4[3];
            }
        }
        return v;
    };
    constexpr int kChunks = kTileRowsMax * 5;
    const uint4 va = fetch_chunk(tid);
    uint4 vb = {0u, 0u, 0u, 0u};
    if (tid + 256 < kChunks) vb = fetch_chunk(tid + 256);

    const uint8_t* fmask = mask ? mask + (size_t)frame * frame_stride0 : nullptr;   // same layout as the level-0 frames
    // upstream: skip the cell if one of its corners is masked (mask is indexed in level-0 coordinates, float scale, trunc)
    if (fmask) {
        auto in_mask = [&](unsigned y, unsigned x) {
            return fmask[(size_t)(unsigned)(y * scale) * stride0 + (unsigned)(x * scale)] == 0;
        };
        if (in_mask(min_y, min_x) || in_mask(max_y, min_x) || in_mask(min_y, max_x) || in_mask(max_y, max_x)) return;
    }

    // the score map holds S for the pixels that were evaluated and 0 elsewhere (pixel (x, y) at byte 4 + x of row y + 1)
    uint32_t* const smap_flat = &smap[0][0];
    static_assert((kSmapRows * kSmapWords) % 4 == 0, "score map is cleared with 16-byte stores");
    for (int i = tid; i < kSmapRows * kSmapWords / 4; i += 256) reinterpret_cast<uint4*>(smap_flat)[i] = uint4{0u, 0u, 0u, 0u};
    if (tid == 0) n_out = 0;
    if (tid < 4) wave_cnt[tid] = 0;
    {
        const int r = tid / 5, q = tid - r * 5;
        *reinterpret_cast<uint4*>(&tile[r][4 * q]) = va;
        if (tid + 256 < kChunks) {
            const int r2 = (tid + 256) / 5, q2 = (tid + 256) - r2 * 5;
            *reinterpret_cast<uint4*>(&tile[r2][4 * q2]) = vb;
        }
    }

    const int run = tid & 7, rp = tid >> 3;
    const int c0 = run * 8, row0 = 2 * rp;
    const int lane = tid & 63, wv = tid >> 6;
    const uint8_t* const tbytes = reinterpret_cast<const uint8_t*>(&tile[0][0]);
    uint8_t* const sbytes = reinterpret_cast<uint8_t*>(smap_flat);
    __syncthreads();

    // candidate-mask layout: pair j (pixels c0 + 2j, c0 + 2j + 1) of row row0 -> bits j and 16 + j; of row row0 + 1 -> bits 4 + j and
    // 20 + j. Where the level border clips the testable area (workgroup-uniform, ~9 % of the cells) the pixels outside are masked. The
    // empty asm keeps this a real branch: hipcc otherwise hoists the block out of the loop below AND evaluates it speculatively for
    // every cell (~60 VALU per thread, a tenth of the kernel's instructions -- found in the ISA, round 2).
    uint32_t valid = ~0u;
    if (iw < kCellSize || ih < kCellSize) {
        asm volatile("" ::: "memory");
        valid = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = c0 + 2 * j;
            const uint32_t colm = (x < iw ? 1u : 0u) | (x + 1 < iw ? 0x10000u : 0u);
            valid |= (row0 < ih ? colm : 0u) << j;
            valid |= (row0 + 1 < ih ? colm : 0u) << (4 + j);
        }
    }

    int thr = geo->ini_thr;
    for (;;) {
        // ---- 1. diameter test on packed pairs -> candidate mask. It reads this part of the thread's 8x5-word window (re-read in the rare
        //         second pass rather than kept in 28 registers across the scoring): words 1..3 of every row, words 0 and 4 of rows 3, 4
        uint32_t cmask = 0;
        {
            uint32_t w[8][5];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint2 b = *reinterpret_cast<const uint2*>(&tile[row0 + r][2 * run + 2]);
                w[r][1] = tile[row0 + r][2 * run + 1];
                w[r][2] = b.x;
                w[r][3] = b.y;
                if (r == 3 || r == 4) {
                    w[r][0] = tile[row0 + r][2 * run];
                    w[r][4] = tile[row0 + r][2 * run + 4];
                }
            }
            const s16x2 thrv = {(short)thr, (short)thr};
            cmask |= diameter_test_pair<0, 0>(w, thrv) & 0x00010001u;
            cmask |= diameter_test_pair<2, 0>(w, thrv) & 0x00020002u;
            cmask |= diameter_test_pair<4, 0>(w, thrv) & 0x00040004u;
            cmask |= diameter_test_pair<6, 0>(w, thrv) & 0x00080008u;
            cmask |= diameter_test_pair<0, 1>(w, thrv) & 0x00100010u;
            cmask |= diameter_test_pair<2, 1>(w, thrv) & 0x00200020u;
            cmask |= diameter_test_pair<4, 1>(w, thrv) & 0x00400040u;
            cmask |= diameter_test_pair<6, 1>(w, thrv) & 0x00800080u;
            cmask &= valid;
        }
        // ---- 2. compact the wave's candidates into its own list segment (order is irrelevant; no workgroup barrier needed: a wave's LDS
        //         operations complete in order, and the wave is the only reader of its segment)
        const int n_mine = __popc(cmask);
        uint32_t pos = n_mine ? atomicAdd(&wave_cnt[wv], (uint32_t)n_mine) : 0u;
        uint16_t* const my_list = clist[wv];
        while (cmask) {
            const int b = __ffs(cmask) - 1;
            cmask &= cmask - 1;
            const int j = b & 3, rowsel = (b >> 2) & 1, hi = b >> 4;
            my_list[pos++] = (uint16_t)(((row0 + rowsel) << 8) | (c0 + 2 * j + hi));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int n_cand = (int)*reinterpret_cast<volatile uint32_t*>(&wave_cnt[wv]);
        // ---- 3. exact S for the wave's candidates, one per lane, into the shared score map
        for (int i = lane; i < n_cand; i += 64) {
            const uint32_t e = my_list[i];
            const int x = e & 255, y = e >> 8;
            const uint32_t sc = fast_strength_one(tbytes + y * (kTileWords * 4) + x + 3);
            sbytes[(y + 1) * (kSmapWords * 4) + 4 + x] = (uint8_t)sc;
        }
        __syncthreads();
        // ---- 4. strict NMS over the 8 neighbours (unevaluated neighbours have S <= thr < S(p): 0 in the map), survivors -> olist
        int any = 0;
        for (int i = lane; i < n_cand; i += 64) {
            const uint32_t e = my_list[i];
            const int x = e & 255, y = e >> 8;
            const uint8_t* q = sbytes + (y + 1) * (kSmapWords * 4) + 4 + x;
            const uint32_t sc = q[0];
            if ((int)sc <= thr) continue;
            constexpr int kS = kSmapWords * 4;
            const uint32_t nb = mx16(mx16(mx16((uint32_t)q[-kS - 1], (uint32_t)q[-kS]), mx16((uint32_t)q[-kS + 1], (uint32_t)q[-1])),
                                     mx16(mx16((uint32_t)q[1], (uint32_t)q[kS - 1]), mx16((uint32_t)q[kS], (uint32_t)q[kS + 1])));
            if (sc <= nb) continue;
            any = 1;
            if (fmask) {   // upstream drops masked keypoints after the empty-cell decision
                const uint32_t gx = min_x + 3 + x, gy = min_y + 3 + y;
                if (fmask[(size_t)(unsigned)(gy * scale) * stride0 + (unsigned)(gx * scale)] == 0) continue;
            }
            const uint32_t o = atomicAdd(&n_out, 1u);
            olist[o] = (sc << 16) | e;
        }
        if (__syncthreads_or(any) || thr <= geo->min_thr) break;
        // "if keypts_in_cell.empty()": again with min_fast_thr (rare: flat cells)
        thr = geo->min_thr;
        for (int i = tid; i < kSmapRows * kSmapWords / 4; i += 256) reinterpret_cast<uint4*>(smap_flat)[i] = uint4{0u, 0u, 0u, 0u};
        if (tid < 4) wave_cnt[tid] = 0;
        __syncthreads();
    }

    // ---- 5. append to the (frame, level) candidate list: one global atomic per workgroup; FAST response = S - 1
    const uint32_t total = n_out;
    if (total == 0) return;
    if (tid == 0) list_base = atomicAdd(&cand_count[frame * L + level], total);
    __syncthreads();
    const uint32_t base = list_base;
    uint64_t* const list = cand + (size_t)frame * cand_frame_entries + g.cand_off;
    const uint32_t cap = (uint32_t)g.cand_cap;
    for (uint32_t i = tid; i < total; i += 256) {
        const uint32_t o = olist[i];
        const uint32_t x = o & 255u, y = (o >> 8) & 255u, sc = o >> 16;
        if (base + i < cap) list[base + i] = cand_pack((uint32_t)(min_x + 3) + x, (uint32_t)(min_y + 3) + y, sc - 1u, 0);
    }
}

hipError_t launch_fast(const FrameGeo& hgeo, const DevBuffers& d, const uint8_t* img0, size_t stride0, size_t frame_stride0,
                       const uint8_t* mask, int mask_rows, int batch, hipStream_t s, int cell_lo, int n_cells) {
    (void)mask_rows;
    if (n_cells < 0) n_cells = hgeo.total_cells - cell_lo;
    if (n_cells <= 0 || batch <= 0) return hipSuccess;
    const unsigned per_xcd = (unsigned)(n_cells + 7) / 8u;
    hipLaunchKernelGGL(k_fast_cells, dim3(8u * per_xcd * (unsigned)batch), dim3(256), 0, s, d.geo, img0, stride0, frame_stride0, d.pyr,
                       d.pyr_frame_bytes, d.cand, d.cand_frame_entries, d.cand_count, mask, batch, cell_lo, n_cells);
    return hipGetLastError();
}

}   // namespace ovs
