// orb_fast.hip -- A2 + A3: orb_extractor::compute_fast_keypoints' cell loop and cv::FAST(TYPE_9_16, nonmax=true)
// (expected: src/openvslam/feature/orb_extractor.cc; OpenCV features2d fast.cpp / fast_score.cpp).
//
// Reformulation (proved in DESIGN.md, checked bit-for-bit against the oracle's literal OpenCV restatement):
//   S(p)      = max( max_arcs min_{9 contiguous}(v - ring), max_arcs min_{9 contiguous}(ring - v) )
//   corner(t) <=> S(p) > t,  cornerScore = S(p) - 1 (independent of t for detected corners),
//   NMS keeps p <=> S(p) > t and S(p) > S(q) for the 8 neighbours q inside the SAME cell's testable area
//   (neighbours with S(q) <= t score 0 upstream, but then S(q) <= t < S(p) anyway).
// So one threshold-free score plane per cell serves both thresholds: the cell uses ini_fast_thr if any NMS survivor
// exceeds it, else min_fast_thr ("if keypts_in_cell.empty() retry") -- no second pass over the pixels.
// The 64x64 testable areas of neighbouring cells tile the level exactly (cell stride 64, overlap 6 = two 3-px dead
// frames), so every pixel is scored once.
//
// S in 75 packed ops per PAIR of pixels (v1 used 176). With r[0..15] the raw ring values and j even:
//   the two 9-arcs {j-1..j+7} and {j..j+8} share the 8-window W_j = {j..j+7}, so
//   min_arcs max_9 r = min_j max( max W_j, min(r[j-1], r[j+8]) ),   max_arcs min_9 r = max_j min( min W_j, max(r[j-1], r[j+8]) )
//   S = max( c - min_arcs max_9 r,  max_arcs min_9 r - c ).
//   The eight even windows come from 8 pair + 8 quad min/max, the oct step folds into the arc step as a three-input
//   min/max (two pixels per register; 3-input packed min/max exists only in the f16 pipe -- see umin3/umax3 below).
//   Checked against the 16-arc definition on random rings (tests/test_oracle_kat.py).
// The kernel is VALU-issue bound (rocprofv3 PMC, profiles/r01_pmc_*.txt: SQ_ACTIVE_INST_VALU ~ 90 % of SIMD time, packed
// 16-bit integer ops issue at one wave-instruction per 4 cycles), so instructions per pixel are the only currency.
//
// Mapping: one 256-thread workgroup per cell; the <=70x70 u8 tile is staged in LDS with aligned 16-byte loads (cell
// origin x = 19+64j => tile origin 16+64j). Thread (run, rp) scores pixels [8*run, 8*run+8) of rows 2*rp and 2*rp+1
// from one 8x5-word register window (the two rows share 6 of their 7 window rows). Scores go to an LDS score map; NMS is
// evaluated on packed pairs as well (3x3 max via shared horizontal maxima); survivors are appended to the (frame, level)
// candidate list with ONE global atomic per workgroup. List order is irrelevant downstream (the quad-tree kernel uses
// counts and an explicit emission-order key).
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
// the integers and returned unflushed (checked exhaustively on gfx950 by tools/ubench/valu_rate.hip), and gfx950 has
// THREE-input packed f16 minimum / maximum (v_pk_minimum3_f16 / v_pk_maximum3_f16) at the issue rate of v_pk_max_i16,
// which the integer pipe lacks.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h16x2 as_h(s16x2 v) { return __builtin_bit_cast(h16x2, v); }
__device__ __forceinline__ s16x2 as_s(h16x2 v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ s16x2 umin2(s16x2 a, s16x2 b) { return as_s(__builtin_elementwise_minimum(as_h(a), as_h(b))); }
__device__ __forceinline__ s16x2 umax2(s16x2 a, s16x2 b) { return as_s(__builtin_elementwise_maximum(as_h(a), as_h(b))); }
__device__ __forceinline__ s16x2 umin3(s16x2 a, s16x2 b, s16x2 c) {
    return as_s(__builtin_elementwise_minimum(__builtin_elementwise_minimum(as_h(a), as_h(b)), as_h(c)));
}
__device__ __forceinline__ s16x2 umax3(s16x2 a, s16x2 b, s16x2 c) {
    return as_s(__builtin_elementwise_maximum(__builtin_elementwise_maximum(as_h(a), as_h(b)), as_h(c)));
}

// Threshold-free FAST-9/16 strength S (clamped at 0) for pixels P and P+1 (P even) of the run, window rows R0..R0+6.
template <int P, int R0>
__device__ __forceinline__ s16x2 fast_strength_pair(const uint32_t (&w)[8][5]) {
    const s16x2 c = window_pair<6 + P, R0 + 3>(w);
    s16x2 r[16];
    r[0] = window_pair<6 + P + 0, R0 + 6>(w);
    r[1] = window_pair<6 + P + 1, R0 + 6>(w);
    r[2] = window_pair<6 + P + 2, R0 + 5>(w);
    r[3] = window_pair<6 + P + 3, R0 + 4>(w);
    r[4] = window_pair<6 + P + 3, R0 + 3>(w);
    r[5] = window_pair<6 + P + 3, R0 + 2>(w);
    r[6] = window_pair<6 + P + 2, R0 + 1>(w);
    r[7] = window_pair<6 + P + 1, R0 + 0>(w);
    r[8] = window_pair<6 + P + 0, R0 + 0>(w);
    r[9] = window_pair<6 + P - 1, R0 + 0>(w);
    r[10] = window_pair<6 + P - 2, R0 + 1>(w);
    r[11] = window_pair<6 + P - 3, R0 + 2>(w);
    r[12] = window_pair<6 + P - 3, R0 + 3>(w);
    r[13] = window_pair<6 + P - 3, R0 + 4>(w);
    r[14] = window_pair<6 + P - 2, R0 + 5>(w);
    r[15] = window_pair<6 + P - 1, R0 + 6>(w);
    s16x2 pmx[8], pmn[8], qmx[8], qmn[8], ta[8], tb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        pmx[i] = umax2(r[2 * i], r[2 * i + 1]);
        pmn[i] = umin2(r[2 * i], r[2 * i + 1]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        qmx[i] = umax2(pmx[i], pmx[(i + 1) & 7]);   // max r[2i .. 2i+3]
        qmn[i] = umin2(pmn[i], pmn[(i + 1) & 7]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const s16x2 ea = r[(2 * i + 15) & 15], eb = r[(2 * i + 8) & 15];
        ta[i] = umax3(qmx[i], qmx[(i + 2) & 7], umin2(ea, eb));   // the cheaper-to-beat of the two 9-arcs around window 2i..2i+7
        tb[i] = umin3(qmn[i], qmn[(i + 2) & 7], umax2(ea, eb));
    }
    const s16x2 min_a = umin2(umin3(ta[0], ta[1], ta[2]), umin3(umin3(ta[3], ta[4], ta[5]), ta[6], ta[7]));   // min over arcs of max(ring)
    const s16x2 max_b = umax2(umax3(tb[0], tb[1], tb[2]), umax3(umax3(tb[3], tb[4], tb[5]), tb[6], tb[7]));   // max over arcs of min(ring)
    const s16x2 zero = {0, 0};
    return vmax(vmax(c - min_a, max_b - c), zero);
}

// bytes (B, B+1) of the 16-byte row {w0..w3} as a packed pair
template <int B>
__device__ __forceinline__ s16x2 row_pair(const uint32_t (&w)[4]) {
    return as_s16x2(pick2<(B & 3)>(w[(B >> 2) + ((B & 3) == 3 ? 1 : 0)], w[B >> 2]));
}

// Horizontal neighbourhood maxima of one score-map row around the run's 8 pixels (row bytes: pixel p at byte 4+p).
// h2[i] = max(s[x-1], s[x+1]), h3[i] = max(h2, s[x]) for the pixel pair x = 2i, 2i+1.
__device__ __forceinline__ void row_hmax(const uint32_t (&w)[4], s16x2 (&h2)[4], s16x2 (&h3)[4]) {
    const s16x2 l0 = row_pair<3>(w), l1 = row_pair<5>(w), l2 = row_pair<7>(w), l3 = row_pair<9>(w), l4 = row_pair<11>(w);
    h2[0] = umax2(l0, l1);
    h2[1] = umax2(l1, l2);
    h2[2] = umax2(l2, l3);
    h2[3] = umax2(l3, l4);
    h3[0] = umax3(l0, l1, row_pair<4>(w));
    h3[1] = umax3(l1, l2, row_pair<6>(w));
    h3[2] = umax3(l2, l3, row_pair<8>(w));
    h3[3] = umax3(l3, l4, row_pair<10>(w));
}

__global__ __launch_bounds__(256) void k_fast_cells(const FrameGeo* __restrict__ geo, const uint8_t* __restrict__ img0, size_t stride0,
                                                   size_t frame_stride0, const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                   uint64_t* __restrict__ cand, size_t cand_frame_entries,
                                                   uint32_t* __restrict__ cand_count, const uint8_t* __restrict__ mask,
                                                   int mask_rows) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[kTileRowsMax][kTileWords];
    __shared__ uint32_t smap[kSmapRows][kSmapWords];
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t list_base;

    const int tid = threadIdx.x;
    const int frame = blockIdx.y;
    const int L = geo->num_levels;
    // XCD-aware cell order: workgroup b runs on XCD b % 8 (each XCD has its own L2), so XCD k takes the k-th CONTIGUOUS eighth of the
    // frame's cells: neighbouring cells, which share tile halo lines, then hit the same L2 instead of fetching the line once per XCD
    // (fabric traffic 3.0x -> see profiles/). The grid is padded to a multiple of 8; surplus workgroups exit.
    const int per_xcd = gridDim.x >> 3;
    const int cell_id = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (cell_id >= geo->total_cells) return;
    int level = 0;
    for (int l = 1; l < L; ++l)
        if (cell_id >= geo->lv[l].cell_base) level = l;
    const LevelGeo& g = geo->lv[level];
    const int cell = cell_id - g.cell_base;
    const int ci = (int)(((float)cell + 0.5f) * g.inv_ncx), cj = cell - ci * g.ncx;   // exact: cell < 2^22 (see orb_pyramid.hip)

    const int min_x = kOrbPatchRadius + cj * kCellSize, min_y = kOrbPatchRadius + ci * kCellSize;
    int max_x = min_x + kCellSize + kCellOverlap, max_y = min_y + kCellSize + kCellOverlap;
    if (g.max_bx < max_x) max_x = g.max_bx;
    if (g.max_by < max_y) max_y = g.max_by;
    const int cw = max_x - min_x, ch = max_y - min_y;
    const int iw = cw - 6, ih = ch - 6;   // testable area of this cell (> 0 for every valid cell)

    const uint8_t* img;
    int pitch;
    if (level == 0) {
        img = img0 + (size_t)frame * frame_stride0;
        pitch = (int)stride0;
    } else {
        img = pyr + (size_t)frame * pyr_frame_bytes + g.plane_off;
        pitch = g.pitch;
    }
    const float scale = g.scale;
    const uint8_t* fmask = mask ? mask + (size_t)frame * frame_stride0 : nullptr;   // same layout as the level-0 frames
    (void)mask_rows;
    // upstream: skip the cell if one of its corners is masked (mask is indexed in level-0 coordinates, float scale, trunc)
    if (fmask) {
        auto in_mask = [&](unsigned y, unsigned x) {
            return fmask[(size_t)(unsigned)(y * scale) * stride0 + (unsigned)(x * scale)] == 0;
        };
        if (in_mask(min_y, min_x) || in_mask(max_y, min_x) || in_mask(min_y, max_x) || in_mask(max_y, max_x)) return;
    }

    // ---- stage the tile: tile byte u of row r <-> image (min_x - 3 + u, min_y + r); min_x - 3 = 16 + 64*cj
    const int ax0 = min_x - 3;
    const bool vec16 = ((pitch & 15) == 0) && ((reinterpret_cast<uintptr_t>(img) & 15) == 0);
    for (int idx = tid; idx < kTileRowsMax * 5; idx += 256) {
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
        *reinterpret_cast<uint4*>(&tile[r][4 * q]) = v;
    }
    // zero frame of the score map (interior is fully written below)
    if (tid < kSmapWords) {
        smap[0][tid] = 0;
        smap[kSmapRows - 1][tid] = 0;
    } else if (tid >= 64 && tid < 128) {
        smap[tid - 63][0] = 0;
        smap[tid - 63][kSmapWords - 1] = 0;
    }
    __syncthreads();

    const int run = tid & 7, rp = tid >> 3;
    const int c0 = run * 8, row0 = 2 * rp;
    // ---- scores of 2 rows x 8 pixels, packed pairs: sa[i] = row0 pixels (2i, 2i+1), sb[i] = row0+1
    s16x2 sa[4], sb[4];
    {
        uint32_t w[8][5];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint2 a = *reinterpret_cast<const uint2*>(&tile[row0 + r][2 * run]);
            const uint2 b = *reinterpret_cast<const uint2*>(&tile[row0 + r][2 * run + 2]);
            w[r][0] = a.x;
            w[r][1] = a.y;
            w[r][2] = b.x;
            w[r][3] = b.y;
            w[r][4] = tile[row0 + r][2 * run + 4];
        }
        sa[0] = fast_strength_pair<0, 0>(w);
        sa[1] = fast_strength_pair<2, 0>(w);
        sa[2] = fast_strength_pair<4, 0>(w);
        sa[3] = fast_strength_pair<6, 0>(w);
        sb[0] = fast_strength_pair<0, 1>(w);
        sb[1] = fast_strength_pair<2, 1>(w);
        sb[2] = fast_strength_pair<4, 1>(w);
        sb[3] = fast_strength_pair<6, 1>(w);
    }
    // outside the testable area -> 0 (tile bytes there are padding or belong to the next cell)
    {
        const uint32_t ra = (row0 < ih) ? 0xFFFFFFFFu : 0u, rb = (row0 + 1 < ih) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = c0 + 2 * i;
            const uint32_t m = (x + 1 < iw) ? 0xFFFFFFFFu : ((x < iw) ? 0x0000FFFFu : 0u);
            sa[i] = as_s16x2(as_u32(sa[i]) & m & ra);
            sb[i] = as_s16x2(as_u32(sb[i]) & m & rb);
        }
    }
    // score map rows (pixel row + 1), pixel p of the run at byte 4 + c0 + p
    {
        constexpr uint32_t pack = 0x06040200u;   // bytes 0,2 of src1 then bytes 0,2 of src0
        smap[row0 + 1][1 + 2 * run] = __builtin_amdgcn_perm(as_u32(sa[1]), as_u32(sa[0]), pack);
        smap[row0 + 1][2 + 2 * run] = __builtin_amdgcn_perm(as_u32(sa[3]), as_u32(sa[2]), pack);
        smap[row0 + 2][1 + 2 * run] = __builtin_amdgcn_perm(as_u32(sb[1]), as_u32(sb[0]), pack);
        smap[row0 + 2][2 + 2 * run] = __builtin_amdgcn_perm(as_u32(sb[3]), as_u32(sb[2]), pack);
    }
    __syncthreads();

    // ---- NMS (strict, 8 neighbours inside the cell): rows U = row0-1, A = row0, B = row0+1, D = row0+2
    int above_ini = 0;
    {
        uint32_t wu[4], wa[4], wb[4], wd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wu[i] = smap[row0 + 0][2 * run + i];
            wa[i] = smap[row0 + 1][2 * run + i];
            wb[i] = smap[row0 + 2][2 * run + i];
            wd[i] = smap[row0 + 3][2 * run + i];
        }
        s16x2 h2u[4], h3u[4], h2a[4], h3a[4], h2b[4], h3b[4], h2d[4], h3d[4];
        row_hmax(wu, h2u, h3u);
        row_hmax(wa, h2a, h3a);
        row_hmax(wb, h2b, h3b);
        row_hmax(wd, h2d, h3d);
        s16x2 top = {0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const s16x2 na = umax3(h3u[i], h2a[i], h3b[i]);
            const s16x2 nb = umax3(h3a[i], h2b[i], h3d[i]);
            // keep s where s > neighbours: (n - s) < 0  -> arithmetic shift gives an all-ones half
            sa[i] = sa[i] & ((na - sa[i]) >> 15);
            sb[i] = sb[i] & ((nb - sb[i]) >> 15);
            top = umax3(top, sa[i], sb[i]);
        }
        above_ini = (top.x > geo->ini_thr) | (top.y > geo->ini_thr);
    }
    const int thr = __syncthreads_or(above_ini) ? geo->ini_thr : geo->min_thr;

    // ---- emit: FAST response = S - 1; optional per-keypoint mask test. emit bit p <-> pixel (row0 + (p >> 3), c0 + (p & 7)).
    uint32_t emit = 0;
    {
        const s16x2 thrv = {(short)thr, (short)thr};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t ua = (as_u32(thrv - sa[i]) >> 15) & 0x00010001u;   // sign of (thr - s): s > thr
            const uint32_t ub = (as_u32(thrv - sb[i]) >> 15) & 0x00010001u;
            emit |= ((ua | (ua >> 15)) & 3u) << (2 * i);
            emit |= ((ub | (ub >> 15)) & 3u) << (8 + 2 * i);
        }
    }
    if (fmask && emit) {
        uint32_t m = emit;
        while (m) {
            const int p = __ffs(m) - 1;
            m &= m - 1;
            const uint32_t x = min_x + 3 + c0 + (p & 7), y = min_y + 3 + row0 + (p >> 3);
            if (fmask[(size_t)(unsigned)(y * scale) * stride0 + (unsigned)(x * scale)] == 0) emit &= ~(1u << p);
        }
    }
    const int n_out = __popc(emit);
    // block-level exclusive scan of n_out
    const int lane = tid & 63, wv = tid >> 6;
    int incl = n_out;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    if (tid == 0) {
        const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        list_base = total ? atomicAdd(&cand_count[frame * L + level], total) : 0u;
    }
    __syncthreads();
    uint32_t pos = list_base + (incl - n_out);
    for (int w2 = 0; w2 < wv; ++w2) pos += wave_tot[w2];
    uint64_t* list = cand + (size_t)frame * cand_frame_entries + g.cand_off;
    const uint32_t cap = (uint32_t)g.cand_cap;
    // survivors are rare (<= 1 % of the pixels): walk the set bits; an NMS survivor's score is still in the score map
    const uint8_t* sbytes = reinterpret_cast<const uint8_t*>(&smap[0][0]);
    while (emit) {
        const int p = __ffs(emit) - 1;
        emit &= emit - 1;
        const int px = c0 + (p & 7), py = row0 + (p >> 3);
        const uint32_t v = sbytes[(py + 1) * (kSmapWords * 4) + 4 + px];
        if (pos < cap) list[pos] = cand_pack((uint32_t)(min_x + 3 + px), (uint32_t)(min_y + 3 + py), v - 1u, 0);
        ++pos;
    }
}

hipError_t launch_fast(const FrameGeo& hgeo, const DevBuffers& d, const uint8_t* img0, size_t stride0, size_t frame_stride0,
                       const uint8_t* mask, int mask_rows, int batch, hipStream_t s) {
    if (hgeo.total_cells == 0) return hipSuccess;
    dim3 grid(((hgeo.total_cells + 7) / 8) * 8, batch);
    hipLaunchKernelGGL(k_fast_cells, grid, dim3(256), 0, s, d.geo, img0, stride0, frame_stride0, d.pyr, d.pyr_frame_bytes, d.cand,
                       d.cand_frame_entries, d.cand_count, mask, mask_rows);
    return hipGetLastError();
}

}   // namespace ovs
