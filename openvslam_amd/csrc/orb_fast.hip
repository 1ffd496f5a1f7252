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
// Mapping: one 256-thread workgroup per cell; the <=70x70 u8 tile is staged in LDS with aligned u32 loads (cell origin
// is 19+64j, i.e. always 3 bytes past a 4-byte boundary of the 256-byte-pitched plane); each thread scores two runs of
// 8 pixels with 2-pixels-per-register packed 16-bit min/max (v_pk_min_i16 / v_pk_max_i16) from a 7x5-word register
// window; survivors are appended to the (frame, level) candidate list with ONE global atomic per workgroup. List order
// is irrelevant downstream (the quad-tree kernel uses counts and an explicit emission-order key).
#include "ovs_common.h"

namespace ovs {

typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int kTileRowsMax = kCellSize + kCellOverlap;   // 70
constexpr int kTileWords = 20;                           // 80-byte LDS pitch (19 words used)
constexpr int kSmapPitch = 72;                           // 4 + 64 + 4
constexpr int kSmapRows = 66;                            // 1 + 64 + 1

__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ s16x2 vmin(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 vmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }

// bytes q and q+1 of the 8-byte window {hi,lo} zero-extended into the two 16-bit halves
template <int M>
__device__ __forceinline__ uint32_t pick2(uint32_t hi, uint32_t lo) {
    constexpr uint32_t sel = (uint32_t)M | (0x0cu << 8) | ((uint32_t)(M + 1) << 16) | (0x0cu << 24);
    return __builtin_amdgcn_perm(hi, lo, sel);
}

template <int Q, int R>
__device__ __forceinline__ s16x2 window_pair(const uint32_t (&w)[7][5]) {
    return as_s16x2(pick2<(Q & 3)>(w[R][(Q >> 2) + ((Q & 3) == 3 ? 1 : 0)], w[R][Q >> 2]));
}

// Threshold-free FAST-9/16 strength S for pixels P and P+1 of the run (P even).
template <int P>
__device__ __forceinline__ s16x2 fast_strength_pair(const uint32_t (&w)[7][5]) {
    const s16x2 c = window_pair<6 + P, 3>(w);
    s16x2 d[16];
    d[0] = c - window_pair<6 + P + 0, 6>(w);
    d[1] = c - window_pair<6 + P + 1, 6>(w);
    d[2] = c - window_pair<6 + P + 2, 5>(w);
    d[3] = c - window_pair<6 + P + 3, 4>(w);
    d[4] = c - window_pair<6 + P + 3, 3>(w);
    d[5] = c - window_pair<6 + P + 3, 2>(w);
    d[6] = c - window_pair<6 + P + 2, 1>(w);
    d[7] = c - window_pair<6 + P + 1, 0>(w);
    d[8] = c - window_pair<6 + P + 0, 0>(w);
    d[9] = c - window_pair<6 + P - 1, 0>(w);
    d[10] = c - window_pair<6 + P - 2, 1>(w);
    d[11] = c - window_pair<6 + P - 3, 2>(w);
    d[12] = c - window_pair<6 + P - 3, 3>(w);
    d[13] = c - window_pair<6 + P - 3, 4>(w);
    d[14] = c - window_pair<6 + P - 2, 5>(w);
    d[15] = c - window_pair<6 + P - 1, 6>(w);
    s16x2 lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo2[k] = vmin(d[k], d[(k + 1) & 15]);
        hi2[k] = vmax(d[k], d[(k + 1) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo4[k] = vmin(lo2[k], lo2[(k + 2) & 15]);
        hi4[k] = vmax(hi2[k], hi2[(k + 2) & 15]);
    }
    s16x2 a = {-32768, -32768};   // max over arcs of min(d)  (centre brighter than the arc)
    s16x2 b = {32767, 32767};     // min over arcs of max(d)  (centre darker than the arc): B = -b
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const s16x2 lo9 = vmin(vmin(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
        const s16x2 hi9 = vmax(vmax(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
        a = vmax(a, lo9);
        b = vmin(b, hi9);
    }
    const s16x2 zero = {0, 0};
    return vmax(a, zero - b);
}

struct CellInfo {
    int level, ci, cj;
};

__global__ __launch_bounds__(256) void k_fast_cells(const FrameGeo* __restrict__ geo, const uint8_t* __restrict__ img0, size_t stride0,
                                                   size_t frame_stride0, const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                   uint64_t* __restrict__ cand, size_t cand_frame_entries,
                                                   uint32_t* __restrict__ cand_count, const uint8_t* __restrict__ mask,
                                                   int mask_rows) {
    __shared__ uint32_t tile[kTileRowsMax][kTileWords];
    __shared__ uint32_t smap[kSmapRows][kSmapPitch / 4];
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t list_base;

    const int tid = threadIdx.x;
    const int frame = blockIdx.y;
    const int L = geo->num_levels;
    int level = 0;
    for (int l = 1; l < L; ++l)
        if ((int)blockIdx.x >= geo->lv[l].cell_base) level = l;
    const LevelGeo& g = geo->lv[level];
    const int cell = blockIdx.x - g.cell_base;
    const int ci = cell / g.ncx, cj = cell - ci * g.ncx;

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

    // ---- stage the tile: tile byte u of row r <-> image (min_x - 3 + u, min_y + r); min_x - 3 is 4-byte aligned
    const int ax0 = min_x - 3;
    for (int idx = tid; idx < kTileRowsMax * 19; idx += 256) {
        const int r = idx / 19, wi = idx - r * 19;
        uint32_t v = 0;
        const int gx = ax0 + 4 * wi;
        if (r < ch && gx + 4 <= pitch) v = *reinterpret_cast<const uint32_t*>(img + (size_t)(min_y + r) * pitch + gx);
        tile[r][wi] = v;
    }
    for (int idx = tid; idx < kSmapRows * (kSmapPitch / 4); idx += 256) (&smap[0][0])[idx] = 0;
    __syncthreads();

    const int tlo = geo->ini_thr < geo->min_thr ? geo->ini_thr : geo->min_thr;
    const int run = tid & 7;
    const int c0 = run * 8;
    // ---- score plane
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int row = (tid >> 3) + half * 32;
        if (row < ih && c0 < iw) {
            uint32_t w[7][5];
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int i = 0; i < 5; ++i) w[r][i] = (2 * run + i < kTileWords) ? tile[row + r][2 * run + i] : 0;
            const s16x2 s01 = fast_strength_pair<0>(w);
            const s16x2 s23 = fast_strength_pair<2>(w);
            const s16x2 s45 = fast_strength_pair<4>(w);
            const s16x2 s67 = fast_strength_pair<6>(w);
            int s[8] = {s01.x, s01.y, s23.x, s23.y, s45.x, s45.y, s67.x, s67.y};
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int v = (s[p] > tlo && c0 + p < iw) ? s[p] : 0;
                if (p < 4) lo |= (uint32_t)v << (8 * p);
                else hi |= (uint32_t)v << (8 * (p - 4));
            }
            smap[row + 1][1 + 2 * run] = lo;
            smap[row + 1][2 + 2 * run] = hi;
        }
    }
    __syncthreads();

    // ---- NMS (strict, 8 neighbours inside the cell) and the per-cell threshold rule
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(&smap[0][0]);
    uint32_t keep[2] = {0, 0};   // bit p: pixel p of the run survives NMS
    int above_ini = 0;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int row = (tid >> 3) + half * 32;
        const uint8_t* c = sb + (row + 1) * kSmapPitch + 4 + c0;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int v = c[p];
            if (v == 0) continue;
            const bool k = v > c[p - 1] && v > c[p + 1] && v > c[p - kSmapPitch - 1] && v > c[p - kSmapPitch] &&
                           v > c[p - kSmapPitch + 1] && v > c[p + kSmapPitch - 1] && v > c[p + kSmapPitch] &&
                           v > c[p + kSmapPitch + 1];
            if (k) {
                keep[half] |= 1u << p;
                above_ini |= (v > geo->ini_thr);
            }
        }
    }
    const int thr = __syncthreads_or(above_ini) ? geo->ini_thr : geo->min_thr;

    // ---- emit: FAST response = S - 1; optional per-keypoint mask test. Two sweeps over the survivor bits (count, then
    // write) keep everything in registers.
    auto emit_ok = [&](int v, uint32_t x, uint32_t y) -> bool {
        if (v <= thr) return false;
        if (fmask && fmask[(size_t)(unsigned)(y * scale) * stride0 + (unsigned)(x * scale)] == 0) return false;
        return true;
    };
    int n_out = 0;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int row = (tid >> 3) + half * 32;
        const uint8_t* c = sb + (row + 1) * kSmapPitch + 4 + c0;
        uint32_t m = keep[half];
        while (m) {
            const int p = __ffs(m) - 1;
            m &= m - 1;
            if (emit_ok(c[p], min_x + 3 + c0 + p, min_y + 3 + row)) ++n_out;
        }
    }
    // block-level exclusive scan of n_out (<= 8: NMS leaves at most 4 survivors per 8-pixel run)
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
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int row = (tid >> 3) + half * 32;
        const uint8_t* c = sb + (row + 1) * kSmapPitch + 4 + c0;
        uint32_t m = keep[half];
        while (m) {
            const int p = __ffs(m) - 1;
            m &= m - 1;
            const int v = c[p];
            const uint32_t x = min_x + 3 + c0 + p, y = min_y + 3 + row;
            if (!emit_ok(v, x, y)) continue;
            if (pos < (uint32_t)g.cand_cap) list[pos] = cand_pack(x, y, (uint32_t)(v - 1), 0);
            ++pos;
        }
    }
}

hipError_t launch_fast(const FrameGeo& hgeo, const DevBuffers& d, const uint8_t* img0, size_t stride0, size_t frame_stride0,
                       const uint8_t* mask, int mask_rows, int batch, hipStream_t s) {
    if (hgeo.total_cells == 0) return hipSuccess;
    dim3 grid(hgeo.total_cells, batch);
    hipLaunchKernelGGL(k_fast_cells, grid, dim3(256), 0, s, d.geo, img0, stride0, frame_stride0, d.pyr, d.pyr_frame_bytes, d.cand,
                       d.cand_frame_entries, d.cand_count, mask, mask_rows);
    return hipGetLastError();
}

}   // namespace ovs
