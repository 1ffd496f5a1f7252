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

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace ovs {

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

constexpr int kTileRowsMax = kCellSize + kCellOverlap;   // 70
constexpr int kTileWords = 20;                           // 80-byte LDS pitch: five 16-byte chunks per row, chunk i of the tile at byte 16 * i
constexpr int kChunks = kTileRowsMax * 5;                // 350: one per thread + a second one for threads 0..93
constexpr int kSmapWords = 18;                           // 72-byte pitch: 4 + 64 + 4
constexpr int kSmapRows = 66;                            // 1 + 64 + 1
constexpr int kMaxSurvivors = 1024;   // NMS survivors are pairwise non-adjacent: at most 32 x 32 per 64 x 64 cell

// ================================================================================================================================
// v4 (round 3): the same algorithm, restructured after measuring where a cell's time goes (tools/fast_phases.py, OVS_FAST_TIMING):
//   * pre-test on FOUR pixels per register in the fast-class 32-bit ops (v_sub / v_and / v_or, 2.3-2.5 cycles per wave-instruction)
//     instead of two per register in packed 16-bit ops (4.2): the tile is staged a second time as R = (p >> 2) | 0x80 per byte, and with
//     th6 = (t + 1) >> 2 the byte-wise differences
//         X_i = R_i - (C - K),  Y_i = (C + K) - R_i      (K = 0x80808080 - th6 * 0x01010101, C = centre bytes, R_i = ring bytes)
//     never borrow across bytes (every byte stays in [1, 191]) and carry in bit 7 "r6 >= c6 + th6" resp. "r6 <= c6 - th6". Since
//     r - c > t implies (r >> 2) - (c >> 2) >= (t + 1) >> 2, the four-even-diameter condition on these bits is still NECESSARY for
//     S > t; it lets ~1.3x as many pixels through as the exact 8-bit form (tools/fast_pretest_model.py), all of which get the exact S.
//     Ring positions 2 / 6 / 10 / 14 are word-aligned for a 4-pixel group: a group costs 5 v_alignbyte + ~30 fast ops (v3: 104 slow ops);
//   * the pre-test reports which polarity passed, and the exact scoring evaluates only that one (values complemented for "dark"): 17 + 50
//     instead of 102 min / max per candidate;
//   * survivors of the four waves are pooled in ONE list (wave prefix sum by DPP, one LDS atomic per wave);
//   * a workgroup takes six CONSECUTIVE cells: the next cell's tile is requested a whole cell ahead into registers, barriers order LDS
//     only (no vmcnt drain), a cell's geometry is ONE 8-byte record (CellDesc) instead of a chain of dependent scalar loads, and the
//     NMS survivors of the group's cells are buffered in LDS and appended with ONE global reservation per group instead of one atomic
//     round trip per cell (that round trip alone was a fifth of a cell's latency).
// What the measurements say (settled in round 5 by a mixed-stream microbenchmark, tools/ubench/valu_rate mix,
// profiles/r05d_valu_mix.txt): at this kernel's 7 waves per SIMD the pre-test's real opcode mix issues at 3.71 cycles per wave-instruction, the
// scorer's 16-bit min / max mix at 2.69 -- weighted by the kernel's mix 3.3, so its 8.76e8 VALU wave-instructions per 256 frames keep the vector
// ALU ~63 % busy (the "4 cycles per instruction, 0.76-0.78 of the issue slots" of rounds 3-4 read the SQ counters' quad-cycle quantisation as a
// busy measure). The kernel is bound by neither the vector ALU nor LDS (~47 % busy, a third of it bank conflicts) nor HBM (0.9-1.2 TB/s) alone but
// by the latency chain of a cell -- six LDS-only barriers, dependent LDS gathers of the exact scoring / NMS with 3-4x bank conflicts -- at the 28
// waves per CU that 21 KB of LDS leave. Halving the pre-test's cycles, an eight-diameter pre-test (-22 %
// candidates), 8 workgroups per CU (aliased LDS, 64 VGPRs) each moved the launch by < 3 %; the grouping + prefetch + single reservation
// are worth ~6 % together (0.50 -> 0.47 ms per 64 frames).
// LDS: raw tile 5.6 KB + R tile 5.6 KB + score map 4.75 KB + pooled list 4 KB + group buffer 1 KB = 21.1 KB -> 7 workgroups per CU (until late in
// round 3: an 8 KB list for all 4096 pixels of a cell and a 2 KB buffer, 26.2 KB, six workgroups; see kListCap).
__device__ __forceinline__ uint32_t wave_prefix_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return v;
}

__device__ __forceinline__ uint32_t to_r6(uint32_t w) { return __builtin_amdgcn_bitop3_b32(w >> 2, 0x3f3f3f3fu, 0x80808080u, 0xea); }   // (a & b) | c

// bit 7 of byte b <-> pixel 4g + b of the group passes the four-diameter test. w = the thread's 8 x 5-word window of the R tile,
// G = group (0 / 1) inside the thread's 8-pixel run, R0 = row (0 / 1) of its row pair; kk = K above.
// v_bitop3_b32: a & (b | c)
__device__ __forceinline__ uint32_t and_or3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xe0); }

template <int G, int R0>
__device__ __forceinline__ void swar_diameter_test(const uint32_t (&w)[8][5], uint32_t kk, uint32_t& bright, uint32_t& dark) {
    const uint32_t c = __builtin_amdgcn_alignbyte(w[R0 + 3][G + 2], w[R0 + 3][G + 1], 2);
    const uint32_t p0 = __builtin_amdgcn_alignbyte(w[R0 + 6][G + 2], w[R0 + 6][G + 1], 2);    // (0, +3)
    const uint32_t p8 = __builtin_amdgcn_alignbyte(w[R0 + 0][G + 2], w[R0 + 0][G + 1], 2);    // (0, -3)
    const uint32_t p4 = __builtin_amdgcn_alignbyte(w[R0 + 3][G + 3], w[R0 + 3][G + 2], 1);    // (+3, 0)
    const uint32_t p12 = __builtin_amdgcn_alignbyte(w[R0 + 3][G + 1], w[R0 + 3][G + 0], 3);   // (-3, 0)
    const uint32_t p2 = w[R0 + 5][G + 2], p14 = w[R0 + 5][G + 1];                             // (+2, +2), (-2, +2)
    const uint32_t p6 = w[R0 + 1][G + 2], p10 = w[R0 + 1][G + 1];                             // (+2, -2), (-2, -2)
    const uint32_t cb = c - kk, cd = c + kk;
    // bit 7 of every byte: all four diameters have a bright (dark) end
    bright = and_or3(and_or3(and_or3((p0 - cb) | (p8 - cb), p2 - cb, p10 - cb), p4 - cb, p12 - cb), p6 - cb, p14 - cb);
    dark = and_or3(and_or3(and_or3((cd - p0) | (cd - p8), cd - p2, cd - p10), cd - p4, cd - p12), cd - p6, cd - p14);
}

// S of one polarity: max over the sixteen 9-arcs of the arc's minimum ring value, minus the centre (clamped at 0). For the dark polarity
// the caller passes the complemented bytes (255 - v): max_arcs min_9 (c - r) = max_arcs min_9 ((255 - r) - (255 - c)).
__device__ __forceinline__ uint32_t fast_strength_bright(const uint32_t (&r)[16], uint32_t c) {
    uint32_t pmn[8], qmn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pmn[i] = mn16(r[2 * i], r[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 8; ++i) qmn[i] = mn16(pmn[i], pmn[(i + 1) & 7]);
    uint32_t best = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // the two 9-arcs {2i - 1 .. 2i + 7} and {2i .. 2i + 8} share the window {2i .. 2i + 7}
        const uint32_t ea = r[(2 * i + 15) & 15], eb = r[(2 * i + 8) & 15];
        best = mx16(best, mn16(mn16(qmn[i], qmn[(i + 2) & 7]), mx16(ea, eb)));
    }
    return best > c ? best - c : 0u;
}

// exact S of one candidate whose necessary test passed for ONE polarity (flip = 0: bright, 0xff: dark): the other polarity's arc value is
// <= thr, so max(bright, dark) = this polarity's value whenever it matters (S > thr), and the pixel is no corner otherwise.
// p = LDS address of the top-left byte of its 7x7 neighbourhood in the staged tile (pitch kTileWords * 4)
__device__ __forceinline__ void load_ring(const uint8_t* p, uint32_t (&r)[16], uint32_t& c) {
    constexpr int kP = kTileWords * 4;
    c = p[3 * kP + 3];
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
}

// workgroup barrier that orders LDS traffic only: __syncthreads() carries a workgroup-scope fence over ALL address spaces, i.e. an
// s_waitcnt vmcnt(0) that would drain the next cell's tile loads (in flight across the whole body of the loop below) at every barrier
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// level-dependent part of a cell's geometry (all wave-uniform; reloaded only when a group's cells cross a level boundary)
struct LevelRef {
    const uint8_t* img;   // plane of this frame's level
    int pitch, level;
    int64_t cand_off;
    int cand_cap;
    float scale;
    bool vec16;
};

constexpr int kMaxCellsPerWg = 64;
constexpr int kWgSurvivors = 256;   // NMS survivors of the group's cells waiting for their (single) list reservation: ~33 per cell on video
constexpr int kListCap = 2048;      // pooled candidate list of a cell; a denser cell (noise at a low threshold) is scored exhaustively instead

// ---- LDS-DMA (round 4) -------------------------------------------------------------------------------------------------------------
// global_load_lds_dwordx4: lane l of the wave copies 16 bytes from (uniform base + its 32-bit offset) to LDS byte address M0 + 16 * l, with
// no vector register in between; masked-off lanes copy nothing. The tile's layout (chunk i at byte 16 * i) is exactly that image, so a
// wave stages 1 KiB of the tile with ONE instruction whose per-lane operand (row * pitch + 16 * column chunk) changes only at a level
// boundary. hipcc neither knows M0 is live across the statement nor counts the copy: M0 is saved / restored inside the statement, and
// the copy is retired by the explicit s_waitcnt vmcnt(0) of wait_tile() (the issuing wave's own counter: a wave reads ITS chunks back
// right after that wait; other waves' chunks only behind the following barrier). Over-waiting is the only interaction with the waits
// hipcc emits for its own loads (vmcnt counts retire in issue order), never under-waiting.
__device__ __forceinline__ void glds16(const uint8_t* base, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_dst)
                 : "memory");
}
// 4-byte form (lane l -> M0 + 4 * l): planes whose base or pitch is only 4-byte aligned (a caller's level-0 image)
__device__ __forceinline__ void glds4(const uint8_t* base, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void wait_tile() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }

// LDS per workgroup: raw tile 5.6 KB + R tile 5.6 KB + score map 4.75 KB + pooled list 4 KB + group buffer 1 KB = 21.1 KB, seven workgroups
// per CU, 72 VGPRs. The next cell's tile is requested when the current cell's last tile read (the exact scoring; the NMS survivor list aliases
// the R tile) is behind every wave, and lands under the cell's tail and the other six workgroups. (A second raw-tile buffer, requested a whole
// cell ahead -- 26.7 KB, six workgroups per CU -- measured 5-10 % slower in round 4 and is gone.)
// kTiming: wave 0 accumulates shader cycles per phase into tstats (tuning aid, OVS_FAST_TIMING).
template <bool kTiming>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_fast_cells(
    const FrameGeo* __restrict__ geo, const CellDesc* __restrict__ cell_tab, const uint8_t* __restrict__ img0, size_t stride0, size_t frame_stride0,
    const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes, uint64_t* __restrict__ cand, size_t cand_frame_entries, uint32_t* __restrict__ cand_count,
    const uint8_t* __restrict__ mask, int batch, uint32_t batch_magic, int cell_lo, int n_cells, int cells_per_wg, int group_major, unsigned long long* __restrict__ tstats) {
    __shared__ __attribute__((aligned(16))) uint32_t tiles[1][kTileRowsMax][kTileWords];
    __shared__ __attribute__((aligned(16))) uint32_t smap[kSmapRows][kSmapWords];
    __shared__ __attribute__((aligned(16))) uint32_t rtile[kTileRowsMax][kTileWords];
    __shared__ uint16_t clist[kListCap];                // pixels that passed the diameter test, (y << 8) | x, all four waves
    __shared__ uint32_t wgbuf[kWgSurvivors];            // (slot << 26) | (score << 12) | (y << 6) | x of the group's survivors not yet written out
    __shared__ uint32_t n_cand_wg, n_out, list_base;

    static_assert(sizeof(rtile) >= kMaxSurvivors * sizeof(uint32_t), "survivor list aliases the R tile");
    uint32_t* const olist = &rtile[0][0];   // written by the NMS, when every wave is past its last R-tile read (the pre-test)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = geo->num_levels;
    unsigned long long t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
    const bool t_on = kTiming && wave == 0;
#define FAST_MARK(i)                                   \
    if (kTiming && t_on) {                             \
        const unsigned long long t_now = clock64();    \
        t_acc[i] += t_now - t_prev;                    \
        t_prev = t_now;                                \
    }
    if (kTiming && t_on) t_prev = clock64();
    // Work order: a workgroup takes `cells_per_wg` CONSECUTIVE cells of one frame (the workgroup lives long enough that its launch and the
    // first tile's load latency are paid once per group, not once per cell). XCD-aware: XCD k takes the k-th contiguous eighth of the
    // groups, the frame index runs fastest inside an XCD's share.
    const int n_groups = (n_cells + cells_per_wg - 1) / cells_per_wg;
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    int frame, group;
    if (group_major) {   // see k_fast_wave
        const int share = (n_groups * batch + 7) >> 3;
        const int gid = xcd * share + idx;
        if (idx >= share || gid >= n_groups * batch) return;
        frame = gid / n_groups;
        group = gid - frame * n_groups;
    } else {
        const int per_xcd = (n_groups + 7) >> 3;
        const int slot = batch == 1 ? idx : (int)__umulhi((uint32_t)idx, batch_magic);
        frame = idx - slot * batch;
        group = xcd * per_xcd + slot;
        if (slot >= per_xcd || group >= n_groups) return;
    }
    const int cell_first = cell_lo + group * cells_per_wg;
    const int n_here = min(cells_per_wg, cell_lo + n_cells - cell_first);

    // a cell's record: ONE 8-byte scalar load at a wave-uniform address (x | y << 16, cw | ch << 8 | level << 16)
    uint32_t n_buf = 0;   // survivors waiting in wgbuf: every thread keeps the same count (all control flow below is workgroup-uniform)
    const uint2 d0 = reinterpret_cast<const uint2*>(cell_tab)[cell_first];

    auto level_ref = [&](int level) -> LevelRef {
        const LevelGeo& g = geo->lv[level];
        LevelRef r;
        r.level = level;
        if (level == 0) {
            r.img = img0 + (size_t)frame * frame_stride0;
            r.pitch = (int)stride0;
        } else {
            r.img = pyr + (size_t)frame * pyr_frame_bytes + g.plane_off;
            r.pitch = g.pitch;
        }
        r.vec16 = ((r.pitch & 15) == 0) && ((reinterpret_cast<uintptr_t>(r.img) & 15) == 0);
        r.cand_off = g.cand_off;
        r.cand_cap = g.cand_cap;
        r.scale = g.scale;
        return r;
    };
    // this thread's chunks of a tile: chunk tid = (row ra, column chunk qa), chunk tid + 256 = (rb, qb) for the first 94 threads
    const int ra = (tid * 0x3334) >> 16, qa = tid - 5 * ra;                       // tid / 5, tid % 5
    const int rb = ((tid + 256) * 0x3334) >> 16, qb = (tid + 256) - 5 * rb;
    const bool has_b = tid + 256 < kChunks;
    uint32_t off_a = 0, off_b = 0;   // byte offsets of the two chunks from a tile's origin in the current level's plane
    auto chunk_offsets = [&](const LevelRef& lv) {
        off_a = (uint32_t)(ra * lv.pitch + 16 * qa);
        off_b = (uint32_t)(rb * lv.pitch + 16 * qb);
    };
    // tile byte u of row r <-> image (min_x - 3 + u, min_y + r); min_x - 3 = 16 + 64 * column is 16-byte aligned. Rows >= ch and chunks at or
    // beyond the plane's pitch are NOT copied: those tile bytes keep whatever an earlier cell left there, and only feed pixels outside the cell's
    // testable area, whose pre-test bits the `valid` mask removes (the exact scorer and the NMS only ever see valid pixels).
    auto issue_tile = [&](const LevelRef& lv, uint32_t rec_x, uint32_t rec_y, uint32_t lds_tile) {
        const int min_x = (int)(rec_x & 0xffffu), min_y = (int)(rec_x >> 16), ch = (int)((rec_y >> 8) & 255u);
        const int gx = min_x - 3;
        const uint8_t* const base = lv.img + (size_t)min_y * (size_t)lv.pitch + (size_t)gx;
        if (lv.vec16) {
            if (ra < ch && gx + 16 * qa < lv.pitch) glds16(base, off_a, lds_tile + 1024u * (uint32_t)wave);
            if (has_b && rb < ch && gx + 16 * qb < lv.pitch) glds16(base, off_b, lds_tile + 4096u + 1024u * (uint32_t)wave);
        } else {   // 4-byte aligned base / pitch (enforced by the ABI): 1400 words, 64 per wave-instruction
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                int lane = tid & 63;
                asm volatile("" : "+v"(lane));   // keeps the per-lane (row, word) of this rare path out of registers that live across the cell loop
                const int i = 64 * (wave + 4 * j) + lane;
                const int r = (i * 0x0ccd) >> 16, c4 = i - 20 * r;   // i / 20 for i < 1536
                if (i < kTileRowsMax * kTileWords && r < ch && gx + 4 * c4 + 4 <= lv.pitch)
                    glds4(base, (uint32_t)(r * lv.pitch + 4 * c4), lds_tile + 256u * (uint32_t)(wave + 4 * j));
            }
        }
    };
    const int run = tid & 7, rp = tid >> 3;
    const int c0 = run * 8, row0 = 2 * rp;
    uint32_t* const smap_flat = &smap[0][0];
    uint8_t* const sbytes = reinterpret_cast<uint8_t*>(smap_flat);
    static_assert((kSmapRows * kSmapWords) % 4 == 0, "score map is cleared with 16-byte stores");
    const uint8_t* const fmask = mask ? mask + (size_t)frame * frame_stride0 : nullptr;   // same layout as the level-0 frames
    const int ini_thr = geo->ini_thr, min_thr = geo->min_thr;
    const uint32_t lds_tile0 = lds_addr(&tiles[0][0][0]);

    uint32_t dn_x = d0.x, dn_y = d0.y;                          // record of the cell whose tile is in flight
    LevelRef lr_next = level_ref((int)((dn_y >> 16) & 255u));   // its level
    chunk_offsets(lr_next);
    issue_tile(lr_next, dn_x, dn_y, lds_tile0);
    LevelRef lr_buf = lr_next;   // level of the survivors waiting in wgbuf

    // write the buffered survivors of the group to their (frame, level) list: ONE reservation for all of them
    auto flush = [&](const LevelRef& lv) {
        const uint32_t nb = n_buf;
        if (nb != 0) {
            if (tid == 0) list_base = atomicAdd(&cand_count[frame * L + lv.level], nb);
            lds_barrier();   // thread 0 waits for the atomic's return before its LDS store; the barrier publishes it
            const uint32_t base = list_base;
            uint64_t* const list = cand + (size_t)frame * cand_frame_entries + lv.cand_off;
            const uint32_t cap = (uint32_t)lv.cand_cap;
            for (uint32_t i = tid; i < nb; i += 256) {
                const uint32_t e = wgbuf[i];
                const uint2 dsc = reinterpret_cast<const uint2*>(cell_tab)[cell_first + (int)(e >> 26)];
                const uint32_t x = (dsc.x & 0xffffu) + 3u + (e & 63u), y = (dsc.x >> 16) + 3u + ((e >> 6) & 63u), sc = (e >> 12) & 255u;
                if (base + i < cap) list[base + i] = cand_pack(x, y, sc - 1u, 0);
            }
            n_buf = 0;
        }
    };
    // request the tile of cell k + 1 (workgroup-uniform)
    auto request_next = [&](int k) {
        if (k + 1 < n_here) {
            const uint2 dsc = reinterpret_cast<const uint2*>(cell_tab)[cell_first + k + 1];   // uniform address: scalar load, not an LDS round trip
            dn_x = dsc.x;
            dn_y = dsc.y;
            const int nl = (int)((dn_y >> 16) & 255u);
            if (nl != lr_next.level) {
                lr_next = level_ref(nl);
                chunk_offsets(lr_next);
            }
            issue_tile(lr_next, dn_x, dn_y, lds_tile0);
        }
    };

    for (int k = 0; k < n_here; ++k) {
        FAST_MARK(0)   // loop overhead / previous cell's tail
        uint32_t(*const tile)[kTileWords] = tiles[0];
        const uint8_t* const tbytes = reinterpret_cast<const uint8_t*>(&tile[0][0]);
        const uint32_t dc_x = dn_x, dc_y = dn_y;   // this cell's record
        const LevelRef lr = lr_next;
        // ---- the tile has landed (this wave's chunks; with the 4-byte form a chunk is other waves' words: barrier first)
        wait_tile();
        if (!lr.vec16) lds_barrier();
        FAST_MARK(1)   // wait for the tile
        // R = (p >> 2) | 0x80 per byte for the diameter test: every thread converts the chunks its own lanes copied
        {
            const uint4 va = *reinterpret_cast<const uint4*>(&tile[ra][4 * qa]);
            *reinterpret_cast<uint4*>(&rtile[ra][4 * qa]) = uint4{to_r6(va.x), to_r6(va.y), to_r6(va.z), to_r6(va.w)};
            if (has_b) {
                const uint4 vb = *reinterpret_cast<const uint4*>(&tile[rb][4 * qb]);
                *reinterpret_cast<uint4*>(&rtile[rb][4 * qb]) = uint4{to_r6(vb.x), to_r6(vb.y), to_r6(vb.z), to_r6(vb.w)};
            }
        }
        for (int i = tid; i < kSmapRows * kSmapWords / 4; i += 256) reinterpret_cast<uint4*>(smap_flat)[i] = uint4{0u, 0u, 0u, 0u};
        if (tid == 0) {
            n_out = 0;
            n_cand_wg = 0;
        }
        lds_barrier();
        FAST_MARK(2)   // R pass + clears + barrier 1
        FAST_MARK(3)   // next cell's record + copy issue

        const int min_x = (int)(dc_x & 0xffffu), min_y = (int)(dc_x >> 16);
        const int cw = (int)(dc_y & 255u), ch = (int)((dc_y >> 8) & 255u);
        const int iw = cw - 6, ih = ch - 6;   // testable area of this cell (> 0 for every valid cell)
        const float scale = lr.scale;
        bool skip = false;
        if (fmask) {   // upstream: skip the cell if one of its corners is masked (level-0 coordinates, float scale, trunc)
            auto in_mask = [&](unsigned y, unsigned x) {
                return fmask[(size_t)(unsigned)(y * scale) * stride0 + (unsigned)(x * scale)] == 0;
            };
            skip = in_mask(min_y, min_x) || in_mask(min_y + ch, min_x) || in_mask(min_y, min_x + cw) || in_mask(min_y + ch, min_x + cw);
        }
        if (!skip) {
            // candidate-mask layout: bit 8 * b + 2 * R0 + G <-> pixel c0 + 4 * G + b of row row0 + R0. Where the level border clips the testable
            // area (workgroup-uniform, ~9 % of the cells) the pixels outside are masked; the empty asm keeps this a real branch.
            uint32_t valid = 0x0f0f0f0fu;
            if (iw < kCellSize || ih < kCellSize) {
                asm volatile("" ::: "memory");
                valid = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int G = 0; G < 2; ++G) {
                        const uint32_t colok = (c0 + 4 * G + b < iw) ? 1u : 0u;
                        valid |= ((row0 < ih ? colok : 0u) << (8 * b + G)) | ((row0 + 1 < ih ? colok : 0u) << (8 * b + 2 + G));
                    }
            }

            int thr = ini_thr;
            for (;;) {
                // ---- 1. four-diameter test, four pixels per register, on the R tile: candidate mask + "dark end" mask
                uint32_t cmask, dmask;
                {
                    uint32_t w[8][5];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const uint2 b = *reinterpret_cast<const uint2*>(&rtile[row0 + r][2 * run + 2]);
                        w[r][1] = rtile[row0 + r][2 * run + 1];
                        w[r][2] = b.x;
                        w[r][3] = b.y;
                        if (r == 3 || r == 4) {
                            w[r][0] = rtile[row0 + r][2 * run];
                            w[r][4] = rtile[row0 + r][2 * run + 4];
                        }
                    }
                    const uint32_t kk = 0x80808080u - (uint32_t)((thr + 1) >> 2) * 0x01010101u;
                    uint32_t b00, d00, b10, d10, b01, d01, b11, d11;
                    swar_diameter_test<0, 0>(w, kk, b00, d00);
                    swar_diameter_test<1, 0>(w, kk, b10, d10);
                    swar_diameter_test<0, 1>(w, kk, b01, d01);
                    swar_diameter_test<1, 1>(w, kk, b11, d11);
                    constexpr uint32_t kM = 0x80808080u;
                    const uint32_t bm = ((b00 & kM) >> 7) | ((b10 & kM) >> 6) | ((b01 & kM) >> 5) | ((b11 & kM) >> 4);
                    dmask = (((d00 & kM) >> 7) | ((d10 & kM) >> 6) | ((d01 & kM) >> 5) | ((d11 & kM) >> 4)) & valid;
                    cmask = (bm & valid) | dmask;
                    // entry flags: 0x40 = evaluate the dark polarity (the bright test failed), 0x80 = both tests passed (evaluate both)
                    dmask = (dmask & ~bm) | ((dmask & bm) << 4);   // bits 4..7 of every byte are free in the mask layout
                }
                FAST_MARK(4)   // diameter test
                // ---- 2. pool the candidates of the four waves: exclusive prefix inside the wave, one LDS atomic per wave for its base
                {
                    const uint32_t n_mine = (uint32_t)__popc(cmask);
                    const uint32_t incl = wave_prefix_incl(n_mine);
                    const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    uint32_t base = 0;
                    if ((tid & 63) == 0 && wave_total) base = atomicAdd(&n_cand_wg, wave_total);
                    const uint32_t wave_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    uint32_t pos = wave_base + incl - n_mine;
                    // a wave whose range does not fit the list writes nothing: the cell then holds more than kListCap candidates in total and
                    // takes the exhaustive path below, which does not read the list (no per-entry bound check in the common case)
                    if (wave_base + wave_total <= (uint32_t)kListCap) {
                        while (cmask) {
                            const int b = __ffs(cmask) - 1;
                            cmask &= cmask - 1;
                            const uint32_t fl = (((dmask >> b) & 1u) << 6) | (((dmask >> (b + 4)) & 1u) << 7);
                            clist[pos++] = (uint16_t)(((row0 + ((b >> 1) & 1)) << 8) | (c0 + 4 * (b & 1) + (b >> 3)) | fl);
                        }
                    }
                }
                FAST_MARK(5)   // prefix + list writes
                lds_barrier();
                int n_cand = (int)n_cand_wg;
                FAST_MARK(6)   // barrier 2
                if (n_cand > kListCap) {
                    // Dense cell (workgroup-uniform; white noise at a low threshold): more candidates than the list holds. No list then: every
                    // thread scores ITS sixteen pixels exhaustively, both polarities, and suppresses them itself. A pixel the pre-test rejected
                    // has S <= thr, so the complete score map gives the same survivors as the sparse one (whose unscored pixels read 0).
                    for (uint32_t m = valid; m;) {
                        const int b = __ffs(m) - 1;
                        m &= m - 1;
                        const int x = c0 + 4 * (b & 1) + (b >> 3), y = row0 + ((b >> 1) & 1);
                        uint32_t r[16], c;
                        load_ring(tbytes + y * (kTileWords * 4) + x + 3, r, c);
                        uint32_t sc = fast_strength_bright(r, c);
#pragma unroll
                        for (int q = 0; q < 16; ++q) r[q] ^= 0xffu;
                        sc = mx16(sc, fast_strength_bright(r, c ^ 0xffu));
                        sbytes[(y + 1) * (kSmapWords * 4) + 4 + x] = (uint8_t)sc;
                    }
                    lds_barrier();
                    for (uint32_t m = valid; m;) {
                        const int b = __ffs(m) - 1;
                        m &= m - 1;
                        const int x = c0 + 4 * (b & 1) + (b >> 3), y = row0 + ((b >> 1) & 1);
                        const uint8_t* q = sbytes + (y + 1) * (kSmapWords * 4) + 4 + x;
                        const uint32_t sc = q[0];
                        if ((int)sc <= thr) continue;
                        constexpr int kS = kSmapWords * 4;
                        const uint32_t nb = mx16(mx16(mx16((uint32_t)q[-kS - 1], (uint32_t)q[-kS]), mx16((uint32_t)q[-kS + 1], (uint32_t)q[-1])),
                                                 mx16(mx16((uint32_t)q[1], (uint32_t)q[kS - 1]), mx16((uint32_t)q[kS], (uint32_t)q[kS + 1])));
                        if (sc <= nb) continue;
                        const uint32_t o = atomicAdd(&n_out, 1u);
                        olist[o] = (sc << 12) | ((uint32_t)y << 6) | (uint32_t)x;
                    }
                    n_cand = 0;   // the sparse passes below have nothing left to do
                }
                // ---- 3. exact S for the candidates, one per lane, into the score map: one polarity per candidate (both where both tests passed)
                for (int i = tid; i < n_cand; i += 256) {
                    const uint32_t e = clist[i];
                    const int x = e & 63, y = e >> 8;
                    uint32_t r[16], c;
                    load_ring(tbytes + y * (kTileWords * 4) + x + 3, r, c);
                    const uint32_t flip = (e & 0x40u) ? 0xffu : 0u;
#pragma unroll
                    for (int q = 0; q < 16; ++q) r[q] ^= flip;
                    c ^= flip;
                    uint32_t sc = fast_strength_bright(r, c);
                    if (__builtin_amdgcn_ballot_w64((e & 0x80u) != 0) != 0) {   // rare: some lane's pixel passed both tests
                        if (e & 0x80u) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) r[q] ^= 0xffu;
                            sc = mx16(sc, fast_strength_bright(r, c ^ 0xffu));
                        }
                    }
                    sbytes[(y + 1) * (kSmapWords * 4) + 4 + x] = (uint8_t)sc;
                }
                FAST_MARK(7)   // exact scoring
                lds_barrier();
                FAST_MARK(8)   // barrier 3
                // ---- 4. strict NMS over the 8 neighbours (unevaluated neighbours have S <= thr < S(p): 0 in the map), survivors -> olist
                //      (olist aliases the R tile: every wave is past its last R-tile read, the pre-test, by the barriers above; a retry with
                //      min_fast_thr only happens when NO survivor was written, i.e. with the R tile intact)
                for (int i = tid; i < n_cand; i += 256) {
                    const uint32_t e = clist[i] & 0x3f3fu;
                    const int x = e & 255, y = e >> 8;
                    const uint8_t* q = sbytes + (y + 1) * (kSmapWords * 4) + 4 + x;
                    const uint32_t sc = q[0];
                    if ((int)sc <= thr) continue;
                    constexpr int kS = kSmapWords * 4;
                    const uint32_t nb = mx16(mx16(mx16((uint32_t)q[-kS - 1], (uint32_t)q[-kS]), mx16((uint32_t)q[-kS + 1], (uint32_t)q[-1])),
                                             mx16(mx16((uint32_t)q[1], (uint32_t)q[kS - 1]), mx16((uint32_t)q[kS], (uint32_t)q[kS + 1])));
                    if (sc <= nb) continue;
                    // n_out counts the NMS survivors ("keypts_in_cell" before upstream's mask filter); masked ones are dropped below
                    const uint32_t o = atomicAdd(&n_out, 1u);
                    olist[o] = (sc << 12) | ((uint32_t)y << 6) | (uint32_t)x;
                }
                FAST_MARK(9)   // NMS
                lds_barrier();
                FAST_MARK(10)  // barrier 4
                if (n_out != 0 || thr <= min_thr) break;
                // "if keypts_in_cell.empty()": again with min_fast_thr (rare: flat cells). No survivor was written, so the R tile is intact.
                thr = min_thr;
                for (int i = tid; i < kSmapRows * kSmapWords / 4; i += 256) reinterpret_cast<uint4*>(smap_flat)[i] = uint4{0u, 0u, 0u, 0u};
                if (tid == 0) n_cand_wg = 0;
                lds_barrier();
            }
        }
        // one buffer: the raw tile's last readers (the exact scoring of the last pass) are behind barrier 4 -- request the next cell's tile now;
        // it lands under the tail below and under the other workgroups of the CU
        request_next(k);
        if (!skip) {
            // ---- 5. the cell's survivors join the group's buffer; the (frame, level) list is reserved ONCE per group (or when the buffer is
            //      full / the level changes): the reservation's global atomic round trip, a fifth of a cell's time when every cell paid it, is
            //      paid per group. FAST response = S - 1.
            uint32_t total = n_out;
            if (total != 0) {
                if (fmask) {
                    // upstream drops masked keypoints after the empty-cell decision: compact olist in place (rare path, one wave)
                    if (tid < 64) {
                        uint32_t kept = 0;
                        for (uint32_t i0 = 0; i0 < total; i0 += 64) {
                            const uint32_t i = i0 + tid;
                            uint32_t o = 0;
                            bool keep = false;
                            if (i < total) {
                                o = olist[i];
                                const uint32_t gx = min_x + 3 + (o & 63u), gy = min_y + 3 + ((o >> 6) & 63u);
                                keep = fmask[(size_t)(unsigned)(gy * scale) * stride0 + (unsigned)(gx * scale)] != 0;
                            }
                            const unsigned long long bal = __ballot(keep);
                            const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                            if (keep) olist[kept + off] = o;   // kept + off <= i: in-place compaction never overtakes its own reads
                            kept += (uint32_t)__popcll(bal);
                        }
                        if (tid == 0) n_out = kept;
                    }
                    lds_barrier();
                    total = n_out;
                }
                if (total != 0) {
                    if (lr.level != lr_buf.level || n_buf + total > (uint32_t)kWgSurvivors) {
                        flush(lr_buf);
                        lds_barrier();   // the buffer's readers are done before it is refilled below
                    }
                    lr_buf = lr;
                    if (total <= (uint32_t)kWgSurvivors) {
                        for (uint32_t i = tid; i < total; i += 256) wgbuf[n_buf + i] = olist[i] | ((uint32_t)k << 26);
                        n_buf += total;
                    } else {
                        // more survivors than the buffer holds (a cell can have 1024): written straight from olist with their own reservation
                        if (tid == 0) list_base = atomicAdd(&cand_count[frame * L + lr.level], total);
                        lds_barrier();
                        const uint32_t base = list_base;
                        uint64_t* const list = cand + (size_t)frame * cand_frame_entries + lr.cand_off;
                        const uint32_t cap = (uint32_t)lr.cand_cap;
                        for (uint32_t i = tid; i < total; i += 256) {
                            const uint32_t o = olist[i];
                            if (base + i < cap)
                                list[base + i] = cand_pack((uint32_t)(min_x + 3) + (o & 63u), (uint32_t)(min_y + 3) + ((o >> 6) & 63u), ((o >> 12) & 255u) - 1u, 0);
                        }
                    }
                }
            }
        }
        lds_barrier();   // the next cell's staging overwrites the R tile / olist, smap and the counters
        FAST_MARK(11)  // buffer / flush + last barrier
    }
    flush(lr_buf);
    if (kTiming && t_on && (tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) atomicAdd(&tstats[i], t_acc[i]);
        atomicAdd(&tstats[12], 1ull);
    }
#undef FAST_MARK
}

// ================================================================================================================================
// v5 (round 6): the WAVE-AUTONOMOUS form. One wavefront owns one cell from its tile to its survivors: no s_barrier anywhere, no other wave
// to wait for, and the "cell empty -> again with min_fast_thr" decision is a wave-uniform branch. What changed against v4 and why
// (DESIGN.md section 3.1): v4's cell was a latency chain of six workgroup barriers, four partially filled scoring / suppression passes (one
// per wave, whatever the candidate count) and an 8 x 5-word window re-read from LDS by every thread (2.5 words per pixel).
//   * pre-test: lane (band, g) = (lane >> 4, lane & 15) walks DOWN the 16 rows of its band of the cell for the 4-pixel column group g with
//     a rolling register window: every tile row is read ONCE per band (0.69 words per pixel), converted to the 6-bit form R in registers (no R
//     tile, no conversion pass), and its centre-aligned word serves three pixel rows (as ring point 8, centre, ring point 0). The loop is
//     fully unrolled: the window rotates by register renaming.
//   * the pre-test bits of 16 rows x 4 pixels are collected in two registers per polarity (bit 7 of a byte shifted in per row) and compacted
//     ONCE per cell into one list (wave prefix sum by DPP, no atomics);
//   * exact scoring and the 8-neighbour suppression run in passes of 64 candidates, all lanes busy whatever the cell's candidate count;
//     survivors are compacted by ballot / mbcnt in place over the list;
//   * the kernel is bound by latency (a cell's tile comes from HBM; measured: time ~ waves per CU ^ -0.65), so a wave's LDS is kept small:
//     the scores of the passes stay in registers until the last ring has been read, THEN the score map is built over the dead tile
//     (8.2 KB per wave, 19 waves per CU; with a score map of its own 12.9 KB, 12 waves);
//   * the cell's survivors get their list space by an atomic whose return the wave does not wait for (reserve / commit below).
// Work order (both kernels, round 6): consecutive waves of an XCD take consecutive cells of the SAME frame, so the waves launched together
// read neighbouring 64-byte pieces of the same image rows (DRAM pages, L2 lines): 13 % (v4) / 40 % (v5) faster than frames-fastest.
constexpr int kWListCap = 1024;      // candidates of a cell in the list; a denser cell (noise at a low threshold) is scored exhaustively instead
constexpr int kWBufCap = 64;         // survivors of a cell waiting for their list reservation (a cell with more reserves synchronously)
constexpr int kWMaxCells = 64;

// ring of the pixel whose 7x7 neighbourhood's top-left byte is p, rows `pitch` bytes apart (the staged tile, or the level's plane itself)
__device__ __forceinline__ void load_ring_pitch(const uint8_t* p, int pitch, uint32_t (&r)[16], uint32_t& c) {
    c = p[3 * pitch + 3];
    r[0] = p[6 * pitch + 3];
    r[1] = p[6 * pitch + 4];
    r[2] = p[5 * pitch + 5];
    r[3] = p[4 * pitch + 6];
    r[4] = p[3 * pitch + 6];
    r[5] = p[2 * pitch + 6];
    r[6] = p[1 * pitch + 5];
    r[7] = p[4];
    r[8] = p[3];
    r[9] = p[2];
    r[10] = p[1 * pitch + 1];
    r[11] = p[2 * pitch];
    r[12] = p[3 * pitch];
    r[13] = p[4 * pitch];
    r[14] = p[5 * pitch + 1];
    r[15] = p[6 * pitch + 2];
}

template <bool kTiming>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_fast_wave(
    const FrameGeo* __restrict__ geo, const CellDesc* __restrict__ cell_tab, const uint8_t* __restrict__ img0, size_t stride0, size_t frame_stride0,
    const uint8_t* __restrict__ pyr, size_t pyr_frame_bytes, uint64_t* __restrict__ cand, size_t cand_frame_entries, uint32_t* __restrict__ cand_count,
    const uint8_t* __restrict__ mask, int batch, uint32_t batch_magic, int cell_lo, int n_cells, int cells_per_wave, int group_major, int pf_dist, unsigned long long* __restrict__ tstats) {
    // the staged tile; from the moment the exact scoring has read its last ring, its first 4752 bytes are the cell's score map (66 rows x 72 bytes)
    __shared__ __attribute__((aligned(16))) uint32_t tile[kTileRowsMax][kTileWords];
    __shared__ __attribute__((aligned(16))) uint16_t clist[kWListCap];   // (flags << 12) | (y << 6) | x; the suppression's survivors in place (y << 6 | x)
    __shared__ uint8_t cscore[kWListCap];                                // S of the list's candidates until the score map exists
    __shared__ uint64_t wbuf[kWBufCap];                                  // finished candidate entries (cand_pack) waiting for their list reservation
    static_assert(sizeof(tile) >= kSmapRows * kSmapWords * 4, "the score map lives in the tile");

    const int lane = threadIdx.x;
    const int L = geo->num_levels;
    unsigned long long t_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
#define WMARK(i)                                       \
    if (kTiming) {                                     \
        const unsigned long long t_now = clock64();    \
        t_acc[i] += t_now - t_prev;                    \
        t_prev = t_now;                                \
    }
    if (kTiming) t_prev = clock64();
    const int n_groups = (n_cells + cells_per_wave - 1) / cells_per_wave;
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    int frame, group, gid_pf = -1;
    if (group_major) {
        // XCD k takes the k-th contiguous eighth of the (frame, group) sequence, groups fastest
        const int share = (n_groups * batch + 7) >> 3;
        const int gid = xcd * share + idx;
        if (idx >= share || gid >= n_groups * batch) return;
        frame = gid / n_groups;
        group = gid - frame * n_groups;
        const int g2 = gid + pf_dist;
        gid_pf = (g2 < (xcd + 1) * share && g2 < n_groups * batch) ? g2 : -1;
    } else {   // round 3-5: XCD k takes the k-th contiguous eighth of the groups, the frame index runs fastest inside an XCD's share
        const int per_xcd = (n_groups + 7) >> 3;
        const int slot = batch == 1 ? idx : (int)__umulhi((uint32_t)idx, batch_magic);
        frame = idx - slot * batch;
        group = xcd * per_xcd + slot;
        if (slot >= per_xcd || group >= n_groups) return;
    }
    const int cell_first = cell_lo + group * cells_per_wave;
    const int n_here = min(cells_per_wave, cell_lo + n_cells - cell_first);

    auto level_ref_of = [&](int fr, int level) -> LevelRef {
        const LevelGeo& g = geo->lv[level];
        LevelRef r;
        r.level = level;
        if (level == 0) {
            r.img = img0 + (size_t)fr * frame_stride0;
            r.pitch = (int)stride0;
        } else {
            r.img = pyr + (size_t)fr * pyr_frame_bytes + g.plane_off;
            r.pitch = g.pitch;
        }
        r.vec16 = ((r.pitch & 15) == 0) && ((reinterpret_cast<uintptr_t>(r.img) & 15) == 0);
        r.cand_off = g.cand_off;
        r.cand_cap = g.cand_cap;
        r.scale = g.scale;
        return r;
    };
    auto level_ref = [&](int level) -> LevelRef { return level_ref_of(frame, level); };
    // the wave's chunks of a tile: chunk 64 * j + lane = (row, column chunk), j < 6 (350 chunks of 16 bytes); byte offsets from the tile's
    // origin in the current level's plane, recomputed at a level boundary only
    uint32_t voff[6];
    auto chunk_offsets = [&](const LevelRef& lv) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int i = 64 * j + lane, r = (i * 0x3334) >> 16, q = i - 5 * r;
            voff[j] = (uint32_t)(r * lv.pitch + 16 * q);
        }
    };
    const uint32_t lds_tile0 = lds_addr(&tile[0][0]);
    // tile byte u of row r <-> image (min_x - 3 + u, min_y + r). Rows >= ch and chunks at or beyond the plane's pitch are NOT copied: those
    // bytes keep what was there (an earlier cell's score map) and only feed pixels outside the testable area, which the `valid` masks remove.
    auto issue_tile = [&](const LevelRef& lv, uint32_t rec_x, uint32_t rec_y) {
        const int min_x = (int)(rec_x & 0xffffu), min_y = (int)(rec_x >> 16), cw = (int)(rec_y & 255u), ch = (int)((rec_y >> 8) & 255u);
        const int gx = min_x - 3;
        const uint8_t* const base = lv.img + (size_t)min_y * (size_t)lv.pitch + (size_t)gx;
        if (lv.vec16) {
            if (cw == kTileRowsMax && ch == kTileRowsMax) {   // a whole cell: every chunk lies inside the plane (gx + 73 <= cols <= pitch)
#pragma unroll
                for (int j = 0; j < 5; ++j) glds16(base, voff[j], lds_tile0 + 1024u * (uint32_t)j);
                if (lane < kChunks - 320) glds16(base, voff[5], lds_tile0 + 5120u);
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int i = 64 * j + lane, r = (i * 0x3334) >> 16, q = i - 5 * r;
                    if (i < kChunks && r < ch && gx + 16 * q < lv.pitch) glds16(base, voff[j], lds_tile0 + 1024u * (uint32_t)j);
                }
            }
        } else {   // 4-byte aligned base / pitch (enforced by the ABI): 1400 words, 64 per wave-instruction
            for (int j = 0; j < 22; ++j) {
                const int i = 64 * j + lane;
                const int r = (i * 0x0ccd) >> 16, c4 = i - 20 * r;   // i / 20 for i < 1536
                if (i < kTileRowsMax * kTileWords && r < ch && gx + 4 * c4 + 4 <= lv.pitch)
                    glds4(base, (uint32_t)(r * lv.pitch + 4 * c4), lds_tile0 + 256u * (uint32_t)j);
            }
        }
    };

    // L2 prefetch for a LATER wave of this XCD: three plain loads per lane whose values nobody reads (rows 0 .. 69 x the tile's first and last
    // word: both 128-byte lines a tile row can touch), issued BEHIND this wave's first copy so that the copy's wait can leave them in flight
    // (loads retire in issue order: s_waitcnt vmcnt(3)). The later wave's copy then hits the XCD's L2 instead of waiting for HBM. The three
    // registers stay allocated until the values are "used" by an empty statement at the end of the wave's first cell.
    uint32_t pf0 = 0, pf1 = 0, pf2 = 0;
    bool pf_behind_copy = false;
    auto prefetch_l2 = [&](const LevelRef& lv, uint32_t rec_x, uint32_t rec_y) {
        const int min_x = (int)(rec_x & 0xffffu), min_y = (int)(rec_x >> 16), ch = (int)((rec_y >> 8) & 255u);
        const uint8_t* const base = lv.img + (size_t)min_y * (size_t)lv.pitch + (size_t)(min_x - 3);
        const int ra = min(lane, ch - 1), rb = min(64 + (lane & 7), ch - 1), wb = (lane & 8) ? 76 : 0;
        pf0 = *reinterpret_cast<const volatile uint32_t*>(base + (size_t)ra * (size_t)lv.pitch);
        pf1 = *reinterpret_cast<const volatile uint32_t*>(base + (size_t)ra * (size_t)lv.pitch + 76);
        pf2 = *reinterpret_cast<const volatile uint32_t*>(base + (size_t)rb * (size_t)lv.pitch + wb);
        pf_behind_copy = true;
    };

    const int band = lane >> 4, g = lane & 15;
    const uint8_t* const tbytes = reinterpret_cast<const uint8_t*>(&tile[0][0]);
    uint8_t* const sbytes = reinterpret_cast<uint8_t*>(&tile[0][0]);   // the score map: same bytes, later
    const uint32_t* const trow = &tile[16 * band][g];   // the lane's words g .. g + 3 of its band's 22 tile rows
    const uint8_t* const fmask = mask ? mask + (size_t)frame * frame_stride0 : nullptr;
    const int ini_thr = geo->ini_thr, min_thr = geo->min_thr;
    constexpr int kP = kTileWords * 4, kS = kSmapWords * 4;
    auto clear_smap = [&]() {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int i = lane + 64 * j;
            if (i < kSmapRows * kSmapWords / 4) reinterpret_cast<uint4*>(&tile[0][0])[i] = uint4{0u, 0u, 0u, 0u};
        }
    };
    static_assert((kSmapRows * kSmapWords) % 4 == 0 && kSmapRows * kSmapWords / 4 <= 5 * 64, "score map is cleared with five 16-byte stores per lane");

    const uint2 d0 = reinterpret_cast<const uint2*>(cell_tab)[cell_first];
    uint32_t dn_x = d0.x, dn_y = d0.y;
    LevelRef lr_next = level_ref((int)((dn_y >> 16) & 255u));
    chunk_offsets(lr_next);
    issue_tile(lr_next, dn_x, dn_y);
    if (pf_dist > 0 && gid_pf >= 0) {
        // the first cell of the group `pf_dist` groups ahead in this XCD's sequence: some wave of this XCD will copy it a few microseconds from now
        const int f2 = gid_pf / n_groups, g2 = gid_pf - f2 * n_groups;
        const uint2 d1 = reinterpret_cast<const uint2*>(cell_tab)[cell_lo + g2 * cells_per_wave];
        LevelRef l1 = level_ref_of(f2, (int)((d1.y >> 16) & 255u));
        prefetch_l2(l1, d1.x, d1.y);
    }

    // A cell's survivors go to their (frame, level) list with ONE reservation -- whose atomic round trip (microseconds under load) the wave does
    // not wait for: reserve() only issues it; the entries are written by commit() behind the next cell's tile wait, whose s_waitcnt vmcnt(0)
    // has retired the atomic by then (or at the end of the wave). The buffer is not written in between.
    uint32_t pend_n = 0, pend_v = 0;   // entries wbuf[0, pend_n) wait for the base that lane 0 of pend_v will hold
    int64_t pend_off = 0;
    uint32_t pend_cap = 0;
    auto reserve = [&](const LevelRef& lv, uint32_t n) {
        uint32_t b0 = 0;
        if (lane == 0) b0 = atomicAdd(&cand_count[frame * L + lv.level], n);
        pend_v = b0;
        pend_n = n;
        pend_off = lv.cand_off;
        pend_cap = (uint32_t)lv.cand_cap;
    };
    auto commit = [&]() {
        const uint32_t nb = pend_n;
        if (nb != 0) {
            const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend_v);
            uint64_t* const list = cand + (size_t)frame * cand_frame_entries + pend_off;
            if ((uint32_t)lane < nb && base + (uint32_t)lane < pend_cap) list[base + (uint32_t)lane] = wbuf[lane];
            pend_n = 0;
        }
    };

    for (int k = 0; k < n_here; ++k) {
        WMARK(0)   // previous cell's tail
        const uint32_t dc_x = dn_x, dc_y = dn_y;
        const LevelRef lr = lr_next;
        // the next cell's record: a scalar load issued a whole cell before its use
        const uint2 dsc_next = reinterpret_cast<const uint2*>(cell_tab)[cell_first + min(k + 1, n_here - 1)];
        const int min_x = (int)(dc_x & 0xffffu), min_y = (int)(dc_x >> 16);
        const int cw = (int)(dc_y & 255u), ch = (int)((dc_y >> 8) & 255u);
        const int iw = cw - 6, ih = ch - 6;   // testable area (> 0 for every valid cell)
        const float scale = lr.scale;
        bool skip = false;
        if (fmask) {   // upstream: skip the cell if one of its corners is masked (level-0 coordinates, float scale, trunc)
            auto in_mask = [&](unsigned y, unsigned x) {
                return fmask[(size_t)(unsigned)(y * scale) * stride0 + (unsigned)(x * scale)] == 0;
            };
            skip = in_mask(min_y, min_x) || in_mask(min_y + ch, min_x) || in_mask(min_y, min_x + cw) || in_mask(min_y + ch, min_x + cw);
        }
        // valid-pixel masks of the lane's two 8-row blocks: bit t of byte b <-> pixel (4 g + b, 16 band + 8 blk + t)
        uint32_t valid0 = 0xffffffffu, valid1 = 0xffffffffu;
        if (iw < kCellSize || ih < kCellSize) {
            asm volatile("" ::: "memory");
            uint32_t colm = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) colm |= (4 * g + b < iw) ? (0xffu << (8 * b)) : 0u;
            const int n0 = min(8, max(0, ih - 16 * band)), n1 = min(8, max(0, ih - 16 * band - 8));
            valid0 = colm & (((1u << n0) - 1u) * 0x01010101u);
            valid1 = colm & (((1u << n1) - 1u) * 0x01010101u);
        }
        // this wave's own copies: nobody else reads or writes this tile
        if (pf_behind_copy) {
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            pf_behind_copy = false;
        } else {
            wait_tile();
        }
        WMARK(1)   // tile wait
        commit();   // the previous cell's survivors: their reservation has returned

        uint32_t n_out = 0;   // wave-uniform: suppression survivors of this cell (in place over the list)
        if (!skip) {
            int thr = ini_thr;
            for (;;) {
                // ---- 1. four-diameter test on R = (p >> 2) | 0x80, four pixels per register, rolling down the band's rows
                uint32_t ab0 = 0, ab1 = 0, ad0 = 0, ad1 = 0;   // bright / dark pre-test bits of the two 8-row blocks
                {
                    const uint32_t kk = 0x80808080u - (uint32_t)((thr + 1) >> 2) * 0x01010101u;
                    constexpr uint32_t kM = 0x80808080u;
                    uint32_t cen[22], l2[22], r2[22], l3[22], r3[22];
                    // the rows' words are requested kAhead steps before they are used (a step is ~45 vector instructions: the LDS latency is
                    // covered twice); the scheduling barriers keep hipcc from hoisting all 76 reads (and their 100 registers) to the top
                    constexpr int kAhead = 2;
                    uint32_t raw[22][4];
                    auto request = [&](int rr) {
                        const uint2 m = *reinterpret_cast<const uint2*>(trow + rr * kTileWords + 1);   // ds_read2_b32: any 4-byte alignment
                        raw[rr][1] = m.x;
                        raw[rr][2] = m.y;
                        if (rr >= 3 && rr < 19) {   // rows that are a pixel row's centre row
                            raw[rr][0] = trow[rr * kTileWords];
                            raw[rr][3] = trow[rr * kTileWords + 3];
                        }
                    };
#pragma unroll
                    for (int rr = 0; rr < kAhead; ++rr) request(rr);
#pragma unroll
                    for (int rr = 0; rr < 22; ++rr) {
                        if (rr + kAhead < 22) request(rr + kAhead);
                        // pixel group at tile bytes 4 g + 6 .. 4 g + 9: words g + 1 (bytes 2, 3) and g + 2 (bytes 0, 1)
                        const uint32_t w1 = to_r6(raw[rr][1]), w2 = to_r6(raw[rr][2]);
                        l2[rr] = w1;                                      // x - 2 .. x + 1
                        r2[rr] = w2;                                      // x + 2 .. x + 5
                        cen[rr] = __builtin_amdgcn_alignbyte(w2, w1, 2);   // x .. x + 3
                        if (rr >= 3 && rr < 19) {
                            const uint32_t w0 = to_r6(raw[rr][0]), w3 = to_r6(raw[rr][3]);
                            l3[rr] = __builtin_amdgcn_alignbyte(w1, w0, 3);   // x - 3
                            r3[rr] = __builtin_amdgcn_alignbyte(w3, w2, 1);   // x + 3
                        }
                        if (rr >= 6) {
                            const int j = rr - 6;   // pixel row of the band: centre = tile row j + 3
                            const uint32_t c = cen[j + 3];
                            const uint32_t cb = c - kk, cd = c + kk;
                            const uint32_t p0 = cen[j + 6], p8 = cen[j], p4 = r3[j + 3], p12 = l3[j + 3];
                            const uint32_t p2 = r2[j + 5], p14 = l2[j + 5], p6 = r2[j + 1], p10 = l2[j + 1];
                            const uint32_t bright = and_or3(and_or3(and_or3((p0 - cb) | (p8 - cb), p2 - cb, p10 - cb), p4 - cb, p12 - cb), p6 - cb, p14 - cb);
                            const uint32_t dark = and_or3(and_or3(and_or3((cd - p0) | (cd - p8), cd - p2, cd - p10), cd - p4, cd - p12), cd - p6, cd - p14);
                            if (j < 8) {
                                ab0 = (ab0 >> 1) | (bright & kM);
                                ad0 = (ad0 >> 1) | (dark & kM);
                            } else {
                                ab1 = (ab1 >> 1) | (bright & kM);
                                ad1 = (ad1 >> 1) | (dark & kM);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                WMARK(2)   // diameter test
                // ---- 2. one list per cell: exclusive prefix of the lanes' counts, entries (flags << 12) | (y << 6) | x
                //      flags: 1 = evaluate the dark polarity (the bright test failed), 2 = both tests passed (evaluate both)
                uint32_t cm0 = (ab0 | ad0) & valid0, cm1 = (ab1 | ad1) & valid1;
                const uint32_t n_mine = (uint32_t)(__popc(cm0) + __popc(cm1));
                const uint32_t incl = wave_prefix_incl(n_mine);
                const int n_cand = __builtin_amdgcn_readlane((int)incl, 63);
                if (n_cand <= kWListCap) {
                    uint32_t pos = incl - n_mine;
                    const uint32_t ebase = ((uint32_t)(16 * band) << 6) | (uint32_t)(4 * g);
                    while (cm0) {
                        const int b = __ffs(cm0) - 1;
                        cm0 &= cm0 - 1;
                        const uint32_t dk = (ad0 >> b) & 1u, br = (ab0 >> b) & 1u;
                        clist[pos++] = (uint16_t)((ebase + (((uint32_t)b & 7u) << 6) + ((uint32_t)b >> 3)) | ((dk & ~br) << 12) | ((dk & br) << 13));
                    }
                    while (cm1) {
                        const int b = __ffs(cm1) - 1;
                        cm1 &= cm1 - 1;
                        const uint32_t dk = (ad1 >> b) & 1u, br = (ab1 >> b) & 1u;
                        clist[pos++] = (uint16_t)((ebase + (8u << 6) + (((uint32_t)b & 7u) << 6) + ((uint32_t)b >> 3)) | ((dk & ~br) << 12) | ((dk & br) << 13));
                    }
                }
                WMARK(3)   // prefix + list writes
                if (n_cand > kWListCap) {
                    // Dense cell (white noise at a low threshold): more candidates than the list holds. Every lane scores ITS 64 pixels
                    // exhaustively, both polarities, and suppresses them itself. A pixel the pre-test rejected has S <= thr, so the complete
                    // score map gives the same survivors as the sparse one (whose unscored pixels read 0). The score map takes the tile's
                    // place, so the rings come from the level's plane itself (slow and rare).
                    clear_smap();
                    const uint8_t* const org = lr.img + (size_t)min_y * (size_t)lr.pitch + (size_t)min_x;   // image byte of tile (row 0, byte 3)
                    for (int blk = 0; blk < 2; ++blk)
                        for (uint32_t m = blk ? valid1 : valid0; m;) {
                            const int b = __ffs(m) - 1;
                            m &= m - 1;
                            const int x = 4 * g + (b >> 3), y = 16 * band + 8 * blk + (b & 7);
                            uint32_t r[16], c;
                            load_ring_pitch(org + (size_t)y * (size_t)lr.pitch + x, lr.pitch, r, c);
                            uint32_t sc = fast_strength_bright(r, c);
#pragma unroll
                            for (int q = 0; q < 16; ++q) r[q] ^= 0xffu;
                            sc = mx16(sc, fast_strength_bright(r, c ^ 0xffu));
                            sbytes[(y + 1) * kS + 4 + x] = (uint8_t)sc;
                        }
                    for (int blk = 0; blk < 2; ++blk)
                        for (uint32_t m = blk ? valid1 : valid0; __builtin_amdgcn_ballot_w64(m != 0) != 0;) {   // wave-uniform trip count: ballots inside
                            const bool on = m != 0;
                            const int b = on ? __ffs(m) - 1 : 0;
                            m &= m - 1;
                            const int x = 4 * g + (b >> 3), y = 16 * band + 8 * blk + (b & 7);
                            const uint8_t* q = sbytes + (y + 1) * kS + 4 + x;
                            const uint32_t sc = q[0];
                            const uint32_t nb = mx16(mx16(mx16((uint32_t)q[-kS - 1], (uint32_t)q[-kS]), mx16((uint32_t)q[-kS + 1], (uint32_t)q[-1])),
                                                     mx16(mx16((uint32_t)q[1], (uint32_t)q[kS - 1]), mx16((uint32_t)q[kS], (uint32_t)q[kS + 1])));
                            const bool keep = on && (int)sc > thr && sc > nb;
                            const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
                            const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                            if (keep) clist[n_out + off] = (uint16_t)((y << 6) | x);   // survivors are pairwise non-adjacent: at most 1024
                            n_out += (uint32_t)__popcll(bal);
                        }
                } else {
                    // ---- 3. exact S for the candidates, 64 per pass: one polarity per candidate (both where both tests passed). The scores
                    //      wait beside the list until the last ring has been read ...
                    for (int i0 = 0; i0 < n_cand; i0 += 64) {
                        const int i = i0 + lane;
                        if (i < n_cand) {
                            const uint32_t e = clist[i];
                            const int x = e & 63, y = (e >> 6) & 63;
                            uint32_t r[16], c;
                            load_ring_pitch(tbytes + y * kP + x + 3, kP, r, c);
                            const uint32_t flip = (e & 0x1000u) ? 0xffu : 0u;
#pragma unroll
                            for (int q = 0; q < 16; ++q) r[q] ^= flip;
                            c ^= flip;
                            uint32_t sc = fast_strength_bright(r, c);
                            if (__builtin_amdgcn_ballot_w64((e & 0x2000u) != 0) != 0) {   // rare: some lane's pixel passed both tests
                                if (e & 0x2000u) {
#pragma unroll
                                    for (int q = 0; q < 16; ++q) r[q] ^= 0xffu;
                                    sc = mx16(sc, fast_strength_bright(r, c ^ 0xffu));
                                }
                            }
                            cscore[i] = (uint8_t)sc;
                        }
                    }
                    WMARK(4)   // exact scoring
                    // ... then the score map takes the tile's place: zero wherever no candidate was scored (such pixels have S <= thr)
                    clear_smap();
                    for (int i0 = 0; i0 < n_cand; i0 += 64) {
                        const int i = i0 + lane;
                        if (i < n_cand) {
                            const uint32_t e = clist[i];
                            sbytes[(((e >> 6) & 63u) + 1u) * kS + 4u + (e & 63u)] = cscore[i];
                        }
                    }
                    // ---- 4. strict suppression over the 8 neighbours (unevaluated neighbours have S <= thr < S(p): 0 in the map); survivors are
                    //      compacted in place over the list (a pass writes at most as many entries as it has read)
                    for (int i0 = 0; i0 < n_cand; i0 += 64) {
                        const int i = i0 + lane;
                        bool keep = false;
                        uint32_t e = 0;
                        if (i < n_cand) {
                            e = clist[i] & 0x0fffu;
                            const int x = e & 63, y = e >> 6;
                            const uint8_t* q = sbytes + (y + 1) * kS + 4 + x;
                            const uint32_t sc = q[0];
                            if ((int)sc > thr) {
                                const uint32_t nb = mx16(mx16(mx16((uint32_t)q[-kS - 1], (uint32_t)q[-kS]), mx16((uint32_t)q[-kS + 1], (uint32_t)q[-1])),
                                                         mx16(mx16((uint32_t)q[1], (uint32_t)q[kS - 1]), mx16((uint32_t)q[kS], (uint32_t)q[kS + 1])));
                                keep = sc > nb;
                            }
                        }
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
                        const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                        if (keep) clist[n_out + off] = (uint16_t)e;
                        n_out += (uint32_t)__popcll(bal);
                    }
                }
                WMARK(5)   // score map + suppression
                if (n_out != 0 || thr <= min_thr) break;
                // "if keypts_in_cell.empty()": again with min_fast_thr (rare: flat cells). The score map has taken the tile's place: copy it again.
                thr = min_thr;
                issue_tile(lr, dc_x, dc_y);
                wait_tile();
            }
        }
        // ---- 5. the cell's survivors: n_out counts "keypts_in_cell" before upstream's mask filter; masked ones are dropped here. FAST response = S - 1.
        uint32_t total = n_out;
        if (total != 0 && fmask) {
            uint32_t kept = 0;
            for (uint32_t i0 = 0; i0 < total; i0 += 64) {
                const uint32_t i = i0 + lane;
                uint32_t o = 0;
                bool keep = false;
                if (i < total) {
                    o = clist[i];
                    const uint32_t gx = min_x + 3 + (o & 63u), gy = min_y + 3 + ((o >> 6) & 63u);
                    keep = fmask[(size_t)(unsigned)(gy * scale) * stride0 + (unsigned)(gx * scale)] != 0;
                }
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
                const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (keep) clist[kept + off] = (uint16_t)o;   // kept + off <= i
                kept += (uint32_t)__popcll(bal);
            }
            total = kept;
        }
        if (total != 0) {
            if (total <= (uint32_t)kWBufCap) {
                if ((uint32_t)lane < total) {
                    const uint32_t o = clist[lane];
                    const uint32_t sc = sbytes[((o >> 6) + 1u) * kS + 4u + (o & 63u)];
                    wbuf[lane] = cand_pack((uint32_t)(min_x + 3) + (o & 63u), (uint32_t)(min_y + 3) + (o >> 6), sc - 1u, 0);
                }
                reserve(lr, total);
            } else {
                // more survivors than the buffer holds (a cell can have 1024): written straight from the list, waiting for their reservation
                uint32_t b0 = 0;
                if (lane == 0) b0 = atomicAdd(&cand_count[frame * L + lr.level], total);
                const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
                uint64_t* const list = cand + (size_t)frame * cand_frame_entries + lr.cand_off;
                const uint32_t cap = (uint32_t)lr.cand_cap;
                for (uint32_t i = lane; i < total; i += 64) {
                    const uint32_t o = clist[i];
                    const uint32_t sc = sbytes[((o >> 6) + 1u) * kS + 4u + (o & 63u)];
                    if (base + i < cap) list[base + i] = cand_pack((uint32_t)(min_x + 3) + (o & 63u), (uint32_t)(min_y + 3) + (o >> 6), sc - 1u, 0);
                }
            }
        }
        WMARK(6)   // survivors
        asm volatile("" ::"v"(pf0), "v"(pf1), "v"(pf2));   // the prefetch's registers were reserved up to here (its loads landed long ago)
        // the score map's last readers are behind this wave: request the next cell's tile
        if (k + 1 < n_here) {
            dn_x = dsc_next.x;
            dn_y = dsc_next.y;
            const int nl = (int)((dn_y >> 16) & 255u);
            if (nl != lr_next.level) {
                lr_next = level_ref(nl);
                chunk_offsets(lr_next);
            }
            issue_tile(lr_next, dn_x, dn_y);
        }
        WMARK(7)   // next tile's request
    }
    commit();
    if (kTiming && lane == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) atomicAdd(&tstats[i], t_acc[i]);
        atomicAdd(&tstats[12], 1ull);
    }
#undef WMARK
}

hipError_t launch_fast(const FrameGeo& hgeo, const DevBuffers& d, const uint8_t* img0, size_t stride0, size_t frame_stride0,
                       const uint8_t* mask, int mask_rows, int batch, hipStream_t s, int cell_lo, int n_cells) {
    (void)mask_rows;
    if (n_cells < 0) n_cells = hgeo.total_cells - cell_lo;
    if (n_cells <= 0 || batch <= 0) return hipSuccess;
    const Tuning& tn = tuning();   // environment switches, read ONCE per process
    const unsigned per_xcd = (unsigned)(n_cells + 7) / 8u;
    // slot = idx / batch as a multiply-high: exact while idx * batch < 2^32 (idx < per_xcd * batch)
    if ((uint64_t)per_xcd * (uint64_t)batch * (uint64_t)batch >= (1ull << 32)) return hipErrorInvalidValue;
    const uint32_t batch_magic = batch > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)batch - 1) / (uint64_t)batch) : 0u;
    const long long launch_cells = (long long)n_cells * batch;
    const bool wave_form = tn.fast_impl == 2;
    // Cells per group. v4 (workgroup per cell): consecutive cells amortise a group's set-up and let the next tile's copy overlap the current
    // cell's tail; with the group-major work order (round 6) short groups are best, because the waves in flight then cover a compact run of
    // cell rows whose image lines are fetched once (measured at 256 frames, 2 / 3 / 4 / 6 cells: 1.603 / 1.600 / 1.626 / 1.66 ms; one frame:
    // 22.1 us with two, 26.4 with one). v5 (wave per cell): one cell per wave (1 / 2 / 3 / 8: 0.452 / 0.482 / 0.526 / 0.671 ms per 64 frames).
    const int cells_auto = wave_form ? 1 : (launch_cells >= 65536 ? 3 : 2);
    const int cells_per_wg = tn.fast_cells > 0 ? std::min(tn.fast_cells, wave_form ? kWMaxCells : kMaxCellsPerWg) : cells_auto;
    const unsigned n_groups = ((unsigned)n_cells + (unsigned)cells_per_wg - 1) / (unsigned)cells_per_wg, gper = (n_groups + 7) / 8u;
    const dim3 grid(8u * gper * (unsigned)batch), block(wave_form ? 64 : 256);
    const size_t pad_lds = (size_t)tn.fast_pad_lds;   // occupancy probe: extra dynamic LDS per workgroup (0 in production)
    if (tn.fast_timing) {   // tuning aid: per-phase shader cycles (v4: of wave 0 of every workgroup; v5: of every wave), printed per launch
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxTuningDevices) return hipErrorInvalidDevice;
        static unsigned long long* d_t[kMaxTuningDevices] = {};
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        if (!d_t[dev] && hipMalloc(&d_t[dev], 16 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_t[dev], 0, 16 * sizeof(unsigned long long), s);
        if (wave_form)
            hipLaunchKernelGGL((k_fast_wave<true>), grid, block, pad_lds, s, d.geo, d.cells, img0, stride0, frame_stride0, d.pyr, d.pyr_frame_bytes, d.cand,
                               d.cand_frame_entries, d.cand_count, mask, batch, batch_magic, cell_lo, n_cells, cells_per_wg, tn.fast_map, tn.fast_pf, d_t[dev]);
        else
            hipLaunchKernelGGL((k_fast_cells<true>), grid, block, pad_lds, s, d.geo, d.cells, img0, stride0, frame_stride0, d.pyr, d.pyr_frame_bytes, d.cand,
                               d.cand_frame_entries, d.cand_count, mask, batch, batch_magic, cell_lo, n_cells, cells_per_wg, tn.fast_map, d_t[dev]);
        unsigned long long h_t[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h_t, d_t[dev], sizeof(h_t), hipMemcpyDeviceToHost);
        static const char* nm4[12] = {"loop", "tile-wait", "rpass+bar1", "next-issue", "pretest", "pool", "bar2", "score", "bar3", "nms", "bar4", "append+bar5"};
        static const char* nm5[12] = {"tail", "tile-wait", "pretest", "compact", "score", "map+suppress", "survivors", "next-issue", "-", "-", "-", "-"};
        const char* const* nm = wave_form ? nm5 : nm4;
        unsigned long long tot = 0;
        for (int i = 0; i < 12; ++i) tot += h_t[i];
        fprintf(stderr, "[%s timing] %llu %s x %d cells, %.0f cycles per cell:", wave_form ? "k_fast_wave" : "k_fast_cells", h_t[12],
                wave_form ? "waves" : "workgroups", cells_per_wg, (double)tot / ((double)h_t[12] * cells_per_wg + 1e-9));
        for (int i = 0; i < (wave_form ? 8 : 12); ++i) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * (double)h_t[i] / (double)(tot + 1));
        fprintf(stderr, "\n");
        return hipGetLastError();
    }
    if (wave_form)
        hipLaunchKernelGGL((k_fast_wave<false>), grid, block, pad_lds, s, d.geo, d.cells, img0, stride0, frame_stride0, d.pyr, d.pyr_frame_bytes, d.cand,
                           d.cand_frame_entries, d.cand_count, mask, batch, batch_magic, cell_lo, n_cells, cells_per_wg, tn.fast_map, tn.fast_pf, (unsigned long long*)nullptr);
    else
        hipLaunchKernelGGL((k_fast_cells<false>), grid, block, pad_lds, s, d.geo, d.cells, img0, stride0, frame_stride0, d.pyr, d.pyr_frame_bytes, d.cand,
                           d.cand_frame_entries, d.cand_count, mask, batch, batch_magic, cell_lo, n_cells, cells_per_wg, tn.fast_map, (unsigned long long*)nullptr);
    return hipGetLastError();
}

}   // namespace ovs
