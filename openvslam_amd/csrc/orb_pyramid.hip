// orb_pyramid.hip -- A1: orb_extractor::compute_image_pyramid (expected: src/openvslam/feature/orb_extractor.cc), i.e.
// cv::resize(prev_level, level, size, 0, 0, INTER_LINEAR) for CV_8UC1 in OpenCV's 11-bit fixed point.
// Integer-only on the device: the float/double part of OpenCV (source coordinate, coefficient rounding) is evaluated
// once per geometry on the host into ResizeTap tables (orb_api.hip: build_taps) so no device float can change a pixel.
//
// HBM-bound streaming kernel. A 256-thread workgroup produces a 128 x 32 output tile: the source rectangle its taps
// touch (<= 160 x 41 bytes at scale 1.2) is staged in LDS with coalesced, aligned u32 loads -- v1 gathered 16 single bytes
// per thread straight from global memory and was bound by the texture-addresser rate, not by bandwidth -- then the two
// fixed-point passes run separably through LDS (horizontal pass once per staged source row into 16-bit words, vertical pass
// per output row) and every thread stores 4 pixels as one aligned u32 (two such groups per thread). v2 evaluated the horizontal
// pass per output pixel (twice per source row on average) and was VALU-bound at ~38 instructions per pixel. Algorithmic traffic per level = source plane + destination plane, each touched once
// (tile halos overlap by one row/column and hit L2).
#include "ovs_common.h"

#include <cstdlib>

namespace ovs {

constexpr int kTileW = 128, kTileH = 32;
constexpr int kGroups = kTileW * kTileH / 4 / 256;   // 4-pixel groups per thread
constexpr int kSlots = 8;                            // u32 staging loads per thread in flight (generic-alignment path)
constexpr int kSrcWords = 48;    // 192-byte LDS pitch: source span of 128 output px is <= 128*src/dst + 2 <= 160 at scale <= 1.25, + 15
                                 // bytes so that the staged rectangle starts on a 16-byte boundary (16-byte staging loads)
constexpr int kRowChunks = kSrcWords / 4;   // 16-byte chunks per staged row
constexpr int kChunkSlots = 3;              // 44 rows x 12 chunks <= 3 * 256
constexpr int kSrcRows = 44;    // 32 output rows span <= 32 * 1.25 + 2 source rows (16-row tiles: 0.167 ms, 32: 0.142, 64: 0.165)

// high 32 bits of the 48-bit product of two 24-bit operands (full-rate VOP3; hipcc has no builtin for it)
__device__ __forceinline__ uint32_t mulhi_u24(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__global__ __launch_bounds__(256) void k_resize_linear_u8_generic(const uint8_t* __restrict__ src, size_t src_frame_stride, int src_pitch,
                                                         int srows, int scols, uint8_t* __restrict__ dst, size_t dst_frame_stride,
                                                         int dst_pitch, int drows, int dcols, const ResizeTap* __restrict__ xt,
                                                         const ResizeTap* __restrict__ yt, int tiles_x, int tiles_y, int batch, float inv_tiles_x,
                                                         float inv_tiles_frame) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[kSrcRows][kSrcWords];
    __shared__ __attribute__((aligned(8))) uint32_t hrow[kSrcRows][kTileW / 2];   // horizontal pass, two u16 per word
    const int tid = threadIdx.x;
    // XCD-aware tile order (workgroup b runs on XCD b % 8): XCD k takes the k-th contiguous eighth of the (frame, tile row, tile
    // column) sequence, so tiles that share halo source lines share an L2. 1-D grid padded to a multiple of 8.
    const int per_xcd = gridDim.x >> 3;
    const int tile_id = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile_id >= tiles_x * tiles_y * batch) return;
    // exact for tile_id < 2^22: (n + 0.5) / d is at least 0.5 / d away from an integer, far more than the float error
    const int frame = (int)(((float)tile_id + 0.5f) * inv_tiles_frame);
    const int trem = tile_id - frame * (tiles_x * tiles_y);
    const int tyi = (int)(((float)trem + 0.5f) * inv_tiles_x), txi = trem - tyi * tiles_x;
    const int x0 = txi * kTileW, y0 = tyi * kTileH;
    const int x1 = min(x0 + kTileW, dcols) - 1, y1 = min(y0 + kTileH, drows) - 1;   // last output pixel of the tile
    const uint8_t* s = src + (size_t)frame * src_frame_stride;
    uint8_t* d = dst + (size_t)frame * dst_frame_stride;
    // source rectangle touched by the tile's taps (tables are monotone)
    const int sx_lo = xt[x0].o0 & ~15, sx_hi = xt[x1].o1;
    const int sy_lo = yt[y0].o0, sy_hi = yt[y1].o1;
    const int nwords = (sx_hi - sx_lo) / 4 + 1, nrows = sy_hi - sy_lo + 1;
    if (nwords <= kSrcWords && nrows <= kSrcRows && (nrows - 1) * kSrcWords + nwords <= kSlots * 256 && nrows * kRowChunks <= kChunkSlots * 256) {   // kSlots staging slots per thread
        // taps of this thread's column pair and of its two output rows: issued before the staging loads so their latency overlaps
        const int xp = tid & 63, q = tid >> 6;
        const int xa = min(x0 + 2 * xp, dcols - 1), xb = min(x0 + 2 * xp + 1, dcols - 1);
        const ResizeTap ta = xt[xa], tbp = xt[xb];
        ResizeTap tys[kGroups];
#pragma unroll
        for (int g = 0; g < kGroups; ++g) tys[g] = yt[min(y0 + 8 * g + (tid >> 5), drows - 1)];
        // stage the source rectangle. Planes whose rows are 16-byte aligned (every pyramid level; level 0 when the caller's stride and
        // base allow): three 16-byte loads per thread in flight, one ds_write_b128 each. Otherwise aligned u32 loads, eight in flight.
        if (((src_pitch & 15) == 0) && ((reinterpret_cast<uintptr_t>(s) & 15) == 0)) {
            uint4 v[kChunkSlots];
            int slot[kChunkSlots];
#pragma unroll
            for (int k = 0; k < kChunkSlots; ++k) {
                const int i = tid + 256 * k;
                const int r = i / kRowChunks, c = i - r * kRowChunks;
                slot[k] = (r < nrows && 4 * c < nwords) ? i : -1;
                v[k] = uint4{0u, 0u, 0u, 0u};
                if (slot[k] >= 0) {
                    const int gx = sx_lo + 16 * c;
                    const uint8_t* p = s + (size_t)(sy_lo + r) * src_pitch + gx;
                    if (gx + 16 <= src_pitch) {
                        v[k] = *reinterpret_cast<const uint4*>(p);
                    } else {   // row tail
                        const uint32_t* p4 = reinterpret_cast<const uint32_t*>(p);
                        if (gx + 4 <= src_pitch) v[k].x = p4[0];
                        if (gx + 8 <= src_pitch) v[k].y = p4[1];
                        if (gx + 12 <= src_pitch) v[k].z = p4[2];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < kChunkSlots; ++k)
                if (slot[k] >= 0) reinterpret_cast<uint4*>(&tile[0][0])[slot[k]] = v[k];
        } else {
            uint32_t v[kSlots];
            int slot[kSlots];
#pragma unroll
            for (int k = 0; k < kSlots; ++k) {
                const int i = tid + 256 * k;
                const int r = i / kSrcWords, w = i - r * kSrcWords;
                slot[k] = (r < nrows && w < nwords) ? i : -1;
                v[k] = 0;
                if (slot[k] >= 0) {
                    const int gx = sx_lo + 4 * w;
                    const uint8_t* p = s + (size_t)(sy_lo + r) * src_pitch + gx;
                    if (gx + 4 <= src_pitch) v[k] = *reinterpret_cast<const uint32_t*>(p);
                    else   // last partial word of an unpadded row (level 0 with stride == cols): byte-wise, never past the row
                        for (int b = 0; b < 4 && gx + b < src_pitch; ++b) v[k] |= (uint32_t)p[b] << (8 * b);
                }
            }
#pragma unroll
            for (int k = 0; k < kSlots; ++k)
                if (slot[k] >= 0) (&tile[0][0])[slot[k]] = v[k];
        }
        __syncthreads();
        const uint8_t* tb = reinterpret_cast<const uint8_t*>(&tile[0][0]);
        // ---- horizontal pass: hrow[r][x] = (S[r][o0]*a0 + S[r][o1]*a1) >> 4 (<= 32640: exact in 16 bits) for every staged source
        // row and tile column. The ~1.7 output rows that share a source row reuse it; a thread owns a column pair, so its two
        // x-taps are loaded once.
        {
            const int a_o0 = ta.o0 - sx_lo, a_o1 = ta.o1 - sx_lo, b_o0 = tbp.o0 - sx_lo, b_o1 = tbp.o1 - sx_lo;
            // pointers stepped by four rows: indexed by r, hipcc multiplied with v_mul_lo_u32 (quarter rate) twice per unrolled iteration
            const uint8_t* S = tb + q * (kSrcWords * 4);
            uint32_t* H = &hrow[q][xp];
            for (int r = q; r < nrows; r += 4, S += 4 * (kSrcWords * 4), H += 4 * (kTileW / 2)) {
                const uint32_t ha = (uint32_t)(S[a_o0] * ta.a0 + S[a_o1] * ta.a1) >> 4;
                const uint32_t hb = (uint32_t)(S[b_o0] * tbp.a0 + S[b_o1] * tbp.a1) >> 4;
                *H = ha | (hb << 16);
            }
        }
        __syncthreads();
        // ---- vertical pass + store: 4-pixel groups (32 per row), one aligned u32 store each
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            const int idx = tid + g * 256;
            const int y = y0 + (idx >> 5), xg = (idx & 31) * 4, x4 = x0 + xg;
            if (y >= drows || x4 >= dcols) continue;
            const ResizeTap ty = tys[g];
            const uint2 h0 = *reinterpret_cast<const uint2*>(&hrow[ty.o0 - sy_lo][xg >> 1]);
            const uint2 h1 = *reinterpret_cast<const uint2*>(&hrow[ty.o1 - sy_lo][xg >> 1]);
            // ((b0 * r0) >> 16) + ((b1 * r1) >> 16): both factors shifted left by 8 make it the HIGH half of a 24 x 24-bit product
            // (v_mul_hi_u32_u24, full rate); v_perm_b32 extracts a 16-bit half and shifts it in one go. b0 + b1 = 2048 and r <= 32640, so the sum
            // is <= 1020 and (sum + 2) >> 2 <= 255: OpenCV's saturate_cast never clamps here.
            const uint32_t b0 = (uint32_t)ty.a0 << 8, b1 = (uint32_t)ty.a1 << 8;
            constexpr uint32_t kLo = 0x0c01000cu, kHi = 0x0c03020cu;   // (half << 8) as a 32-bit value
            auto vpass = [&](uint32_t w0, uint32_t w1, uint32_t sel) -> uint32_t {
                const uint32_t r0 = __builtin_amdgcn_perm(w0, w0, sel), r1 = __builtin_amdgcn_perm(w1, w1, sel);
                return (mulhi_u24(b0, r0) + mulhi_u24(b1, r1) + 2u) >> 2;
            };
            const uint32_t out = vpass(h0.x, h1.x, kLo) | (vpass(h0.x, h1.x, kHi) << 8) | (vpass(h0.y, h1.y, kLo) << 16) |
                                 (vpass(h0.y, h1.y, kHi) << 24);
            *reinterpret_cast<uint32_t*>(d + (size_t)y * dst_pitch + x4) = out;
        }
    } else {
        // generic fallback (scale factors far from 1.2 whose source rectangle does not fit the LDS tile): gather from global
#pragma unroll 1
        for (int g = 0; g < kGroups; ++g) {
            const int idx = tid + g * 256;
            const int y = y0 + (idx >> 5), x4 = x0 + (idx & 31) * 4;
            if (y >= drows || x4 >= dcols) continue;
            const ResizeTap ty = yt[y];
            const uint8_t* S0 = s + (size_t)ty.o0 * src_pitch;
            const uint8_t* S1 = s + (size_t)ty.o1 * src_pitch;
            const int b0 = ty.a0, b1 = ty.a1;
            uint32_t out = 0;
            for (int i = 0; i < 4; ++i) {
                const int x = x4 + i;
                if (x < dcols) {
                    const ResizeTap tx = xt[x];
                    const int r0 = S0[tx.o0] * tx.a0 + S0[tx.o1] * tx.a1;
                    const int r1 = S1[tx.o0] * tx.a0 + S1[tx.o1] * tx.a1;
                    int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                    v = v < 0 ? 0 : (v > 255 ? 255 : v);
                    out |= (uint32_t)v << (8 * i);
                }
            }
            *reinterpret_cast<uint32_t*>(d + (size_t)y * dst_pitch + x4) = out;
        }
    }
    (void)srows;
    (void)scols;
}

// ================================================================================================================================
// v4 (round 3): the same two fixed-point passes with about half the vector instructions (~33 lane-ops per output pixel
// for ~12 of arithmetic).
//   * horizontal pass: the four source bytes of a thread's column pair lie inside two aligned words (checked per level on the host,
//     LevelGeo::resize_hwin_ok; the v3 kernel stays as the path for geometries where they do not): ONE ds_read2_b32 instead of four
//     ds_read_u8, one v_perm_b32 per column to pair (S[o0], S[o1]) as 2 x u16, one v_dot2_u32_u16 against (16 a0, 16 a1) -- the
//     factor 16 makes h >> 4 the byte-aligned middle of the result, so a third v_perm_b32 packs both columns' 16-bit values;
//   * vertical pass: v_mul_u32_u24 / v_add_u32 / v_lshrrev_b32 in their SDWA forms read the 16-bit halves and write the result byte
//     in place: 5 instructions per pixel (v3: 2 v_perm + 2 v_mul_hi + 3 + 1.75 to pack), same integers:
//         out = (((b0 * r0) >> 16) + ((b1 * r1) >> 16) + 2) >> 2,   r = h >> 4,   h = S[o0] * a0 + S[o1] * a1;
//   * frame / tile indices from integer reciprocals on the scalar unit.
__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_dot2_u32_u16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// b * (low / high 16 bits of w), b < 2^24
__device__ __forceinline__ uint32_t mul_lo16(uint32_t b, uint32_t w) {
    uint32_t d;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(d) : "v"(b), "v"(w));
    return d;
}
__device__ __forceinline__ uint32_t mul_hi16(uint32_t b, uint32_t w) {
    uint32_t d;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d) : "v"(b), "v"(w));
    return d;
}
// (p >> 16) + (q >> 16)
__device__ __forceinline__ uint32_t add_hi16(uint32_t p, uint32_t q) {
    uint32_t d;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(d) : "v"(p), "v"(q));
    return d;
}
// bytes (t0 >> 2, t1 >> 2, t2 >> 2, t3 >> 2), every t < 1024. The byte writes preserve the rest of the destination, i.e. read it: one
// wait state between them (partial-dword write followed by a read of the same register).
__device__ __forceinline__ uint32_t pack_shr2(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t two) {
    uint32_t d;
    asm volatile(
        "v_lshrrev_b32_sdwa %0, %5, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
        "s_nop 0\n"
        "v_lshrrev_b32_sdwa %0, %5, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
        "s_nop 0\n"
        "v_lshrrev_b32_sdwa %0, %5, %3 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
        "s_nop 0\n"
        "v_lshrrev_b32_sdwa %0, %5, %4 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
        "s_nop 0\n"
        : "=&v"(d)
        : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(two));
    return d;
}
__device__ __forceinline__ uint32_t udiv_magic(uint32_t n, uint32_t d, uint32_t magic) { return d == 1 ? n : __umulhi(n, magic); }

__global__ __launch_bounds__(256) void k_resize_linear_u8(const uint8_t* __restrict__ src, size_t src_frame_stride, int src_pitch,
                                                         uint8_t* __restrict__ dst, size_t dst_frame_stride, int dst_pitch, int drows,
                                                         int dcols, const ResizeTap* __restrict__ xt, const ResizeTap* __restrict__ yt,
                                                         int tiles_x, int tiles_frame, int batch, uint32_t tiles_x_magic,
                                                         uint32_t tiles_frame_magic) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[kSrcRows * kSrcWords + 4];   // + 4: the pair read of a row's last word
    __shared__ __attribute__((aligned(8))) uint32_t hrow[kSrcRows][kTileW / 2];        // horizontal pass, two u16 per word
    const int tid = threadIdx.x;
    const int per_xcd = gridDim.x >> 3;
    const int tile_id = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);   // XCD-aware order, see v3
    if (tile_id >= tiles_frame * batch) return;
    const int frame = (int)udiv_magic((uint32_t)tile_id, (uint32_t)tiles_frame, tiles_frame_magic);
    const int trem = tile_id - frame * tiles_frame;
    const int tyi = (int)udiv_magic((uint32_t)trem, (uint32_t)tiles_x, tiles_x_magic), txi = trem - tyi * tiles_x;
    const int x0 = txi * kTileW, y0 = tyi * kTileH;
    const int x1 = min(x0 + kTileW, dcols) - 1, y1 = min(y0 + kTileH, drows) - 1;   // last output pixel of the tile
    const uint8_t* s = src + (size_t)frame * src_frame_stride;
    uint8_t* d = dst + (size_t)frame * dst_frame_stride;
    const int sx_lo = xt[x0].o0 & ~15, sx_hi = xt[x1].o1;
    const int sy_lo = yt[y0].o0, sy_hi = yt[y1].o1;
    const int nwords = (sx_hi - sx_lo) / 4 + 1, nrows = sy_hi - sy_lo + 1;
    // the launcher chose this kernel only for levels whose source rectangles fit (scale <= 1.25, 16-byte aligned rows)
    const int xp = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xa = min(x0 + 2 * xp, dcols - 1), xb = min(x0 + 2 * xp + 1, dcols - 1);
    const ResizeTap ta = xt[xa], tbp = xt[xb];
    ResizeTap tys[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) tys[g] = yt[min(y0 + 8 * g + (tid >> 5), drows - 1)];
    {
        uint4 v[kChunkSlots];
        int slot[kChunkSlots];
        const uint8_t* const org = s + (size_t)sy_lo * src_pitch + sx_lo;
#pragma unroll
        for (int k = 0; k < kChunkSlots; ++k) {
            const int i = tid + 256 * k;
            const int r = (i * 0x1556) >> 16, c = i - r * kRowChunks;   // i / 12 for i < 768
            static_assert(kRowChunks == 12 && kChunkSlots * 256 <= 768, "chunk index split");
            slot[k] = (r < nrows && 4 * c < nwords) ? i : -1;
            v[k] = uint4{0u, 0u, 0u, 0u};
            if (slot[k] >= 0) {
                const int gx = sx_lo + 16 * c;
                const uint8_t* p = org + (size_t)r * src_pitch + 16 * c;
                if (gx + 16 <= src_pitch) {
                    v[k] = *reinterpret_cast<const uint4*>(p);
                } else {   // row tail
                    const uint32_t* p4 = reinterpret_cast<const uint32_t*>(p);
                    if (gx + 4 <= src_pitch) v[k].x = p4[0];
                    if (gx + 8 <= src_pitch) v[k].y = p4[1];
                    if (gx + 12 <= src_pitch) v[k].z = p4[2];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kChunkSlots; ++k)
            if (slot[k] >= 0) reinterpret_cast<uint4*>(&tile[0])[slot[k]] = v[k];
    }
    __syncthreads();
    // ---- horizontal pass
    {
        const int a_o0 = ta.o0 - sx_lo, wa = a_o0 >> 2, base = 4 * wa;
        const uint32_t sa0 = (uint32_t)(a_o0 - base) & 7u, sa1 = (uint32_t)(ta.o1 - sx_lo - base) & 7u;
        const uint32_t sb0 = (uint32_t)(tbp.o0 - sx_lo - base) & 7u, sb1 = (uint32_t)(tbp.o1 - sx_lo - base) & 7u;
        const uint32_t sel_a = sa0 | 0x0c00u | (sa1 << 16) | 0x0c000000u, sel_b = sb0 | 0x0c00u | (sb1 << 16) | 0x0c000000u;
        const uint32_t ca = ((uint32_t)(uint16_t)ta.a0 << 4) | ((uint32_t)(uint16_t)ta.a1 << 20);
        const uint32_t cb = ((uint32_t)(uint16_t)tbp.a0 << 4) | ((uint32_t)(uint16_t)tbp.a1 << 20);
        // wave q owns source rows q, q + 4, ...: eleven fixed steps cover all 44 tile rows (rows >= nrows hold stale bytes and produce
        // values nobody reads), so the loop is straight-line code with immediate LDS offsets and every read in flight before the first use
        const uint32_t* T = &tile[q * kSrcWords + wa];
        uint32_t* H = &hrow[q][xp];
        static_assert(kSrcRows == 44, "eleven steps of four rows");
        uint32_t lo[11], hi[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            lo[k] = T[4 * k * kSrcWords];
            hi[k] = T[4 * k * kSrcWords + 1];
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const uint32_t ha = dot2_u16(__builtin_amdgcn_perm(hi[k], lo[k], sel_a), ca);   // 16 * (S[o0] * a0 + S[o1] * a1) < 2^23
            const uint32_t hb = dot2_u16(__builtin_amdgcn_perm(hi[k], lo[k], sel_b), cb);
            H[4 * k * (kTileW / 2)] = __builtin_amdgcn_perm(hb, ha, 0x06050201u);            // (ha >> 8) | ((hb >> 8) << 16)
        }
    }
    __syncthreads();
    // ---- vertical pass + store: 4-pixel groups (32 per row), one aligned u32 store each
    const uint32_t two = 2u;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        const int idx = tid + g * 256;
        const int y = y0 + (idx >> 5), xg = (idx & 31) * 4, x4 = x0 + xg;
        if (y >= drows || x4 >= dcols) continue;
        const ResizeTap ty = tys[g];
        const uint2 h0 = *reinterpret_cast<const uint2*>(&hrow[ty.o0 - sy_lo][xg >> 1]);
        const uint2 h1 = *reinterpret_cast<const uint2*>(&hrow[ty.o1 - sy_lo][xg >> 1]);
        const uint32_t b0 = (uint32_t)(uint16_t)ty.a0, b1 = (uint32_t)(uint16_t)ty.a1;
        const uint32_t t0 = add_hi16(mul_lo16(b0, h0.x), mul_lo16(b1, h1.x)) + 2u;
        const uint32_t t1 = add_hi16(mul_hi16(b0, h0.x), mul_hi16(b1, h1.x)) + 2u;
        const uint32_t t2 = add_hi16(mul_lo16(b0, h0.y), mul_lo16(b1, h1.y)) + 2u;
        const uint32_t t3 = add_hi16(mul_hi16(b0, h0.y), mul_hi16(b1, h1.y)) + 2u;
        *reinterpret_cast<uint32_t*>(d + (size_t)y * dst_pitch + x4) = pack_shr2(t0, t1, t2, t3, two);
    }
}

hipError_t launch_resize(const uint8_t* src, size_t src_frame_stride, int src_pitch, int srows, int scols, uint8_t* dst,
                         size_t dst_frame_stride, int dst_pitch, int drows, int dcols, const ResizeTap* xt, const ResizeTap* yt,
                         int batch, hipStream_t s, int hwin_ok) {
    const int tiles_x = (dcols + kTileW - 1) / kTileW, tiles_y = (drows + kTileH - 1) / kTileH;
    const int tiles_frame = tiles_x * tiles_y;
    dim3 grid(((tiles_frame * batch + 7) / 8) * 8);
    // v4 needs: the host-checked tap windows (hwin_ok also says every tile's source rectangle fits the LDS tile), 16-byte aligned source
    // rows, and tile_id * tiles_frame < 2^32 for the multiply-high divisions
    const bool v4 = hwin_ok && ((src_pitch & 15) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) &&
                    ((src_frame_stride & 15) == 0) && (uint64_t)tiles_frame * (uint64_t)tiles_frame * (uint64_t)batch < (1ull << 32);
    auto magic = [](int d) { return d > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d) : 0u; };
    if (v4)
        hipLaunchKernelGGL(k_resize_linear_u8, grid, dim3(256), 0, s, src, src_frame_stride, src_pitch, dst, dst_frame_stride, dst_pitch,
                           drows, dcols, xt, yt, tiles_x, tiles_frame, batch, magic(tiles_x), magic(tiles_frame));
    else
        hipLaunchKernelGGL(k_resize_linear_u8_generic, grid, dim3(256), 0, s, src, src_frame_stride, src_pitch, srows, scols, dst,
                           dst_frame_stride, dst_pitch, drows, dcols, xt, yt, tiles_x, tiles_y, batch, 1.0f / (float)tiles_x,
                           1.0f / (float)tiles_frame);
    return hipGetLastError();
}

// ================================================================================================================================
// k_resize_pair_u8 (round 6): TWO levels per launch -- level l + 1 AND level l + 2 from a staged rectangle of level l, so the middle level is
// written once and never read back by the pyramid (the seven per-level launches write every level and read it again:
// 2.72 GB of traffic per 256 1080p frames against 2.08 GB for three pairs + one single level). A workgroup owns one 128 x 32 tile of the UPPER
// level (the v4 tile). Stage A computes the region of the MIDDLE level that tile's taps touch (<= 160 x 44, starting on a 4-pixel boundary) from
// the lower level's rectangle in LDS (<= 224 x 52 bytes) with v4's two passes; the region lands in LDS exactly where v4 would have staged it
// (same pitch, same 16-byte origin), and the part of it this tile OWNS -- from its own first column / row up to the next tile's -- goes to the
// middle level's plane: ownership boundaries are multiples of four pixels in x, so every stored word is whole, the regions of neighbouring
// tiles overlap by a halo of 1-6 columns / 1-2 rows that both compute (identical integers) and only one stores. Stage B is v4's code on that
// LDS tile. Same integers as two v4 launches (tests: planes byte-equal; the oracle's resize is the reference for both).
// LDS: bufA = lower rectangle, later the middle region; bufH = stage A's horizontal pass, later stage B's: 28.3 KB, five workgroups per CU.
// The regions (PairSpan per tile column / tile row) are computed on the host from the tap tables (orb_api.hip build_pair_plan), which also checks
// that everything fits (pair_ok); a level pair that does not fit runs as two v4 / generic launches.
constexpr int kPairSrcWords = 56;     // 224-byte pitch of the lower rectangle: 160 middle columns x 1.25 + 2 + 15 <= 217
constexpr int kPairSrcRows = 52;      // 44 middle rows x 1.2 + 2 (host-checked)
constexpr int kPairRowChunks = kPairSrcWords / 4;   // 52 rows x 14 sixteen-byte chunks
constexpr int kPairGroups = 40;       // 4-pixel groups per row of the middle region (160 columns)

struct HTaps {
    uint32_t sel_a, sel_b, ca, cb;
    int wa;
};
// selectors and coefficients of v4's horizontal pass for the column pair (ta, tb), source offsets relative to `lo` (a multiple of 4)
__device__ __forceinline__ HTaps make_htaps(const ResizeTap ta, const ResizeTap tb, int lo) {
    HTaps t;
    const int a_o0 = ta.o0 - lo;
    t.wa = a_o0 >> 2;
    const int base = 4 * t.wa;
    const uint32_t sa0 = (uint32_t)(a_o0 - base) & 7u, sa1 = (uint32_t)(ta.o1 - lo - base) & 7u;
    const uint32_t sb0 = (uint32_t)(tb.o0 - lo - base) & 7u, sb1 = (uint32_t)(tb.o1 - lo - base) & 7u;
    t.sel_a = sa0 | 0x0c00u | (sa1 << 16) | 0x0c000000u;
    t.sel_b = sb0 | 0x0c00u | (sb1 << 16) | 0x0c000000u;
    t.ca = ((uint32_t)(uint16_t)ta.a0 << 4) | ((uint32_t)(uint16_t)ta.a1 << 20);
    t.cb = ((uint32_t)(uint16_t)tb.a0 << 4) | ((uint32_t)(uint16_t)tb.a1 << 20);
    return t;
}
__device__ __forceinline__ uint32_t hpair(uint32_t lo, uint32_t hi, const HTaps& t) {
    const uint32_t ha = dot2_u16(__builtin_amdgcn_perm(hi, lo, t.sel_a), t.ca);   // 16 * (S[o0] * a0 + S[o1] * a1) < 2^23
    const uint32_t hb = dot2_u16(__builtin_amdgcn_perm(hi, lo, t.sel_b), t.cb);
    return __builtin_amdgcn_perm(hb, ha, 0x06050201u);                            // (ha >> 8) | ((hb >> 8) << 16)
}
// four output pixels from the horizontal-pass words of their two source rows
__device__ __forceinline__ uint32_t vgroup(const uint2 h0, const uint2 h1, const ResizeTap ty) {
    const uint32_t b0 = (uint32_t)(uint16_t)ty.a0, b1 = (uint32_t)(uint16_t)ty.a1;
    const uint32_t t0 = add_hi16(mul_lo16(b0, h0.x), mul_lo16(b1, h1.x)) + 2u;
    const uint32_t t1 = add_hi16(mul_hi16(b0, h0.x), mul_hi16(b1, h1.x)) + 2u;
    const uint32_t t2 = add_hi16(mul_lo16(b0, h0.y), mul_lo16(b1, h1.y)) + 2u;
    const uint32_t t3 = add_hi16(mul_hi16(b0, h0.y), mul_hi16(b1, h1.y)) + 2u;
    return pack_shr2(t0, t1, t2, t3, 2u);
}

// coef = b0 | b1 << 16, w = two 16-bit horizontal-pass values: products of the selected halves (SDWA picks both operands' halves: no extraction)
#define OVS_MUL_SEL(name, s0, s1)                                                                                                                   \
    __device__ __forceinline__ uint32_t name(uint32_t coef, uint32_t w) {                                                                           \
        uint32_t d;                                                                                                                                  \
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" s0 " src1_sel:" s1 : "=v"(d) : "v"(coef), "v"(w));        \
        return d;                                                                                                                                    \
    }
OVS_MUL_SEL(mul_c0_w0, "WORD_0", "WORD_0")
OVS_MUL_SEL(mul_c0_w1, "WORD_0", "WORD_1")
OVS_MUL_SEL(mul_c1_w0, "WORD_1", "WORD_0")
OVS_MUL_SEL(mul_c1_w1, "WORD_1", "WORD_1")
#undef OVS_MUL_SEL
// four output pixels: rows h0 (coefficient b0 = low half of coef) and h1 (b1 = high half); the bytes are gathered by shifts and one v_perm_b32
// instead of four partial-register writes (each of which costs a wait state): t <= 1022, so (t0 | t1 << 16) >> 2 has t0 >> 2 in byte 0 and
// t1 >> 2 in byte 2
__device__ __forceinline__ uint32_t vgroup_packed(const uint2 h0, const uint2 h1, const uint32_t coef) {
    const uint32_t u0 = add_hi16(mul_c0_w0(coef, h0.x), mul_c1_w0(coef, h1.x));
    const uint32_t u1 = add_hi16(mul_c0_w1(coef, h0.x), mul_c1_w1(coef, h1.x));
    const uint32_t u2 = add_hi16(mul_c0_w0(coef, h0.y), mul_c1_w0(coef, h1.y));
    const uint32_t u3 = add_hi16(mul_c0_w1(coef, h0.y), mul_c1_w1(coef, h1.y));
    const uint32_t p01 = (((u1 << 16) | u0) + 0x00020002u) >> 2, p23 = (((u3 << 16) | u2) + 0x00020002u) >> 2;
    return __builtin_amdgcn_perm(p23, p01, 0x06040200u);
}

// One tile's scalars (workgroup-uniform) ...
struct PairTile {
    int x0, y0, sx_lo, sy_lo, bx0, nG, nrB, own_g, own_r, ax_lo, nwordsA, ay_lo, nrA, cx_hi;
    const uint8_t* s;
    uint8_t *d1, *d2;
};
// ... and what a thread fetches for it ahead of time: its three staging chunks, its three column-pair records, its y-tap record
struct PairFetch {
    uint4 v[3];
    uint4 r1, r2, r3;
    uint32_t w1, w2, w3;
    ResizeTap ty;
};

__global__ __launch_bounds__(256) void k_resize_pair_u8(const uint8_t* __restrict__ src, size_t src_frame_stride, int src_pitch,
                                                       uint8_t* __restrict__ dst1, int pitch1, int rows1, int cols1, uint8_t* __restrict__ dst2,
                                                       int pitch2, int rows2, int cols2, size_t dst_frame_stride,
                                                       const ResizeTap* __restrict__ yt1, const ResizeTap* __restrict__ yt2,
                                                       const HTapRec* __restrict__ ht1, const HTapRec* __restrict__ ht2,
                                                       const PairSpan* __restrict__ plan, int tiles_x, int tiles_frame, int batch,
                                                       uint32_t tiles_x_magic, uint32_t tiles_frame_magic) {
    __shared__ __attribute__((aligned(16))) uint32_t bufA[kPairSrcRows * kPairSrcWords + 4];
    __shared__ __attribute__((aligned(8))) uint32_t bufH[kPairSrcRows * 2 * kPairGroups];
    // y taps as LDS records {byte offset of source row o0 in bufH | that of o1 << 16, a0 | a1 << 16}: the vertical passes read them with immediate offsets
    __shared__ __attribute__((aligned(8))) uint2 s_tyA[48], s_tyB[kTileH];
    static_assert(kPairSrcRows * kPairSrcWords >= kSrcRows * kSrcWords && kPairSrcRows * 2 * kPairGroups >= kSrcRows * (kTileW / 2), "stage B reuses both buffers");
    const int tid = threadIdx.x;
    const int xp = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: XCD k owns the k-th contiguous eighth of the tile sequence. (A form in which a workgroup took 2 - 8 consecutive tiles and fetched
    // tile i + 1 into registers under tile i's arithmetic was measured: 121 VGPRs, 0.70 / 0.75 / 0.78 ms at 2 / 4 / 8 tiles against 0.705 for one tile
    // per workgroup -- the kernel waits for its vector ALU, not for memory: profiles/r06z_pyr_pair.txt.)
    const int per_xcd = gridDim.x >> 3;
    const int first = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (first >= tiles_frame * batch) return;
    // staging split: thread = (16-byte chunk column c, row phase sph), rows sph, sph + 18, sph + 36
    const int sph = (tid * 4682) >> 16, sc = tid - sph * kPairRowChunks;   // tid / 14, tid % 14
    static_assert(kPairRowChunks == 14 && 18 * kPairRowChunks <= 256 && 3 * 18 >= kPairSrcRows, "staging split");
    const int p2 = 64 + (tid & 15);

    auto decode = [&](int tile_id) -> PairTile {
        PairTile t;
        const int frame = (int)udiv_magic((uint32_t)tile_id, (uint32_t)tiles_frame, tiles_frame_magic);
        const int trem = tile_id - frame * tiles_frame;
        const int tyi = (int)udiv_magic((uint32_t)trem, (uint32_t)tiles_x, tiles_x_magic), txi = trem - tyi * tiles_x;
        const PairSpan PX = plan[txi], PY = plan[tiles_x + tyi];
        t.x0 = txi * kTileW;
        t.y0 = tyi * kTileH;
        t.s = src + (size_t)frame * src_frame_stride;
        t.d1 = dst1 + (size_t)frame * dst_frame_stride;
        t.d2 = dst2 + (size_t)frame * dst_frame_stride;
        // stage B's tile origin (v4's sx_lo, sy_lo), the middle region [bx0, bx0 + 4 nG) x [sy_lo, sy_lo + nrB), the lower rectangle
        t.sx_lo = PX.s_lo; t.sy_lo = PY.s_lo; t.bx0 = PX.b0; t.nG = PX.n; t.nrB = PY.n; t.own_g = PX.own; t.own_r = PY.own;
        t.ax_lo = PX.a0; t.nwordsA = PX.an; t.ay_lo = PY.a0; t.nrA = PY.an;
        t.cx_hi = min(t.bx0 + 4 * t.nG - 1, cols1 - 1);   // last middle column that exists
        return t;
    };
    auto fetch = [&](const PairTile& t) -> PairFetch {
        PairFetch f;
        // the x taps of this thread's column pairs (host-built records: selectors, coefficients, first source word). stage A: phase 1 = column pairs
        // 0 .. 63 of the region (lane = pair, wave q takes rows q, q + 4, ..), phase 2 = pairs 64 .. 79 (16 lanes per row group, 16 row groups);
        // stage B: pair xp of the tile. Pairs outside the region take the last pair's record: values nobody reads.
        const int jmax1 = t.cx_hi >> 1;
        const HTapRec* const r1 = ht1 + min((t.bx0 >> 1) + xp, jmax1);
        const HTapRec* const r2 = ht1 + min((t.bx0 >> 1) + p2, jmax1);
        const HTapRec* const r3 = ht2 + min((t.x0 >> 1) + xp, (cols2 - 1) >> 1);
        f.r1 = *reinterpret_cast<const uint4*>(r1);
        f.r2 = *reinterpret_cast<const uint4*>(r2);
        f.r3 = *reinterpret_cast<const uint4*>(r3);
        f.w1 = r1->wa;
        f.w2 = r2->wa;
        f.w3 = r3->wa;
        // the y tap of one row of the middle region (threads 0 .. 47) or of the tile (64 .. 95)
        f.ty = ResizeTap{0, 0, 0, 0};
        if (tid < 48) f.ty = yt1[min(t.sy_lo + tid, rows1 - 1)];
        else if (tid >= 64 && tid < 64 + kTileH) f.ty = yt2[min(t.y0 + tid - 64, rows2 - 1)];
        // the lower rectangle: one address, two strides
        const int gx = t.ax_lo + 16 * sc;
        const bool col_in = sph < 18 && 4 * sc < t.nwordsA, whole = gx + 16 <= src_pitch;
        const uint8_t* p = t.s + (size_t)(t.ay_lo + sph) * src_pitch + gx;
#pragma unroll
        for (int k = 0; k < 3; ++k, p += (size_t)18 * src_pitch) {
            f.v[k] = uint4{0u, 0u, 0u, 0u};
            if (col_in && sph + 18 * k < t.nrA) {
                if (whole) {
                    f.v[k] = *reinterpret_cast<const uint4*>(p);
                } else {   // row tail
                    const uint32_t* p4 = reinterpret_cast<const uint32_t*>(p);
                    if (gx + 4 <= src_pitch) f.v[k].x = p4[0];
                    if (gx + 8 <= src_pitch) f.v[k].y = p4[1];
                    if (gx + 12 <= src_pitch) f.v[k].z = p4[2];
                }
            }
        }
        return f;
    };

    const PairTile T = decode(first);
    const PairFetch F = fetch(T);
    {
        // ---- this tile's fetched data -> LDS
        {
            const bool col_in = sph < 18 && 4 * sc < T.nwordsA;
            uint4* const w = reinterpret_cast<uint4*>(&bufA[0]) + sph * kPairRowChunks + sc;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (col_in && sph + 18 * k < T.nrA) w[k * 18 * kPairRowChunks] = F.v[k];
            if (tid < 48)
                s_tyA[tid] = uint2{(uint32_t)((F.ty.o0 - T.ay_lo) * (8 * kPairGroups)) | ((uint32_t)((F.ty.o1 - T.ay_lo) * (8 * kPairGroups)) << 16),
                                   (uint32_t)(uint16_t)F.ty.a0 | ((uint32_t)(uint16_t)F.ty.a1 << 16)};
            else if (tid >= 64 && tid < 64 + kTileH)
                s_tyB[tid - 64] = uint2{(uint32_t)((F.ty.o0 - T.sy_lo) * (2 * kTileW)) | ((uint32_t)((F.ty.o1 - T.sy_lo) * (2 * kTileW)) << 16),
                                        (uint32_t)(uint16_t)F.ty.a0 | ((uint32_t)(uint16_t)F.ty.a1 << 16)};
        }
        const HTaps t1 = HTaps{F.r1.x, F.r1.y, F.r1.z, F.r1.w, (int)F.w1 - (T.ax_lo >> 2)};
        const HTaps t2 = HTaps{F.r2.x, F.r2.y, F.r2.z, F.r2.w, (int)F.w2 - (T.ax_lo >> 2)};
        const HTaps t3 = HTaps{F.r3.x, F.r3.y, F.r3.z, F.r3.w, (int)F.w3 - (T.sx_lo >> 2)};
        const PairTile& C = T;
        __syncthreads();
        // ---- stage A, horizontal pass: bufH[r][pair] for the 52 rows of the rectangle (rows >= nrA hold stale bytes nobody reads)
        {
            const uint32_t* Ts = &bufA[q * kPairSrcWords + t1.wa];
            uint32_t* H = &bufH[q * (2 * kPairGroups) + xp];
            static_assert(kPairSrcRows == 52, "thirteen steps of four rows, four steps of sixteen");
            uint32_t lo[13], hi[13];
#pragma unroll
            for (int k = 0; k < 13; ++k) {
                lo[k] = Ts[4 * k * kPairSrcWords];
                hi[k] = Ts[4 * k * kPairSrcWords + 1];
            }
#pragma unroll
            for (int k = 0; k < 13; ++k) H[4 * k * (2 * kPairGroups)] = hpair(lo[k], hi[k], t1);
            const int r0 = tid >> 4;
            const uint32_t* T2 = &bufA[r0 * kPairSrcWords + t2.wa];
            uint32_t* H2 = &bufH[r0 * (2 * kPairGroups) + p2];
            uint32_t lo2[4], hi2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = r0 + 16 * k < kPairSrcRows;
                lo2[k] = in ? T2[16 * k * kPairSrcWords] : 0u;
                hi2[k] = in ? T2[16 * k * kPairSrcWords + 1] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (r0 + 16 * k < kPairSrcRows) H2[16 * k * (2 * kPairGroups)] = hpair(lo2[k], hi2[k], t2);
        }
        __syncthreads();
        // ---- stage A, vertical pass: the middle region into bufA (laid out as v4's staged tile: pitch kSrcWords, origin (sx_lo, sy_lo)) and, where
        // this tile owns it, into the middle level's plane. Thread = (4-pixel group grp, row phase ph), rows ph, ph + 6, ..: constant strides.
        {
            const int ph = (tid * 1639) >> 16, grp = tid - ph * kPairGroups;   // tid / 40, tid % 40
            static_assert(kPairGroups == 40 && 6 * kPairGroups <= 256 && 6 * 8 >= kSrcRows, "vertical split of stage A");
            const bool gvalid = ph < 6 && grp < C.nG;
            const int x4 = C.bx0 + 4 * grp;
            const bool own_ok = grp < C.own_g && x4 < cols1;
            uint8_t* gp = C.d1 + (size_t)(C.sy_lo + ph) * pitch1 + x4;
            uint32_t* const tb = &bufA[ph * kSrcWords + ((C.bx0 - C.sx_lo) >> 2) + grp];
            const uint8_t* const hb = reinterpret_cast<const uint8_t*>(&bufH[0]) + 8 * grp;
            const uint2* const ty = &s_tyA[ph];
#pragma unroll
            for (int k = 0; k < 8; ++k, gp += (size_t)6 * pitch1) {
                const int row = ph + 6 * k;
                if (gvalid && row < C.nrB) {
                    const uint2 t = ty[6 * k];
                    const uint2 h0 = *reinterpret_cast<const uint2*>(hb + (t.x & 0xffffu));
                    const uint2 h1 = *reinterpret_cast<const uint2*>(hb + (t.x >> 16));
                    const uint32_t out = vgroup_packed(h0, h1, t.y);
                    tb[6 * k * kSrcWords] = out;
                    if (own_ok && row < C.own_r) *reinterpret_cast<uint32_t*>(gp) = out;
                }
            }
        }
        __syncthreads();
        // ---- stage B = v4 on the LDS tile: horizontal pass
        {
            const uint32_t* Ts = &bufA[q * kSrcWords + t3.wa];
            uint32_t* H = &bufH[q * (kTileW / 2) + xp];
            uint32_t lo[11], hi[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                lo[k] = Ts[4 * k * kSrcWords];
                hi[k] = Ts[4 * k * kSrcWords + 1];
            }
#pragma unroll
            for (int k = 0; k < 11; ++k) H[4 * k * (kTileW / 2)] = hpair(lo[k], hi[k], t3);
        }
        __syncthreads();
        // ---- stage B: vertical pass + store (thread = (group, row phase), rows ph, ph + 8, ..)
        {
            const int xg = tid & 31, ph = tid >> 5, x4 = C.x0 + 4 * xg;
            const bool xok = x4 < cols2;
            uint8_t* gp = C.d2 + (size_t)(C.y0 + ph) * pitch2 + x4;
            const uint8_t* const hb = reinterpret_cast<const uint8_t*>(&bufH[0]) + 8 * xg;
            const uint2* const ty = &s_tyB[ph];
#pragma unroll
            for (int g = 0; g < kGroups; ++g, gp += (size_t)8 * pitch2) {
                if (xok && C.y0 + ph + 8 * g < rows2) {
                    const uint2 t = ty[8 * g];
                    const uint2 h0 = *reinterpret_cast<const uint2*>(hb + (t.x & 0xffffu));
                    const uint2 h1 = *reinterpret_cast<const uint2*>(hb + (t.x >> 16));
                    *reinterpret_cast<uint32_t*>(gp) = vgroup_packed(h0, h1, t.y);
                }
            }
        }
    }
}

hipError_t launch_resize_pair(const uint8_t* src, size_t src_frame_stride, int src_pitch, uint8_t* dst1, int pitch1, int rows1, int cols1,
                              uint8_t* dst2, int pitch2, int rows2, int cols2, size_t dst_frame_stride, const ResizeTap* yt1, const ResizeTap* yt2,
                              const HTapRec* ht1, const HTapRec* ht2, const PairSpan* plan, int batch, hipStream_t s) {
    const int tiles_x = (cols2 + kTileW - 1) / kTileW, tiles_y = (rows2 + kTileH - 1) / kTileH;
    const int tiles_frame = tiles_x * tiles_y;
    dim3 grid(((tiles_frame * batch + 7) / 8) * 8);
    auto magic = [](int d) { return d > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d) : 0u; };
    hipLaunchKernelGGL(k_resize_pair_u8, grid, dim3(256), 0, s, src, src_frame_stride, src_pitch, dst1, pitch1, rows1, cols1, dst2, pitch2, rows2, cols2,
                       dst_frame_stride, yt1, yt2, ht1, ht2, plan, tiles_x, tiles_frame, batch, magic(tiles_x), magic(tiles_frame));
    return hipGetLastError();
}
// the conditions launch_resize_pair adds to those of the plan (pair_ok): what v4 asks of its source and of the tile count
bool resize_pair_launchable(const uint8_t* src, size_t src_frame_stride, int src_pitch, int rows2, int cols2, int batch) {
    const int tiles_x = (cols2 + kTileW - 1) / kTileW, tiles_y = (rows2 + kTileH - 1) / kTileH;
    const uint64_t tiles_frame = (uint64_t)tiles_x * tiles_y;
    return ((src_pitch & 15) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && ((src_frame_stride & 15) == 0) &&
           tiles_frame * tiles_frame * (uint64_t)batch < (1ull << 32);
}

// ================================================================================================================================
// k_pyramid_chain (round 5): ALL levels of ONE frame (or a few) in a single launch -- the tracker's per-frame extract spent 7 x 5.7 us in
// seven DEPENDENT resize launches (plus the gaps between them) for ~1 us of arithmetic each (profiles/r04g_tracked_frame_trace.txt).
// A workgroup owns one tile of every level (the levels are partitioned by the same TX x TY grid, proportionally) and computes the chain
// level 0 -> 1 -> ... -> L-1 for its tile entirely in LDS: the region R_l it computes at level l is its owned tile plus whatever the next
// level's region needs of level l (the tap footprint: a halo of 1-2 pixels per level and side, recomputed redundantly by the neighbours --
// identical integers, and only the OWNED pixels are stored to the level's plane). No workgroup waits for another one, so there are no
// flags, no grid barrier and no ordering between launches: one launch, L - 1 workgroup barriers. The regions (ChainSpan, per level and tile
// column / row) are computed on the host with the tap tables (orb_api.hip build_chain_plan); the arithmetic is the generic kernel's
// (the same integers as v4): h = (S[o0] a0 + S[o1] a1) >> 4 per source row, out = (((b0 h0) >> 16) + ((b1 h1) >> 16) + 2) >> 2.
// Throughput is NOT the point (about 1.2x the pixels, byte-wise LDS reads): the batch path keeps the per-level kernels above.
constexpr int kChainThreads = 1024;
constexpr int kChainRowsPerThread = 8;   // level-0 staging: 16 waves x 8 rows = at most 128 source rows, one row of <= 64 words per wave step

__global__ __launch_bounds__(kChainThreads) void k_pyramid_chain(const uint8_t* __restrict__ img0, size_t frame_stride0, int pitch0,
                                                                uint8_t* __restrict__ pyr, size_t pyr_frame_bytes,
                                                                const FrameGeo* __restrict__ geo, const ResizeTap* __restrict__ taps,
                                                                const ChainSpan* __restrict__ plan, int TX, int TY, int bufA_bytes,
                                                                int bufB_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t chain_smem[];
    uint8_t* const bufA = chain_smem;
    uint8_t* const bufB = chain_smem + bufA_bytes;
    ChainSpan* const s_span = reinterpret_cast<ChainSpan*>(chain_smem + bufA_bytes + bufB_bytes);   // [2 * OVS_MAX_LEVELS]: x spans, then y spans
    int4* const s_lv = reinterpret_cast<int4*>(s_span + 2 * OVS_MAX_LEVELS);                          // [OVS_MAX_LEVELS]: xtab_off, ytab_off, plane_off, pitch
    ResizeTap* const s_tap = reinterpret_cast<ResizeTap*>(s_lv + OVS_MAX_LEVELS);                     // per level l >= 1: W_l x-taps, then H_l y-taps
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = TX * TY;
    const int frame = (int)blockIdx.x / tiles, tile = (int)blockIdx.x - frame * tiles;
    const int tj = tile / TX, ti = tile - tj * TX;
    const int L = geo->num_levels;
    // ---- this tile's spans -> LDS (one global round trip)
    if (tid < 2 * L) {
        const int l = tid < L ? tid : tid - L;
        s_span[tid < L ? l : OVS_MAX_LEVELS + l] = plan[l * (TX + TY) + (tid < L ? ti : TX + tj)];
        if (tid < L) s_lv[l] = int4{(int)geo->lv[l].xtab_off, (int)geo->lv[l].ytab_off, (int)geo->lv[l].plane_off, geo->lv[l].pitch};
    }
    __syncthreads();
    // ---- every tap this tile needs (one per thread and round) and the level-0 source rectangle: all requests first, then the LDS writes
    int n_taps = 0;
    for (int l = 1; l < L; ++l) n_taps += (s_span[l].c1 - s_span[l].c0) + (s_span[OVS_MAX_LEVELS + l].c1 - s_span[OVS_MAX_LEVELS + l].c0);
    const ChainSpan X0 = s_span[0], Y0 = s_span[OVS_MAX_LEVELS];
    const int sp0 = ((X0.c1 - X0.c0) + 3) & ~3, nw0 = sp0 >> 2, H0 = Y0.c1 - Y0.c0;
    const uint8_t* const s0 = img0 + (size_t)frame * frame_stride0;
    const int cols0 = geo->lv[0].cols;   // a row is read up to its last PIXEL, not up to the pitch: a caller's buffer may end at (rows - 1) * stride + cols
    uint32_t v0[kChainRowsPerThread];
#pragma unroll
    for (int k = 0; k < kChainRowsPerThread; ++k) {
        const int r = wave + 16 * k;
        v0[k] = 0;
        if (r < H0 && lane < nw0) {
            const int gx = X0.c0 + 4 * lane;
            const uint8_t* p = s0 + (size_t)(Y0.c0 + r) * pitch0 + gx;
            if (gx + 4 <= cols0) {   // (base, pitch and frame stride are multiples of 4: the launcher's condition)
                v0[k] = *reinterpret_cast<const uint32_t*>(p);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (gx + b < cols0) v0[k] |= (uint32_t)p[b] << (8 * b);
            }
        }
    }
    // taps travel as LDS-relative records: an x-tap's o0 / o1 become byte offsets inside the source region's row, a y-tap's become the byte offsets of
    // its two source rows inside the region (both fit 16 bits: regions are < 64 KB) -- the level loop below then has no address multiplications
    for (int base = 0; base < n_taps; base += kChainThreads) {
        const int t = base + tid;
        int src = -1, cum = 0, sub = 0, mul = 1;
        int px0 = X0.c0, py0 = Y0.c0, ppitch = sp0;   // the source region of level l: origin and row pitch
        for (int l = 1; l < L; ++l) {
            const ChainSpan X = s_span[l], Y = s_span[OVS_MAX_LEVELS + l];
            const int W = X.c1 - X.c0, H = Y.c1 - Y.c0;
            if (t >= cum && t < cum + W + H) {
                const int u = t - cum;
                src = u < W ? s_lv[l].x + X.c0 + u : s_lv[l].y + Y.c0 + (u - W);
                sub = u < W ? px0 : py0;
                mul = u < W ? 1 : ppitch;
            }
            cum += W + H;
            px0 = X.c0;
            py0 = Y.c0;
            ppitch = (W + 3) & ~3;
        }
        if (src >= 0) {
            ResizeTap tp = taps[src];
            tp.o0 = (uint16_t)(((int)tp.o0 - sub) * mul);
            tp.o1 = (uint16_t)(((int)tp.o1 - sub) * mul);
            s_tap[t] = tp;
        }
    }
#pragma unroll
    for (int k = 0; k < kChainRowsPerThread; ++k) {
        const int r = wave + 16 * k;
        if (r < H0 && lane < nw0) reinterpret_cast<uint32_t*>(bufA)[r * nw0 + lane] = v0[k];
    }
    __syncthreads();
    // ---- the chain
    const uint8_t* S = bufA;
    uint8_t* D = bufB;
    int tap_off = 0;
    uint8_t* const pf = pyr + (size_t)frame * pyr_frame_bytes;
    for (int l = 1; l < L; ++l) {
        const ChainSpan X = s_span[l], Y = s_span[OVS_MAX_LEVELS + l];
        const int W = X.c1 - X.c0, H = Y.c1 - Y.c0, dp = (W + 3) & ~3;
        const ResizeTap* const xt = s_tap + tap_off;
        const ResizeTap* const yt = xt + W;
        if (W > 0 && H > 0) {
            const int nchunk = W <= 64 ? 1 : (W <= 128 ? 2 : 3), nrg = W <= 64 ? 16 : (W <= 128 ? 8 : 5);
            const int chunk = W <= 64 ? 0 : (W <= 128 ? (wave & 1) : wave % 3), rg = W <= 64 ? wave : (W <= 128 ? (wave >> 1) : wave / 3);
            const int x = chunk * 64 + lane;
            (void)nchunk;
            if (rg < nrg && x < W) {
                const ResizeTap tx = xt[x];
                const uint8_t* const S0 = S + tx.o0;
                const uint8_t* const S1 = S + tx.o1;
                const uint32_t a0 = (uint32_t)(uint16_t)tx.a0, a1 = (uint32_t)(uint16_t)tx.a1;
                const int gpitch = s_lv[l].w;
                // rows the tile owns, as a range of the thread's own row sequence rg, rg + nrg, ...: no per-pixel ownership test
                const bool own_x = X.c0 + x >= X.o0 && X.c0 + x < X.o1;
                const int oy0 = own_x ? Y.o0 - Y.c0 : H, oy1 = Y.o1 - Y.c0;
                uint8_t* g = pf + (size_t)s_lv[l].z + (size_t)(Y.c0 + rg) * gpitch + X.c0 + x;
                uint8_t* dd = D + rg * dp + x;
                const int dstep = nrg * dp, gstep = nrg * gpitch;
#pragma unroll 4
                for (int y = rg; y < H; y += nrg, dd += dstep, g += gstep) {
                    const ResizeTap ty = yt[y];
                    const uint32_t s00 = S0[ty.o0], s01 = S1[ty.o0], s10 = S0[ty.o1], s11 = S1[ty.o1];
                    const uint32_t h0 = (s00 * a0 + s01 * a1) >> 4, h1 = (s10 * a0 + s11 * a1) >> 4;
                    // a0 + a1 = b0 + b1 = 2048 and h <= 32640: the sum is <= 1020, (sum + 2) >> 2 <= 255 -- OpenCV's saturate_cast never clamps here
                    const uint32_t v = ((((uint32_t)(uint16_t)ty.a0 * h0) >> 16) + (((uint32_t)(uint16_t)ty.a1 * h1) >> 16) + 2u) >> 2;
                    *dd = (uint8_t)v;
                    if (y >= oy0 && y < oy1) *g = (uint8_t)v;
                }
            }
        }
        __syncthreads();
        tap_off += W + H;
        const uint8_t* const t = S;
        S = D;
        D = const_cast<uint8_t*>(t);
    }
}

size_t chain_lds_bytes(int bufA_bytes, int bufB_bytes, int tap_cap) {
    return (size_t)bufA_bytes + bufB_bytes + sizeof(ChainSpan) * 2 * OVS_MAX_LEVELS + sizeof(int4) * OVS_MAX_LEVELS + sizeof(ResizeTap) * (size_t)tap_cap;
}

// plan: L x (TX + TY) spans in device memory (orb_api.hip build_chain_plan, uploaded by ensure_geometry); lds = bufA + bufB + spans + taps
hipError_t launch_pyramid_chain(const uint8_t* img0, size_t frame_stride0, int pitch0, uint8_t* pyr, size_t pyr_frame_bytes, const FrameGeo* d_geo,
                                const ResizeTap* d_taps, const ChainSpan* d_plan, int TX, int TY, int bufA_bytes, int bufB_bytes, int tap_cap,
                                int batch, hipStream_t s) {
    const size_t lds = chain_lds_bytes(bufA_bytes, bufB_bytes, tap_cap);
    if ((reinterpret_cast<uintptr_t>(img0) & 3) || (frame_stride0 & 3) || (pitch0 & 3)) return hipErrorInvalidValue;   // chain_usable() says so first
    dim3 grid((unsigned)(TX * TY * batch));
    hipLaunchKernelGGL(k_pyramid_chain, grid, dim3(kChainThreads), lds, s, img0, frame_stride0, pitch0, pyr, pyr_frame_bytes, d_geo, d_taps, d_plan, TX,
                       TY, bufA_bytes, bufB_bytes);
    return hipGetLastError();
}

}   // namespace ovs
