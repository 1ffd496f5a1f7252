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

__global__ __launch_bounds__(256) void k_resize_linear_u8(const uint8_t* __restrict__ src, size_t src_frame_stride, int src_pitch,
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

hipError_t launch_resize(const uint8_t* src, size_t src_frame_stride, int src_pitch, int srows, int scols, uint8_t* dst,
                         size_t dst_frame_stride, int dst_pitch, int drows, int dcols, const ResizeTap* xt, const ResizeTap* yt,
                         int batch, hipStream_t s) {
    const int tiles_x = (dcols + kTileW - 1) / kTileW, tiles_y = (drows + kTileH - 1) / kTileH;
    dim3 grid(((tiles_x * tiles_y * batch + 7) / 8) * 8);
    hipLaunchKernelGGL(k_resize_linear_u8, grid, dim3(256), 0, s, src, src_frame_stride, src_pitch, srows, scols, dst, dst_frame_stride,
                       dst_pitch, drows, dcols, xt, yt, tiles_x, tiles_y, batch, 1.0f / (float)tiles_x, 1.0f / (float)(tiles_x * tiles_y));
    return hipGetLastError();
}

}   // namespace ovs
