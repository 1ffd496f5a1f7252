// orb_pyramid.hip -- A1: orb_extractor::compute_image_pyramid (expected: src/openvslam/feature/orb_extractor.cc), i.e.
// cv::resize(prev_level, level, size, 0, 0, INTER_LINEAR) for CV_8UC1 in OpenCV's 11-bit fixed point.
// Integer-only on the device: the float/double part of OpenCV (source coordinate, coefficient rounding) is evaluated
// once per geometry on the host into ResizeTap tables (orb_api.hip: build_taps) so no device float can change a pixel.
//
// HBM-bound streaming kernel: a thread produces 4 horizontally adjacent output pixels and stores one aligned u32;
// a 64x4 block therefore writes four 256-byte row segments. The two source rows are re-read by ~1.7 output rows
// (scale 1.2), which the per-CU L1 / XCD L2 absorb; algorithmic traffic per level = src plane + dst plane.
#include "ovs_common.h"

namespace ovs {

__global__ __launch_bounds__(256) void k_resize_linear_u8(const uint8_t* __restrict__ src, size_t src_frame_stride, int src_pitch,
                                                         uint8_t* __restrict__ dst, size_t dst_frame_stride, int dst_pitch,
                                                         int drows, int dcols, const ResizeTap* __restrict__ xt,
                                                         const ResizeTap* __restrict__ yt) {
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (y >= drows || x4 >= dcols) return;
    const uint8_t* s = src + (size_t)blockIdx.z * src_frame_stride;
    uint8_t* d = dst + (size_t)blockIdx.z * dst_frame_stride;
    const ResizeTap ty = yt[y];
    const uint8_t* S0 = s + (size_t)ty.o0 * src_pitch;
    const uint8_t* S1 = s + (size_t)ty.o1 * src_pitch;
    const int b0 = ty.a0, b1 = ty.a1;
    uint32_t out = 0;
#pragma unroll
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

hipError_t launch_resize(const uint8_t* src, size_t src_frame_stride, int src_pitch, int srows, int scols, uint8_t* dst,
                         size_t dst_frame_stride, int dst_pitch, int drows, int dcols, const ResizeTap* xt, const ResizeTap* yt,
                         int batch, hipStream_t s) {
    (void)srows;
    (void)scols;
    dim3 block(64, 4);
    dim3 grid((dcols + 255) / 256, (drows + 3) / 4, batch);
    hipLaunchKernelGGL(k_resize_linear_u8, grid, block, 0, s, src, src_frame_stride, src_pitch, dst, dst_frame_stride, dst_pitch,
                       drows, dcols, xt, yt);
    return hipGetLastError();
}

}   // namespace ovs
