// Micro-benchmark (measurement aid, not product code): issue rate of v_mfma_i32_32x32x32_i8 on gfx950 with 1..4 waves per SIMD,
// two accumulate chains per wave, with and without the 0/1 expansion VALU work of k_hamming_near between the MFMAs.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_i8_ubench.hip -o tools/ubench/mfma_i8_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ in, int* __restrict__ out, int iters) {
    v4i b0, b1;
    uint32_t w = in[threadIdx.x];
    for (int i = 0; i < 4; ++i) { b0[i] = in[threadIdx.x + 64 * i]; b1[i] = in[threadIdx.x + 64 * i + 7]; }
    v16i acc0 = {0}, acc1 = {0};
    int sink = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            v4i a;
            if (MODE == 0) a = b1;
            else {
                const int base = (s & 1) * 4;
                a[0] = (int)((w >> base) & 0x01010101u); a[1] = (int)((w >> (base + 1)) & 0x01010101u);
                a[2] = (int)((w >> (base + 2)) & 0x01010101u); a[3] = (int)((w >> (base + 3)) & 0x01010101u);
                w = w * 1664525u + 1013904223u;
            }
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b0, s == 0 && MODE == 2 ? (v16i){0} : acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1, s == 0 && MODE == 2 ? (v16i){0} : acc1, 0, 0, 0);
        }
        if (MODE == 2) {
            int m = acc0[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = max(m, max(acc0[i], acc1[i]));
            if (m >= 0x7FFFFFF0) sink += m;
        }
    }
    int r = sink;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
    uint32_t* in; int* out;
    hipMalloc(&in, 4096 * 4); hipMemset(in, 1, 4096 * 4);
    hipMalloc(&out, 256 * 64 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (int wgs_per_cu = 1; wgs_per_cu <= 4; ++wgs_per_cu) {
            const int grid = 256 * wgs_per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, in, out, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, in, out, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, in, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mfmas = (double)grid * 4 * iters * 16;
            printf("mode %d (%s) waves/SIMD %d: %.3f ms, %.1f TOPS, %.1f cycles(2.4GHz)/MFMA/SIMD\n", mode,
                   mode == 0 ? "mfma only" : mode == 1 ? "mfma + expansion" : "mfma + expansion + max epilogue", wgs_per_cu, ms,
                   mfmas * 65536 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfmas / 1024));
        }
    return 0;
}
