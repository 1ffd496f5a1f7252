// hbm_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns this library uses
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE under-reports wide coalesced reads by 2x; other widths are uncalibrated).
// Each kernel moves a KNOWN number of bytes from / to a buffer far larger than the 256 MiB Infinity Cache; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE   and   rocprofv3 --kernel-trace --pmc WRITE_SIZE
// and compare the counters (KiB) with the byte counts printed here. tools/pmc_summary.py prints the per-kernel means.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr size_t kBytes = (size_t)1 << 30;   // 1 GiB per pass

__global__ void calib_read_x4(const uint4* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_read_x1(const uint32_t* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
// the FAST / pyramid staging pattern: a workgroup reads a 70-row x 80-byte tile (five 16-byte loads per row) of a 1920-pitch plane
__global__ void calib_read_tile(const uint8_t* __restrict__ p, int pitch, int tiles_x, uint32_t* out) {
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const uint8_t* base = p + (size_t)ty * 64 * pitch + (size_t)tx * 64;
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < 70 * 5; i += blockDim.x) {
        const int r = i / 5, q = i - r * 5;
        const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)r * pitch + 16 * q);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void calib_write_x4(uint4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, (uint32_t)i);
}
__global__ void calib_write_x1(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

int main() {
    uint8_t* buf;
    uint32_t* out;
    CHECK(hipMalloc(&buf, kBytes + (1 << 20)));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 1, kBytes));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(calib_read_x4, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, kBytes / 16, out);
        hipLaunchKernelGGL(calib_read_x1, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, kBytes / 4, out);
        // tile pattern over a 1920 x (kBytes/1920) plane: tiles of 64 x 64 (+ halo), each reads 70 x 80 bytes
        const int pitch = 1920, rows = (int)(kBytes / pitch), tiles_x = (pitch - 16) / 64, tiles_y = (rows - 6) / 64;
        hipLaunchKernelGGL(calib_read_tile, dim3(tiles_x * tiles_y), dim3(256), 0, 0, (const uint8_t*)buf, pitch, tiles_x, out);
        hipLaunchKernelGGL(calib_write_x4, dim3(grid), dim3(256), 0, 0, (uint4*)buf, kBytes / 16);
        hipLaunchKernelGGL(calib_write_x1, dim3(grid), dim3(256), 0, 0, (uint32_t*)buf, kBytes / 4);
        CHECK(hipDeviceSynchronize());
        if (rep == 0) {
            printf("calib_read_x4  reads  %zu bytes\ncalib_read_x1  reads  %zu bytes\n", kBytes, kBytes);
            printf("calib_read_tile reads %zu bytes issued (%d tiles x 5600; unique plane bytes touched %zu)\n", (size_t)tiles_x * tiles_y * 5600,
                   tiles_x * tiles_y, (size_t)tiles_y * 64 * pitch);
            printf("calib_write_x4 writes %zu bytes\ncalib_write_x1 writes %zu bytes\n", kBytes, kBytes);
        }
    }
    return 0;
}
