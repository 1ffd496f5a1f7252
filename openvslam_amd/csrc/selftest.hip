// selftest.hip -- device evaluation of include/ovs_detmath.h, so a test can prove that gfx950 and the host produce the same bits
// for the four libm replacements that decide match pairs (predict_scale_level's log, the equirectangular asin / atan2, the epipolar
// acos). Not on any product path; ovs_detmath_eval is the only entry.
#include <vector>

#include "ovs_common.h"

namespace ovs {

__global__ __launch_bounds__(256) void k_detmath_eval(int fn, const double* __restrict__ a, const double* __restrict__ b,
                                                      double* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (fn) {
        case OVS_DETMATH_LOGF: r = (double)ovs_det_logf((float)a[i]); break;
        case OVS_DETMATH_ASIN: r = ovs_det_asin(a[i]); break;
        case OVS_DETMATH_ACOS: r = ovs_det_acos(a[i]); break;
        case OVS_DETMATH_ATAN2: r = ovs_det_atan2(a[i], b[i]); break;
        case OVS_DETMATH_SINF: r = (double)ovs_det_sinf((float)a[i]); break;
        case OVS_DETMATH_COSF: r = (double)ovs_det_cosf((float)a[i]); break;
        default: r = 0.0; break;
    }
    out[i] = r;
}

}   // namespace ovs

extern "C" ovs_status ovs_detmath_eval(int32_t device, int32_t fn, const double* a, const double* b, double* out, int32_t n) {
    if (n < 0 || !a || !out || fn < OVS_DETMATH_LOGF || fn > OVS_DETMATH_COSF || (fn == OVS_DETMATH_ATAN2 && !b)) return OVS_ERR_INVALID;
    if (ovs_device_count() < 1) return OVS_ERR_NO_DEVICE;
    if (n == 0) return OVS_OK;
    if (hipSetDevice(device) != hipSuccess) return OVS_ERR_HIP;
    double *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
    const size_t bytes = (size_t)n * sizeof(double);
    ovs_status st = OVS_ERR_HIP;
    do {
        if (hipMalloc(&d_a, bytes) != hipSuccess || hipMalloc(&d_o, bytes) != hipSuccess) break;
        if (b && hipMalloc(&d_b, bytes) != hipSuccess) break;
        if (hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice) != hipSuccess) break;
        if (b && hipMemcpy(d_b, b, bytes, hipMemcpyHostToDevice) != hipSuccess) break;
        ovs::k_detmath_eval<<<(n + 255) / 256, 256>>>(fn, d_a, d_b ? d_b : d_a, d_o, n);
        if (hipGetLastError() != hipSuccess) break;
        if (hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost) != hipSuccess) break;
        st = OVS_OK;
    } while (false);
    hipFree(d_a);
    hipFree(d_b);
    hipFree(d_o);
    return st;
}
