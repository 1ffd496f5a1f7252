// ba_linearize.hip -- B1-B3: reprojection residual, analytic 2x3 / 2x6 Jacobians and the normal-equation blocks of local
// bundle adjustment (expected: src/openvslam/optimize/g2o/se3/perspective_reproj_edge.{h,cc}, reproj_edge_wrapper.h; g2o
// BaseBinaryEdge::constructQuadraticForm, RobustKernelHuber, BlockSolver_6_3::buildSystem under
// optimize::local_bundle_adjuster::optimize).
//
// One lane per observation (edge), fp64, every product individually rounded (-ffp-contract=off) in the oracle's order, so the
// per-edge quantities (residual, Jacobians, Hpl block) are bit-identical to the CPU; only the SUMS differ by association:
//   * Hll / bl (3x3 + 3 per landmark): fp64 global atomics -- landmarks are scattered, contention is low;
//   * Hpp / bp (6x6 + 6 per keyframe): edges arrive grouped by keyframe, so a wave whose lanes share one pose reduces the 21
//     upper-triangle terms + 6 gradient terms with cross-lane shuffles and issues ONE atomic per term; mixed waves fall back
//     to per-lane atomics;
//   * Hpl (6x3 per edge) needs no reduction and is written once.
// Bandwidth/atomic-bound (20 MB per 100 k-edge linearisation, ~35 MFLOP): no LDS staging to gain. Across GPUs the edges are
// partitioned by keyframe and only the dense Hll|bl buffer needs a sum (RCCL all-reduce, host layer: openvslam_amd/ba.py).
#include <vector>

#include "ovs_common.h"

namespace ovs {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void k_ba_linearize(const double* __restrict__ poses, const uint8_t* __restrict__ pose_fixed,
                                                     int n_pose, const double* __restrict__ points, int n_pt,
                                                     const ovs_ba_edge* __restrict__ edges, int n_edge, ovs_ba_cam cam,
                                                     double huber_delta, double* __restrict__ Hpp, double* __restrict__ bp,
                                                     double* __restrict__ Hll, double* __restrict__ bl, double* __restrict__ Hpl,
                                                     double* __restrict__ chi2) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool valid = e < n_edge;
    int pose = -1, pt = 0;
    double Jl[2][3] = {}, Jp[2][6] = {};
    double W = 0, r0 = 0, r1 = 0, c2 = 0, rho0 = 0;
    if (valid) {
        const ovs_ba_edge ed = edges[e];
        pose = ed.pose_idx;
        pt = ed.point_idx;
        const double* P = poses + 7 * (size_t)pose;
        const double* X = points + 3 * (size_t)pt;
        const double qx = P[3], qy = P[4], qz = P[5], qw = P[6];
        const double tx2 = 2 * qx, ty2 = 2 * qy, tz2 = 2 * qz;
        const double twx = tx2 * qw, twy = ty2 * qw, twz = tz2 * qw;
        const double txx = tx2 * qx, txy = ty2 * qx, txz = tz2 * qx;
        const double tyy = ty2 * qy, tyz = tz2 * qy, tzz = tz2 * qz;
        const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
        const double X0 = X[0], X1 = X[1], X2 = X[2];
        const double x = R[0][0] * X0 + R[0][1] * X1 + R[0][2] * X2 + P[0];
        const double y = R[1][0] * X0 + R[1][1] * X1 + R[1][2] * X2 + P[1];
        const double z = R[2][0] * X0 + R[2][1] * X1 + R[2][2] * X2 + P[2];
        const double invz = 1.0 / z, invz2 = invz * invz;
        const double e0 = ed.obs_x - (cam.fx * x * invz + cam.cx);
        const double e1 = ed.obs_y - (cam.fy * y * invz + cam.cy);
        const double w = ed.inv_sigma_sq;
        c2 = w * (e0 * e0 + e1 * e1);
        rho0 = c2;
        double rho1 = 1.0;
        const double dsqr = huber_delta * huber_delta;
        if (huber_delta > 0 && c2 > dsqr) {
            const double sq = sqrt(c2);
            rho0 = 2 * sq * huber_delta - dsqr;
            rho1 = huber_delta / sq;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Jl[0][c] = -invz * (cam.fx * R[0][c] - cam.fx * x * invz * R[2][c]);
            Jl[1][c] = -invz * (cam.fy * R[1][c] - cam.fy * y * invz * R[2][c]);
        }
        Jp[0][0] = x * y * invz2 * cam.fx;
        Jp[0][1] = -(1 + x * x * invz2) * cam.fx;
        Jp[0][2] = y * invz * cam.fx;
        Jp[0][3] = -invz * cam.fx;
        Jp[0][4] = 0;
        Jp[0][5] = x * invz2 * cam.fx;
        Jp[1][0] = (1 + y * y * invz2) * cam.fy;
        Jp[1][1] = -x * y * invz2 * cam.fy;
        Jp[1][2] = -x * invz * cam.fy;
        Jp[1][3] = 0;
        Jp[1][4] = -invz * cam.fy;
        Jp[1][5] = y * invz2 * cam.fy;
        W = rho1 * w;
        r0 = -W * e0;
        r1 = -W * e1;
        // landmark block: scattered fp64 atomics
        double* hl = Hll + 9 * (size_t)pt;
        double* gl = bl + 3 * (size_t)pt;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) atomicAdd(&hl[3 * a + b], W * (Jl[0][a] * Jl[0][b] + Jl[1][a] * Jl[1][b]));
            atomicAdd(&gl[a], Jl[0][a] * r0 + Jl[1][a] * r1);
        }
    }
    const bool free_pose = valid && !(pose_fixed && pose_fixed[pose]);
    if (free_pose) {
        double* hpl = Hpl + 18 * (size_t)e;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) hpl[3 * a + b] = W * (Jp[0][a] * Jl[0][b] + Jp[1][a] * Jl[1][b]);
    }
    // chi2: one atomic per wave
    {
        const double s0 = wave_sum_f64(valid ? c2 : 0.0), s1 = wave_sum_f64(valid ? rho0 : 0.0);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&chi2[0], s0);
            atomicAdd(&chi2[1], s1);
        }
    }
    // pose block
    const int p0 = __builtin_amdgcn_readfirstlane(pose);
    const bool uniform = __all(!valid || pose == p0) && p0 >= 0;
    if (uniform) {
        const bool any_free = __any(free_pose);
        if (any_free) {   // wave-uniform: all valid lanes share pose p0, hence the same fixed flag
            double* hp = Hpp + 36 * (size_t)p0;
            double* gp = bp + 6 * (size_t)p0;
            const double m = free_pose ? 1.0 : 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int b = a; b < 6; ++b) {
                    const double s = wave_sum_f64(m * (W * (Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b])));
                    if ((threadIdx.x & 63) == 0) {
                        atomicAdd(&hp[6 * a + b], s);
                        if (b != a) atomicAdd(&hp[6 * b + a], s);
                    }
                }
                const double g = wave_sum_f64(m * (Jp[0][a] * r0 + Jp[1][a] * r1));
                if ((threadIdx.x & 63) == 0) atomicAdd(&gp[a], g);
            }
        }
    } else if (free_pose) {
        double* hp = Hpp + 36 * (size_t)pose;
        double* gp = bp + 6 * (size_t)pose;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = 0; b < 6; ++b) atomicAdd(&hp[6 * a + b], W * (Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b]));
            atomicAdd(&gp[a], Jp[0][a] * r0 + Jp[1][a] * r1);
        }
    }
}

}   // namespace ovs

using namespace ovs;

extern "C" {

ovs_status ovs_ba_linearize_dev(const double* d_poses, const uint8_t* d_pose_fixed, int32_t n_pose, const double* d_points,
                                int32_t n_pt, const ovs_ba_edge* d_edges, int32_t n_edge, const ovs_ba_cam* cam, double huber_delta,
                                double* d_Hpp, double* d_bp, double* d_Hll, double* d_bl, double* d_Hpl, double* d_chi2, void* stream) {
    if (!d_poses || !d_points || !cam || !d_Hpp || !d_bp || !d_Hll || !d_bl || !d_Hpl || !d_chi2 || n_pose < 1 || n_pt < 1 || n_edge < 0 ||
        (n_edge > 0 && !d_edges))
        return OVS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    OVS_HIP_TRY(hipMemsetAsync(d_Hpp, 0, sizeof(double) * 36 * (size_t)n_pose, s));
    OVS_HIP_TRY(hipMemsetAsync(d_bp, 0, sizeof(double) * 6 * (size_t)n_pose, s));
    OVS_HIP_TRY(hipMemsetAsync(d_Hll, 0, sizeof(double) * 9 * (size_t)n_pt, s));
    OVS_HIP_TRY(hipMemsetAsync(d_bl, 0, sizeof(double) * 3 * (size_t)n_pt, s));
    OVS_HIP_TRY(hipMemsetAsync(d_chi2, 0, sizeof(double) * 2, s));
    if (n_edge == 0) return OVS_OK;
    OVS_HIP_TRY(hipMemsetAsync(d_Hpl, 0, sizeof(double) * 18 * (size_t)n_edge, s));   // blocks of fixed poses stay zero
    hipLaunchKernelGGL(k_ba_linearize, dim3((n_edge + 255) / 256), dim3(256), 0, s, d_poses, d_pose_fixed, n_pose, d_points, n_pt, d_edges,
                       n_edge, *cam, huber_delta, d_Hpp, d_bp, d_Hll, d_bl, d_Hpl, d_chi2);
    OVS_HIP_TRY(hipGetLastError());
    return OVS_OK;
}

ovs_status ovs_ba_linearize(int32_t device, const double* poses, const uint8_t* pose_fixed, int32_t n_pose, const double* points,
                            int32_t n_pt, const ovs_ba_edge* edges, int32_t n_edge, const ovs_ba_cam* cam, double huber_delta,
                            double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* chi2) {
    if (!poses || !points || !cam || !Hpp || !bp || !Hll || !bl || !Hpl || !chi2 || n_pose < 1 || n_pt < 1 || n_edge < 0 || (n_edge > 0 && !edges))
        return OVS_ERR_INVALID;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    OVS_HIP_TRY(hipSetDevice(device));
    const size_t sz_pose = sizeof(double) * 7 * n_pose, sz_pt = sizeof(double) * 3 * n_pt, sz_e = sizeof(ovs_ba_edge) * (size_t)std::max(n_edge, 1);
    const size_t out_doubles = (size_t)36 * n_pose + 6 * (size_t)n_pose + 9 * (size_t)n_pt + 3 * (size_t)n_pt + 18 * (size_t)std::max(n_edge, 1) + 2;
    unsigned char* d_in = nullptr;
    double* d_out = nullptr;
    const size_t off_pt = (sz_pose + 255) & ~(size_t)255, off_e = (off_pt + sz_pt + 255) & ~(size_t)255, off_f = (off_e + sz_e + 255) & ~(size_t)255;
    OVS_HIP_TRY(hipMalloc(&d_in, off_f + (size_t)n_pose + 256));
    hipError_t er = hipMalloc(&d_out, sizeof(double) * out_doubles);
    if (er != hipSuccess) {
        hipFree(d_in);
        ovs::set_last_error("hipMalloc(out)", er);
        return OVS_ERR_HIP;
    }
    ovs_status st = OVS_OK;
    do {
#define BA_TRY(expr)                              \
    if ((er = (expr)) != hipSuccess) {            \
        ovs::set_last_error(#expr, er);           \
        st = OVS_ERR_HIP;                         \
        break;                                    \
    }
        BA_TRY(hipMemcpy(d_in, poses, sz_pose, hipMemcpyHostToDevice));
        BA_TRY(hipMemcpy(d_in + off_pt, points, sz_pt, hipMemcpyHostToDevice));
        if (n_edge) BA_TRY(hipMemcpy(d_in + off_e, edges, sizeof(ovs_ba_edge) * (size_t)n_edge, hipMemcpyHostToDevice));
        if (pose_fixed) BA_TRY(hipMemcpy(d_in + off_f, pose_fixed, n_pose, hipMemcpyHostToDevice));
        double* dHpp = d_out;
        double* dbp = dHpp + 36 * (size_t)n_pose;
        double* dHll = dbp + 6 * (size_t)n_pose;
        double* dbl = dHll + 9 * (size_t)n_pt;
        double* dHpl = dbl + 3 * (size_t)n_pt;
        double* dchi = dHpl + 18 * (size_t)std::max(n_edge, 1);
        st = ovs_ba_linearize_dev(reinterpret_cast<double*>(d_in), pose_fixed ? d_in + off_f : nullptr, n_pose,
                                  reinterpret_cast<double*>(d_in + off_pt), n_pt, reinterpret_cast<ovs_ba_edge*>(d_in + off_e), n_edge, cam,
                                  huber_delta, dHpp, dbp, dHll, dbl, dHpl, dchi, nullptr);
        if (st != OVS_OK) break;
        BA_TRY(hipDeviceSynchronize());
        BA_TRY(hipMemcpy(Hpp, dHpp, sizeof(double) * 36 * n_pose, hipMemcpyDeviceToHost));
        BA_TRY(hipMemcpy(bp, dbp, sizeof(double) * 6 * n_pose, hipMemcpyDeviceToHost));
        BA_TRY(hipMemcpy(Hll, dHll, sizeof(double) * 9 * n_pt, hipMemcpyDeviceToHost));
        BA_TRY(hipMemcpy(bl, dbl, sizeof(double) * 3 * n_pt, hipMemcpyDeviceToHost));
        if (n_edge) BA_TRY(hipMemcpy(Hpl, dHpl, sizeof(double) * 18 * (size_t)n_edge, hipMemcpyDeviceToHost));
        BA_TRY(hipMemcpy(chi2, dchi, sizeof(double) * 2, hipMemcpyDeviceToHost));
#undef BA_TRY
    } while (0);
    hipFree(d_in);
    hipFree(d_out);
    return st;
}

}   // extern "C"
