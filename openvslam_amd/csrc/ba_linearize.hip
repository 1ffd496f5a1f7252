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

// D = 2: mono_perspective_reproj_edge; D = 3: stereo_perspective_reproj_edge (third residual u_r = u - bf / z).
template <int D>
struct EdgeOf;
template <>
struct EdgeOf<2> {
    typedef ovs_ba_edge type;
};
template <>
struct EdgeOf<3> {
    typedef ovs_ba_edge_stereo type;
};

template <int D>
__device__ __forceinline__ double dotD(const double (&A)[D][6], int a, const double (&B)[D][6], int b) {
    double s = A[0][a] * B[0][b];
#pragma unroll
    for (int k = 1; k < D; ++k) s = s + A[k][a] * B[k][b];
    return s;
}

constexpr int kEdgesPerThread = 2;   // 1, 2, 4 measured within 2 % of each other at 100 k edges: fabric-side fp64 atomics bound the kernel

template <int D>
__global__ __launch_bounds__(256) void k_ba_linearize(const double* __restrict__ poses, const uint8_t* __restrict__ pose_fixed,
                                                     int n_pose, const double* __restrict__ points, int n_pt,
                                                     const typename EdgeOf<D>::type* __restrict__ edges, int n_edge, ovs_ba_cam cam,
                                                     int model, double bf, double huber_delta, double* __restrict__ Hpp, double* __restrict__ bp,
                                                     double* __restrict__ Hll, double* __restrict__ bl, double* __restrict__ Hpl,
                                                     double* __restrict__ chi2) {
    // A workgroup takes kEdgesPerThread * 256 consecutive edges; lane l of wave w handles edges base + 64 (4 k + w) + l. Edges arrive
    // grouped by keyframe, so a wave usually sees ONE pose for all its iterations: its 21 + 6 pose-block terms are accumulated per lane
    // across the iterations and reduced across the wave once (the 27 fp64 wave reductions per 64 edges were a quarter of the kernel).
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    int acc_pose = -1;
    auto flush = [&]() {
        if (acc_pose < 0) return;
        double* hp = Hpp + 36 * (size_t)acc_pose;
        double* gp = bp + 6 * (size_t)acc_pose;
        int t = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = a; b < 6; ++b) {
                const double sv = wave_sum_f64(acc[t]);
                acc[t++] = 0.0;
                if ((threadIdx.x & 63) == 0) {
                    atomicAdd(&hp[6 * a + b], sv);
                    if (b != a) atomicAdd(&hp[6 * b + a], sv);
                }
            }
            const double g = wave_sum_f64(acc[t]);
            acc[t++] = 0.0;
            if ((threadIdx.x & 63) == 0) atomicAdd(&gp[a], g);
        }
        acc_pose = -1;
    };
#pragma unroll 1
    for (int it = 0; it < kEdgesPerThread; ++it) {
    const int e = (blockIdx.x * kEdgesPerThread + it) * 256 + threadIdx.x;
    const bool valid = e < n_edge;
    int pose = -1, pt = 0;
    // Jacobians padded to 6 columns so the landmark (3) and pose (6) blocks share dotD
    double Jl[D][6] = {}, Jp[D][6] = {};
    double W = 0, r[D] = {}, c2 = 0, rho0 = 0;
    if (valid) {
        const typename EdgeOf<D>::type ed = edges[e];
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
        double er[D];
        double ss;
        // equirectangular (model 1, mono only; cam.fx / cam.fy carry cols / rows): quantities shared by the residual and the Jacobians
        double eq_L = 0, eq_rxz = 0;
        const double invz = 1.0 / z, invz2 = invz * invz;
        if (D == 2 && model == 1) {
            eq_L = sqrt((x * x + y * y) + z * z);
            eq_rxz = x * x + z * z;
            const double theta = ovs_det_atan2(x, z);
            const double phi = -ovs_det_asin(y / eq_L);
            er[0] = ed.obs_x - cam.fx * (0.5 + theta / (2.0 * 3.14159265358979323846));
            er[1] = ed.obs_y - cam.fy * (0.5 - phi / 3.14159265358979323846);
            ss = er[0] * er[0] + er[1] * er[1];
        } else {
        const double u = cam.fx * x * invz + cam.cx;
        er[0] = ed.obs_x - u;
        er[1] = ed.obs_y - (cam.fy * y * invz + cam.cy);
        ss = er[0] * er[0] + er[1] * er[1];
        if constexpr (D == 3) {
            er[2] = ed.obs_x_right - (u - bf * invz);
            ss = ss + er[2] * er[2];
        }
        }
        const double w = ed.inv_sigma_sq;
        c2 = w * ss;
        rho0 = c2;
        double rho1 = 1.0;
        const double dsqr = huber_delta * huber_delta;
        if (huber_delta > 0 && c2 > dsqr) {
            const double sq = sqrt(c2);
            rho0 = 2 * sq * huber_delta - dsqr;
            rho1 = huber_delta / sq;
        }
        if (D == 2 && model == 1) {
            // equirectangular_reproj_edge::linearizeOplus: with dp the derivative of pos_c w.r.t. one state component,
            //   d u = (cols / 2 pi) (z dp_x - x dp_z) / (x^2 + z^2),  d v = (rows / pi) (L dp_y - y dL) / (L sqrt(x^2 + z^2)),  dL = pos_c . dp / L,
            // and J = -d(u, v). Columns: rotation (e_k x pos_c), translation (e_k), landmark (R's columns).
            const double a0 = -(cam.fx / (2.0 * 3.14159265358979323846)) * (1.0 / eq_rxz);
            const double a1 = -(cam.fy / 3.14159265358979323846) * (1.0 / (eq_L * sqrt(eq_rxz)));
            auto col = [&](double dx, double dy, double dz, double& j0, double& j1) {
                const double dL = (1.0 / eq_L) * ((x * dx + y * dy) + z * dz);
                j0 = a0 * (z * dx - x * dz);
                j1 = a1 * (eq_L * dy - y * dL);
            };
            col(0.0, -z, y, Jp[0][0], Jp[1][0]);
            col(z, 0.0, -x, Jp[0][1], Jp[1][1]);
            col(-y, x, 0.0, Jp[0][2], Jp[1][2]);
            col(1.0, 0.0, 0.0, Jp[0][3], Jp[1][3]);
            col(0.0, 1.0, 0.0, Jp[0][4], Jp[1][4]);
            col(0.0, 0.0, 1.0, Jp[0][5], Jp[1][5]);
#pragma unroll
            for (int c = 0; c < 3; ++c) col(R[0][c], R[1][c], R[2][c], Jl[0][c], Jl[1][c]);
        } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Jl[0][c] = -invz * (cam.fx * R[0][c] - cam.fx * x * invz * R[2][c]);
            Jl[1][c] = -invz * (cam.fy * R[1][c] - cam.fy * y * invz * R[2][c]);
            if constexpr (D == 3) Jl[2][c] = Jl[0][c] - bf * R[2][c] * invz2;
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
        if constexpr (D == 3) {
            Jp[2][0] = Jp[0][0] - bf * y * invz2;
            Jp[2][1] = Jp[0][1] + bf * x * invz2;
            Jp[2][2] = Jp[0][2];
            Jp[2][3] = Jp[0][3];
            Jp[2][4] = 0;
            Jp[2][5] = Jp[0][5] - bf * invz2;
        }
        }
        W = rho1 * w;
#pragma unroll
        for (int k = 0; k < D; ++k) r[k] = -W * er[k];
        // landmark block: scattered fp64 atomics
        double* hl = Hll + 9 * (size_t)pt;
        double* gl = bl + 3 * (size_t)pt;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) atomicAdd(&hl[3 * a + b], W * dotD<D>(Jl, a, Jl, b));
            double g = Jl[0][a] * r[0];
#pragma unroll
            for (int k = 1; k < D; ++k) g = g + Jl[k][a] * r[k];
            atomicAdd(&gl[a], g);
        }
    }
    const bool free_pose = valid && !(pose_fixed && pose_fixed[pose]);
    if (valid) {   // blocks of fixed poses are zero (written here: no 14 MB memset in front of the kernel)
        double* hpl = Hpl + 18 * (size_t)e;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) hpl[3 * a + b] = free_pose ? W * dotD<D>(Jp, a, Jl, b) : 0.0;
    }
    // chi2: one atomic per wave
    {
        const double s0 = wave_sum_f64(valid ? c2 : 0.0), s1 = wave_sum_f64(valid ? rho0 : 0.0);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&chi2[0], s0);
            atomicAdd(&chi2[1], s1);
        }
    }
    auto grad = [&](int a) {
        double g = Jp[0][a] * r[0];
#pragma unroll
        for (int k = 1; k < D; ++k) g = g + Jp[k][a] * r[k];
        return g;
    };
    // pose block
    const int p0 = __builtin_amdgcn_readfirstlane(pose);
    const bool uniform = __all(!valid || pose == p0) && p0 >= 0;
    if (uniform) {
        if (__any(free_pose)) {   // wave-uniform: all valid lanes share pose p0, hence the same fixed flag
            if (acc_pose != p0) flush();
            acc_pose = p0;
            if (free_pose) {
                int t = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
#pragma unroll
                    for (int b = a; b < 6; ++b) acc[t++] += W * dotD<D>(Jp, a, Jp, b);
                    acc[t++] += grad(a);
                }
            }
        }
    } else {
        flush();
        if (free_pose) {
            double* hp = Hpp + 36 * (size_t)pose;
            double* gp = bp + 6 * (size_t)pose;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int b = 0; b < 6; ++b) atomicAdd(&hp[6 * a + b], W * dotD<D>(Jp, a, Jp, b));
                atomicAdd(&gp[a], grad(a));
            }
        }
    }
    }   // edges of this thread
    flush();
}

// one launch instead of five hipMemsetAsync calls (each costs a launch: they were ~25 us of a 128 us linearisation call)
__global__ __launch_bounds__(256) void k_ba_zero(double* __restrict__ a, size_t na, double* __restrict__ b, size_t nb, double* __restrict__ c,
                                                size_t nc, double* __restrict__ d, size_t nd, double* __restrict__ e, size_t ne) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t k = i; k < na; k += stride) a[k] = 0.0;
    for (size_t k = i; k < nb; k += stride) b[k] = 0.0;
    for (size_t k = i; k < nc; k += stride) c[k] = 0.0;
    for (size_t k = i; k < nd; k += stride) d[k] = 0.0;
    for (size_t k = i; k < ne; k += stride) e[k] = 0.0;
}

}   // namespace ovs

using namespace ovs;

extern "C" {

static ovs_status ba_linearize_dev_impl(int model, const double* d_poses, const uint8_t* d_pose_fixed, int32_t n_pose, const double* d_points,
                                        int32_t n_pt, const ovs_ba_edge* d_edges, int32_t n_edge, const ovs_ba_cam* cam, double huber_delta,
                                        double* d_Hpp, double* d_bp, double* d_Hll, double* d_bl, double* d_Hpl, double* d_chi2, void* stream) {
    if (!d_poses || !d_points || !cam || !d_Hpp || !d_bp || !d_Hll || !d_bl || !d_Hpl || !d_chi2 || n_pose < 1 || n_pt < 1 || n_edge < 0 ||
        (n_edge > 0 && !d_edges))
        return OVS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ba_zero, dim3(128), dim3(256), 0, s, d_Hpp, (size_t)36 * n_pose, d_bp, (size_t)6 * n_pose, d_Hll, (size_t)9 * n_pt, d_bl,
                       (size_t)3 * n_pt, d_chi2, (size_t)2);
    OVS_HIP_TRY(hipGetLastError());
    if (n_edge == 0) return OVS_OK;
    hipLaunchKernelGGL(k_ba_linearize<2>, dim3((n_edge + 256 * kEdgesPerThread - 1) / (256 * kEdgesPerThread)), dim3(256), 0, s, d_poses, d_pose_fixed, n_pose, d_points, n_pt, d_edges,
                       n_edge, *cam, model, 0.0, huber_delta, d_Hpp, d_bp, d_Hll, d_bl, d_Hpl, d_chi2);
    OVS_HIP_TRY(hipGetLastError());
    return OVS_OK;
}

ovs_status ovs_ba_linearize_dev(const double* d_poses, const uint8_t* d_pose_fixed, int32_t n_pose, const double* d_points,
                                int32_t n_pt, const ovs_ba_edge* d_edges, int32_t n_edge, const ovs_ba_cam* cam, double huber_delta,
                                double* d_Hpp, double* d_bp, double* d_Hll, double* d_bl, double* d_Hpl, double* d_chi2, void* stream) {
    return ba_linearize_dev_impl(0, d_poses, d_pose_fixed, n_pose, d_points, n_pt, d_edges, n_edge, cam, huber_delta, d_Hpp, d_bp, d_Hll, d_bl,
                                 d_Hpl, d_chi2, stream);
}

ovs_status ovs_ba_linearize_equirect_dev(const double* d_poses, const uint8_t* d_pose_fixed, int32_t n_pose, const double* d_points,
                                         int32_t n_pt, const ovs_ba_edge* d_edges, int32_t n_edge, int32_t cols, int32_t rows,
                                         double huber_delta, double* d_Hpp, double* d_bp, double* d_Hll, double* d_bl, double* d_Hpl,
                                         double* d_chi2, void* stream) {
    if (cols < 1 || rows < 1) return OVS_ERR_INVALID;
    const ovs_ba_cam c = {(double)cols, (double)rows, 0.0, 0.0};   // the kernel reads cols / rows from fx / fy for model 1
    return ba_linearize_dev_impl(1, d_poses, d_pose_fixed, n_pose, d_points, n_pt, d_edges, n_edge, &c, huber_delta, d_Hpp, d_bp, d_Hll, d_bl,
                                 d_Hpl, d_chi2, stream);
}

ovs_status ovs_ba_linearize_stereo_dev(const double* d_poses, const uint8_t* d_pose_fixed, int32_t n_pose, const double* d_points,
                                       int32_t n_pt, const ovs_ba_edge_stereo* d_edges, int32_t n_edge, const ovs_ba_cam* cam,
                                       double focal_x_baseline, double huber_delta, int32_t accumulate, double* d_Hpp, double* d_bp,
                                       double* d_Hll, double* d_bl, double* d_Hpl, double* d_chi2, void* stream) {
    if (!d_poses || !d_points || !cam || !d_Hpp || !d_bp || !d_Hll || !d_bl || !d_Hpl || !d_chi2 || n_pose < 1 || n_pt < 1 || n_edge < 0 ||
        (n_edge > 0 && !d_edges))
        return OVS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        hipLaunchKernelGGL(k_ba_zero, dim3(128), dim3(256), 0, s, d_Hpp, (size_t)36 * n_pose, d_bp, (size_t)6 * n_pose, d_Hll, (size_t)9 * n_pt,
                           d_bl, (size_t)3 * n_pt, d_chi2, (size_t)2);
        OVS_HIP_TRY(hipGetLastError());
    }
    if (n_edge == 0) return OVS_OK;
    hipLaunchKernelGGL(k_ba_linearize<3>, dim3((n_edge + 256 * kEdgesPerThread - 1) / (256 * kEdgesPerThread)), dim3(256), 0, s, d_poses, d_pose_fixed, n_pose, d_points, n_pt, d_edges,
                       n_edge, *cam, 0, focal_x_baseline, huber_delta, d_Hpp, d_bp, d_Hll, d_bl, d_Hpl, d_chi2);
    OVS_HIP_TRY(hipGetLastError());
    return OVS_OK;
}

static ovs_status ba_linearize_host_impl(int model, int32_t device, const double* poses, const uint8_t* pose_fixed, int32_t n_pose,
                                         const double* points, int32_t n_pt, const ovs_ba_edge* edges, int32_t n_edge, const ovs_ba_cam* cam,
                                         double huber_delta, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* chi2);

ovs_status ovs_ba_linearize(int32_t device, const double* poses, const uint8_t* pose_fixed, int32_t n_pose, const double* points,
                            int32_t n_pt, const ovs_ba_edge* edges, int32_t n_edge, const ovs_ba_cam* cam, double huber_delta,
                            double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* chi2) {
    return ba_linearize_host_impl(0, device, poses, pose_fixed, n_pose, points, n_pt, edges, n_edge, cam, huber_delta, Hpp, bp, Hll, bl, Hpl, chi2);
}

ovs_status ovs_ba_linearize_equirect(int32_t device, const double* poses, const uint8_t* pose_fixed, int32_t n_pose, const double* points,
                                     int32_t n_pt, const ovs_ba_edge* edges, int32_t n_edge, int32_t cols, int32_t rows, double huber_delta,
                                     double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* chi2) {
    if (cols < 1 || rows < 1) return OVS_ERR_INVALID;
    const ovs_ba_cam c = {(double)cols, (double)rows, 0.0, 0.0};
    return ba_linearize_host_impl(1, device, poses, pose_fixed, n_pose, points, n_pt, edges, n_edge, &c, huber_delta, Hpp, bp, Hll, bl, Hpl, chi2);
}

ovs_status ovs_ba_linearize_stereo(int32_t device, const double* poses, const uint8_t* pose_fixed, int32_t n_pose, const double* points,
                                   int32_t n_pt, const ovs_ba_edge_stereo* edges, int32_t n_edge, const ovs_ba_cam* cam,
                                   double focal_x_baseline, double huber_delta, double* Hpp, double* bp, double* Hll, double* bl,
                                   double* Hpl, double* chi2) {
    if (!poses || !points || !cam || !Hpp || !bp || !Hll || !bl || !Hpl || !chi2 || n_pose < 1 || n_pt < 1 || n_edge < 0 || (n_edge > 0 && !edges))
        return OVS_ERR_INVALID;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    OVS_HIP_TRY(hipSetDevice(device));
    const size_t ne = (size_t)std::max(n_edge, 1);
    const size_t n_out = (size_t)42 * n_pose + (size_t)12 * n_pt + 18 * ne + 2;
    double *d_poses = nullptr, *d_points = nullptr, *d_out = nullptr;
    ovs_ba_edge_stereo* d_edges = nullptr;
    uint8_t* d_fixed = nullptr;
    ovs_status st = OVS_ERR_HIP;
    hipError_t er = hipSuccess;
    do {
#define BA_TRY(expr)                              \
    if ((er = (expr)) != hipSuccess) {            \
        ovs::set_last_error(#expr, er);           \
        break;                                    \
    }
        BA_TRY(hipMalloc(&d_poses, sizeof(double) * 7 * n_pose));
        BA_TRY(hipMalloc(&d_points, sizeof(double) * 3 * n_pt));
        BA_TRY(hipMalloc(&d_edges, sizeof(ovs_ba_edge_stereo) * ne));
        BA_TRY(hipMalloc(&d_fixed, (size_t)n_pose));
        BA_TRY(hipMalloc(&d_out, sizeof(double) * n_out));
        BA_TRY(hipMemcpy(d_poses, poses, sizeof(double) * 7 * n_pose, hipMemcpyHostToDevice));
        BA_TRY(hipMemcpy(d_points, points, sizeof(double) * 3 * n_pt, hipMemcpyHostToDevice));
        if (n_edge) BA_TRY(hipMemcpy(d_edges, edges, sizeof(ovs_ba_edge_stereo) * (size_t)n_edge, hipMemcpyHostToDevice));
        if (pose_fixed) BA_TRY(hipMemcpy(d_fixed, pose_fixed, (size_t)n_pose, hipMemcpyHostToDevice));
        double* dHpp = d_out;
        double* dbp = dHpp + 36 * (size_t)n_pose;
        double* dHll = dbp + 6 * (size_t)n_pose;
        double* dbl = dHll + 9 * (size_t)n_pt;
        double* dHpl = dbl + 3 * (size_t)n_pt;
        double* dchi = dHpl + 18 * ne;
        st = ovs_ba_linearize_stereo_dev(d_poses, pose_fixed ? d_fixed : nullptr, n_pose, d_points, n_pt, d_edges, n_edge, cam, focal_x_baseline,
                                         huber_delta, 0, dHpp, dbp, dHll, dbl, dHpl, dchi, nullptr);
        if (st != OVS_OK) break;
        st = OVS_ERR_HIP;
        BA_TRY(hipDeviceSynchronize());
        BA_TRY(hipMemcpy(Hpp, dHpp, sizeof(double) * 36 * n_pose, hipMemcpyDeviceToHost));
        BA_TRY(hipMemcpy(bp, dbp, sizeof(double) * 6 * n_pose, hipMemcpyDeviceToHost));
        BA_TRY(hipMemcpy(Hll, dHll, sizeof(double) * 9 * n_pt, hipMemcpyDeviceToHost));
        BA_TRY(hipMemcpy(bl, dbl, sizeof(double) * 3 * n_pt, hipMemcpyDeviceToHost));
        if (n_edge) BA_TRY(hipMemcpy(Hpl, dHpl, sizeof(double) * 18 * (size_t)n_edge, hipMemcpyDeviceToHost));
        BA_TRY(hipMemcpy(chi2, dchi, sizeof(double) * 2, hipMemcpyDeviceToHost));
        st = OVS_OK;
#undef BA_TRY
    } while (0);
    hipFree(d_poses);
    hipFree(d_points);
    hipFree(d_edges);
    hipFree(d_fixed);
    hipFree(d_out);
    return st;
}

static ovs_status ba_linearize_host_impl(int model, int32_t device, const double* poses, const uint8_t* pose_fixed, int32_t n_pose,
                                         const double* points, int32_t n_pt, const ovs_ba_edge* edges, int32_t n_edge, const ovs_ba_cam* cam,
                                         double huber_delta, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* chi2) {
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
        st = ba_linearize_dev_impl(model, reinterpret_cast<double*>(d_in), pose_fixed ? d_in + off_f : nullptr, n_pose,
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
