// ba_optimize.hip -- B4: optimize::local_bundle_adjuster::optimize, the part behind the graph build (expected:
// src/openvslam/optimize/local_bundle_adjuster.cc; g2o OptimizationAlgorithmLevenberg + BlockSolver_6_3 with Schur complement).
//
// The caller (the class shim) flattens the local map into poses / landmarks / observation edges; this file runs what
// optimizer.optimize(num_first_iter) -> outlier levels -> optimizer.optimize(num_second_iter) does:
//   * every linearisation (residuals, Jacobians, J^T W J / J^T W e blocks, chi2) is one or two launches of the kernels in
//     ba_linearize.hip over buffers that stay in HBM; poses and points (0.48 MB at config 5) are re-uploaded per Levenberg-Marquardt
//     trial, the blocks come back once per trial;
//   * the reduced camera system (landmarks eliminated: S = Hpp - sum_j W_j Hll_j^-1 W_j^T) is formed (<= 8 host threads) and
//     Cholesky-factored on the HOST, as BASELINE's north star asks -- it is at most 6 * n_pose square;
//   * g2o's damping schedule (ORACLE_SPEC rule 25), the chi-square outlier gates between the two rounds and the final outlier flags.
// A trial's linearisation is kept as the next iteration's system when the step is accepted, so an iteration costs one launch pair.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "ovs_common.h"

namespace {

using ovs::set_last_error;

// per-edge chi2 = e^T Omega e at the current state and the sign of the depth (reproj_edge_wrapper::depth_is_positive)
template <int D, typename EDGE>
__global__ __launch_bounds__(256) void k_ba_edge_chi2(const double* __restrict__ poses, const double* __restrict__ points,
                                                     const EDGE* __restrict__ edges, int n_edge, ovs_ba_cam cam, double bf,
                                                     double* __restrict__ chi2, uint8_t* __restrict__ depth_pos) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edge) return;
    const EDGE ed = edges[e];
    const double* P = poses + 7 * (size_t)ed.pose_idx;
    const double* X = points + 3 * (size_t)ed.point_idx;
    const double qx = P[3], qy = P[4], qz = P[5], qw = P[6];
    const double tx2 = 2 * qx, ty2 = 2 * qy, tz2 = 2 * qz;
    const double twx = tx2 * qw, twy = ty2 * qw, twz = tz2 * qw;
    const double txx = tx2 * qx, txy = ty2 * qx, txz = tz2 * qx;
    const double tyy = ty2 * qy, tyz = tz2 * qy, tzz = tz2 * qz;
    const double x = (1 - (tyy + tzz)) * X[0] + (txy - twz) * X[1] + (txz + twy) * X[2] + P[0];
    const double y = (txy + twz) * X[0] + (1 - (txx + tzz)) * X[1] + (tyz - twx) * X[2] + P[1];
    const double z = (txz - twy) * X[0] + (tyz + twx) * X[1] + (1 - (txx + tyy)) * X[2] + P[2];
    const double invz = 1.0 / z;
    const double u = cam.fx * x * invz + cam.cx;
    const double e0 = ed.obs_x - u, e1 = ed.obs_y - (cam.fy * y * invz + cam.cy);
    double ss = e0 * e0 + e1 * e1;
    if constexpr (D == 3) {
        const double e2 = ed.obs_x_right - (u - bf * invz);
        ss = ss + e2 * e2;
    }
    chi2[e] = ed.inv_sigma_sq * ss;
    depth_pos[e] = z > 0.0 ? 1 : 0;
}

struct Pose {   // world -> camera, rotation matrix row-major + translation
    double R[9], t[3];
};

void quat_to_rot(const double* q, double* R) {   // q = (x, y, z, w)
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z);
    R[1] = 2 * (x * y - z * w);
    R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);
    R[4] = 1 - 2 * (x * x + z * z);
    R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);
    R[7] = 2 * (y * z + x * w);
    R[8] = 1 - 2 * (x * x + y * y);
}

void rot_to_quat(const double* R, double* q) {   // Eigen's Quaternion(Matrix3) branches, then w >= 0 and unit norm (SE3Quat::normalizeRotation)
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double s = std::sqrt(tr + 1.0);
        q[3] = 0.5 * s;
        s = 0.5 / s;
        q[0] = (R[7] - R[5]) * s;
        q[1] = (R[2] - R[6]) * s;
        q[2] = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * s;
        s = 0.5 / s;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
    if (q[3] < 0)
        for (int a = 0; a < 4; ++a) q[a] = -q[a];
    const double n = std::sqrt((q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]));
    for (int a = 0; a < 4; ++a) q[a] /= n;
}

// T <- exp([omega, upsilon]) * T   (g2o SE3Quat::exp: R = I + O + O^2 and V = R below 1e-5 rad)
void se3_oplus(Pose& T, const double* u) {
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = std::sqrt((wx * wx + wy * wy) + wz * wz);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9], E[9], V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = (O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j]) + O[3 * i + 2] * O[6 + j];
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        if (theta < 0.00001) {
            E[i] = (I + O[i]) + O2[i];
            V[i] = E[i];
        } else {
            const double s = std::sin(theta), c = std::cos(theta);
            E[i] = (I + s / theta * O[i]) + (1 - c) / (theta * theta) * O2[i];
            V[i] = (I + (1 - c) / (theta * theta) * O[i]) + (theta - s) / (theta * theta * theta) * O2[i];
        }
    }
    Pose n;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) n.R[3 * i + j] = (E[3 * i] * T.R[j] + E[3 * i + 1] * T.R[3 + j]) + E[3 * i + 2] * T.R[6 + j];
        const double te = (V[3 * i] * u[3] + V[3 * i + 1] * u[4]) + V[3 * i + 2] * u[5];
        n.t[i] = ((E[3 * i] * T.t[0] + E[3 * i + 1] * T.t[1]) + E[3 * i + 2] * T.t[2]) + te;
    }
    T = n;
}

bool inv3_sym(const double* H, double lambda, double* out) {   // (H + lambda I)^-1 by cofactors
    const double a = H[0] + lambda, b = H[1], c = H[2], d = H[4] + lambda, e = H[5], f = H[8] + lambda;
    const double A = d * f - e * e, B = c * e - b * f, Cc = b * e - c * d;
    const double det = (a * A + b * B) + c * Cc;
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    const double id = 1.0 / det;
    out[0] = A * id;
    out[1] = out[3] = B * id;
    out[2] = out[6] = Cc * id;
    out[4] = (a * f - c * c) * id;
    out[5] = out[7] = (b * c - a * e) * id;
    out[8] = (a * d - b * b) * id;
    return true;
}

// In place: A (lower triangle read, row-major) -> L, b -> x. Right-looking form: the inner loop is an axpy over a contiguous row
// segment against a contiguous copy of the pivot column, which the host compiler vectorises without re-associating any sum.
bool cholesky_solve(std::vector<double>& A, int n, std::vector<double>& b) {
    std::vector<double> col((size_t)n);
    for (int j = 0; j < n; ++j) {
        const double piv = A[(size_t)j * n + j];
        if (!(piv > 0.0)) return false;
        const double ljj = std::sqrt(piv);
        A[(size_t)j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            A[(size_t)i * n + j] /= ljj;
            col[i] = A[(size_t)i * n + j];
        }
        for (int i = j + 1; i < n; ++i) {
            const double lij = col[i];
            double* row = &A[(size_t)i * n];
            for (int k = j + 1; k <= i; ++k) row[k] -= lij * col[k];
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    return true;
}

struct Blocks {   // one linearisation, host copy in PINNED memory (16 MB per trial at config 5: pageable copies cost more than the kernels):
                  // Hpp | bp | Hll | bl | Hpl (mono edges, then stereo edges) | chi2[2]
    double* buf = nullptr;
    size_t cap = 0, used = 0;
    double *Hpp = nullptr, *bp = nullptr, *Hll = nullptr, *bl = nullptr, *Hpl = nullptr, *chi2 = nullptr;
    Blocks() = default;
    Blocks(const Blocks&) = delete;
    Blocks& operator=(const Blocks&) = delete;
    ~Blocks() {
        if (buf) hipHostFree(buf);
    }
    hipError_t reserve(int n_pose, int n_pt, size_t n_edge_max) {
        cap = (size_t)42 * n_pose + (size_t)12 * n_pt + 18 * std::max<size_t>(n_edge_max, 1) + 2;
        return hipHostMalloc(reinterpret_cast<void**>(&buf), sizeof(double) * cap, hipHostMallocDefault);
    }
    void layout(int n_pose, int n_pt, size_t n_edge) {
        Hpp = buf;
        bp = Hpp + (size_t)36 * n_pose;
        Hll = bp + (size_t)6 * n_pose;
        bl = Hll + (size_t)9 * n_pt;
        Hpl = bl + (size_t)3 * n_pt;
        chi2 = Hpl + 18 * std::max<size_t>(n_edge, 1);
        used = (size_t)(chi2 - buf) + 2;
    }
    void swap(Blocks& o) {
        std::swap(buf, o.buf);
        std::swap(cap, o.cap);
        std::swap(used, o.used);
        std::swap(Hpp, o.Hpp);
        std::swap(bp, o.bp);
        std::swap(Hll, o.Hll);
        std::swap(bl, o.bl);
        std::swap(Hpl, o.Hpl);
        std::swap(chi2, o.chi2);
    }
};

// upstream: constexpr float chi_sq_2D = 5.99146, chi_sq_3D = 7.81473 and their float square roots, widened to double where g2o consumes them
constexpr double kChi2D = 0x1.7f7414p+2, kChi3D = 0x1.f4248ap+2, kSqrtChi2D = 0x1.394fbcp+1, kSqrtChi3D = 0x1.65d26ap+1;

struct Lba {
    int n_pose = 0, n_pt = 0;
    int setup_type = 0;   // camera::setup_type_t of the rig: selects the Huber delta of the mono edges
    const uint8_t* fixed = nullptr;
    ovs_ba_cam cam{};
    double bf = 0;
    // active edges of the current round (host copies; the device holds the same, mono then stereo)
    std::vector<ovs_ba_edge> mono;
    std::vector<ovs_ba_edge_stereo> stereo;
    // landmark -> its active edges (index into [mono | stereo]) for the Schur complement
    std::vector<int> lm_start, lm_edges, edge_pose;
    std::vector<int> slot;   // pose -> row block of the reduced system, -1 if fixed
    int n_free = 0;
    // device
    double *d_poses = nullptr, *d_points = nullptr, *d_out = nullptr, *d_echi = nullptr;
    uint8_t *d_fixed = nullptr, *d_edepth = nullptr;
    ovs_ba_edge* d_mono = nullptr;
    ovs_ba_edge_stereo* d_stereo = nullptr;
    size_t cap_mono = 0, cap_stereo = 0;
    hipStream_t stream = nullptr;

    ~Lba() {
        hipFree(d_poses);
        hipFree(d_points);
        hipFree(d_out);
        hipFree(d_echi);
        hipFree(d_fixed);
        hipFree(d_edepth);
        hipFree(d_mono);
        hipFree(d_stereo);
        if (stream) hipStreamDestroy(stream);
    }
    size_t n_edge() const { return mono.size() + stereo.size(); }

    ovs_status init_device() {
        OVS_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        const size_t ne = std::max<size_t>(cap_mono + cap_stereo, 1);
        OVS_HIP_TRY(hipMalloc(&d_poses, sizeof(double) * 7 * n_pose));
        OVS_HIP_TRY(hipMalloc(&d_points, sizeof(double) * 3 * n_pt));
        OVS_HIP_TRY(hipMalloc(&d_fixed, (size_t)n_pose));
        OVS_HIP_TRY(hipMalloc(&d_out, sizeof(double) * ((size_t)42 * n_pose + (size_t)12 * n_pt + 18 * ne + 2)));
        OVS_HIP_TRY(hipMalloc(&d_echi, sizeof(double) * ne));
        OVS_HIP_TRY(hipMalloc(&d_edepth, ne));
        OVS_HIP_TRY(hipMalloc(&d_mono, sizeof(ovs_ba_edge) * std::max<size_t>(cap_mono, 1)));
        OVS_HIP_TRY(hipMalloc(&d_stereo, sizeof(ovs_ba_edge_stereo) * std::max<size_t>(cap_stereo, 1)));
        if (fixed)
            OVS_HIP_TRY(hipMemcpyAsync(d_fixed, fixed, (size_t)n_pose, hipMemcpyHostToDevice, stream));
        else
            OVS_HIP_TRY(hipMemsetAsync(d_fixed, 0, (size_t)n_pose, stream));
        return OVS_OK;
    }

    // upload the round's edge set and index it by landmark
    ovs_status set_edges() {
        if (!mono.empty()) OVS_HIP_TRY(hipMemcpyAsync(d_mono, mono.data(), sizeof(ovs_ba_edge) * mono.size(), hipMemcpyHostToDevice, stream));
        if (!stereo.empty())
            OVS_HIP_TRY(hipMemcpyAsync(d_stereo, stereo.data(), sizeof(ovs_ba_edge_stereo) * stereo.size(), hipMemcpyHostToDevice, stream));
        const size_t ne = n_edge();
        edge_pose.resize(ne);
        std::vector<int> edge_pt(ne);
        for (size_t i = 0; i < mono.size(); ++i) {
            edge_pose[i] = mono[i].pose_idx;
            edge_pt[i] = mono[i].point_idx;
        }
        for (size_t i = 0; i < stereo.size(); ++i) {
            edge_pose[mono.size() + i] = stereo[i].pose_idx;
            edge_pt[mono.size() + i] = stereo[i].point_idx;
        }
        lm_start.assign((size_t)n_pt + 1, 0);
        for (size_t i = 0; i < ne; ++i) ++lm_start[(size_t)edge_pt[i] + 1];
        for (int j = 0; j < n_pt; ++j) lm_start[(size_t)j + 1] += lm_start[j];
        lm_edges.resize(ne);
        std::vector<int> fill(lm_start.begin(), lm_start.end() - 1);
        for (size_t i = 0; i < ne; ++i) lm_edges[(size_t)fill[edge_pt[i]]++] = (int)i;
        OVS_HIP_TRY(hipStreamSynchronize(stream));   // the vectors may be rebuilt before the next launch
        return OVS_OK;
    }

    void pack_poses(const std::vector<Pose>& T, std::vector<double>& p7) const {
        p7.resize((size_t)7 * n_pose);
        for (int k = 0; k < n_pose; ++k) {
            p7[(size_t)7 * k] = T[k].t[0];
            p7[(size_t)7 * k + 1] = T[k].t[1];
            p7[(size_t)7 * k + 2] = T[k].t[2];
            rot_to_quat(T[k].R, &p7[(size_t)7 * k + 3]);
        }
    }

    ovs_status upload_state(const std::vector<Pose>& T, const std::vector<double>& X) {
        std::vector<double> p7;
        pack_poses(T, p7);
        OVS_HIP_TRY(hipMemcpyAsync(d_poses, p7.data(), sizeof(double) * 7 * n_pose, hipMemcpyHostToDevice, stream));
        OVS_HIP_TRY(hipMemcpyAsync(d_points, X.data(), sizeof(double) * 3 * n_pt, hipMemcpyHostToDevice, stream));
        OVS_HIP_TRY(hipStreamSynchronize(stream));   // p7 is a local
        return OVS_OK;
    }

    ovs_status linearize(const std::vector<Pose>& T, const std::vector<double>& X, bool robust, Blocks& out) {
        ovs_status st = upload_state(T, X);
        if (st != OVS_OK) return st;
        const size_t ne = std::max<size_t>(n_edge(), 1);
        double* dHpp = d_out;
        double* dbp = dHpp + (size_t)36 * n_pose;
        double* dHll = dbp + (size_t)6 * n_pose;
        double* dbl = dHll + (size_t)9 * n_pt;
        double* dHpl = dbl + (size_t)3 * n_pt;
        double* dchi = dHpl + 18 * ne;
        // upstream: sqrt_chi_sq = (keyfrm->camera_->setup_type_ == Monocular) ? sqrt_chi_sq_2D : sqrt_chi_sq_3D -- the Huber delta follows the
        // RIG, not the edge (a mono observation in a stereo rig gets the 3D delta); float constants (ORACLE_SPEC rule 28)
        const double d_mono_h = robust ? (setup_type == 0 ? kSqrtChi2D : kSqrtChi3D) : 0.0, d_stereo_h = robust ? kSqrtChi3D : 0.0;
        st = ovs_ba_linearize_dev(d_poses, d_fixed, n_pose, d_points, n_pt, d_mono, (int32_t)mono.size(), &cam, d_mono_h, dHpp, dbp, dHll, dbl, dHpl,
                                  dchi, stream);
        if (st != OVS_OK) return st;
        if (!stereo.empty()) {
            st = ovs_ba_linearize_stereo_dev(d_poses, d_fixed, n_pose, d_points, n_pt, d_stereo, (int32_t)stereo.size(), &cam, bf, d_stereo_h, 1, dHpp,
                                             dbp, dHll, dbl, dHpl + 18 * mono.size(), dchi, stream);
            if (st != OVS_OK) return st;
        }
        out.layout(n_pose, n_pt, n_edge());
        OVS_HIP_TRY(hipMemcpyAsync(out.buf, d_out, sizeof(double) * out.used, hipMemcpyDeviceToHost, stream));
        OVS_HIP_TRY(hipStreamSynchronize(stream));
        return OVS_OK;
    }

    // per-edge chi2 / depth sign of the active edges at state (T, X)
    ovs_status edge_chi2(const std::vector<Pose>& T, const std::vector<double>& X, std::vector<double>& chi, std::vector<uint8_t>& depth) {
        ovs_status st = upload_state(T, X);
        if (st != OVS_OK) return st;
        const int nm = (int)mono.size(), ns = (int)stereo.size();
        if (nm) hipLaunchKernelGGL((k_ba_edge_chi2<2, ovs_ba_edge>), dim3((nm + 255) / 256), dim3(256), 0, stream, d_poses, d_points, d_mono, nm, cam, 0.0, d_echi, d_edepth);
        if (ns)
            hipLaunchKernelGGL((k_ba_edge_chi2<3, ovs_ba_edge_stereo>), dim3((ns + 255) / 256), dim3(256), 0, stream, d_poses, d_points, d_stereo, ns, cam, bf,
                               d_echi + nm, d_edepth + nm);
        OVS_HIP_TRY(hipGetLastError());
        chi.resize((size_t)nm + ns);
        depth.resize((size_t)nm + ns);
        if (nm + ns) {
            OVS_HIP_TRY(hipMemcpyAsync(chi.data(), d_echi, sizeof(double) * (nm + ns), hipMemcpyDeviceToHost, stream));
            OVS_HIP_TRY(hipMemcpyAsync(depth.data(), d_edepth, (size_t)(nm + ns), hipMemcpyDeviceToHost, stream));
        }
        OVS_HIP_TRY(hipStreamSynchronize(stream));
        return OVS_OK;
    }

    // (H + lambda I) dx = b through the Schur complement on the landmarks. dxp: n_pose x 6 (0 for fixed poses), dxl: n_pt x 3.
    bool solve(const Blocks& B, double lambda, std::vector<double>& dxp, std::vector<double>& dxl) const {
        const int n = 6 * n_free;
        std::vector<double> S((size_t)n * n, 0.0), g((size_t)n, 0.0);
        for (int k = 0; k < n_pose; ++k) {
            const int s = slot[k];
            if (s < 0) continue;
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) S[(size_t)(6 * s + a) * n + 6 * s + b] = B.Hpp[(size_t)36 * k + 6 * a + b] + (a == b ? lambda : 0.0);
                g[(size_t)6 * s + a] = B.bp[(size_t)6 * k + a];
            }
        }
        std::vector<double> Hinv((size_t)9 * n_pt);
        // landmark elimination: S -= W_j (Hll_j + lambda I)^-1 W_j^T, g -= W_j (..)^-1 bl_j. Landmarks are split over host threads, each
        // with its own accumulator (0.7 MB at 49 free poses), summed in thread order afterwards (deterministic for a given thread count).
        const int n_thr = (int)std::max(1u, std::min(8u, std::min(std::thread::hardware_concurrency(), (unsigned)(n_pt / 512 + 1))));
        std::vector<std::vector<double>> Sacc((size_t)n_thr), gacc((size_t)n_thr);
        std::vector<int> ok((size_t)n_thr, 1);
        auto work = [&](int t) {
            std::vector<double>& St = Sacc[(size_t)t];
            std::vector<double>& gt = gacc[(size_t)t];
            St.assign((size_t)n * n, 0.0);
            gt.assign((size_t)n, 0.0);
            std::vector<double> Y;   // W_e Hll^-1 of the landmark's edges
            const int j0 = (int)((long long)n_pt * t / n_thr), j1 = (int)((long long)n_pt * (t + 1) / n_thr);
            for (int j = j0; j < j1; ++j) {
                if (!inv3_sym(B.Hll + (size_t)9 * j, lambda, &Hinv[(size_t)9 * j])) {
                    ok[(size_t)t] = 0;
                    return;
                }
                const double* Hi = &Hinv[(size_t)9 * j];
                const int e0 = lm_start[j], e1 = lm_start[(size_t)j + 1];
                Y.resize((size_t)18 * (e1 - e0));
                for (int i = e0; i < e1; ++i) {
                    const int e = lm_edges[i];
                    if (slot[edge_pose[e]] < 0) continue;
                    const double* W = B.Hpl + (size_t)18 * e;
                    double* y = &Y[(size_t)18 * (i - e0)];
                    for (int a = 0; a < 6; ++a)
                        for (int c = 0; c < 3; ++c) y[3 * a + c] = (W[3 * a] * Hi[c] + W[3 * a + 1] * Hi[3 + c]) + W[3 * a + 2] * Hi[6 + c];
                    const int sl = slot[edge_pose[e]];
                    const double* blj = B.bl + (size_t)3 * j;
                    for (int a = 0; a < 6; ++a) gt[(size_t)6 * sl + a] -= (y[3 * a] * blj[0] + y[3 * a + 1] * blj[1]) + y[3 * a + 2] * blj[2];
                }
                for (int i = e0; i < e1; ++i) {
                    const int si = slot[edge_pose[lm_edges[i]]];
                    if (si < 0) continue;
                    const double* y = &Y[(size_t)18 * (i - e0)];
                    for (int i2 = e0; i2 < e1; ++i2) {
                        const int e2 = lm_edges[i2];
                        const int s2 = slot[edge_pose[e2]];
                        if (s2 < 0 || s2 < si) continue;   // upper block triangle only; mirrored below
                        const double* W2 = B.Hpl + (size_t)18 * e2;
                        double* dst = &St[(size_t)(6 * si) * n + 6 * s2];
                        for (int a = 0; a < 6; ++a)
                            for (int b = 0; b < 6; ++b)
                                dst[(size_t)a * n + b] -= (y[3 * a] * W2[3 * b] + y[3 * a + 1] * W2[3 * b + 1]) + y[3 * a + 2] * W2[3 * b + 2];
                    }
                }
            }
        };
        {
            std::vector<std::thread> pool;
            for (int t = 1; t < n_thr; ++t) pool.emplace_back(work, t);
            work(0);
            for (auto& th : pool) th.join();
        }
        for (int t = 0; t < n_thr; ++t) {
            if (!ok[(size_t)t]) return false;
            const std::vector<double>& St = Sacc[(size_t)t];
            const std::vector<double>& gt = gacc[(size_t)t];
            for (size_t i = 0; i < S.size(); ++i) S[i] += St[i];
            for (size_t i = 0; i < g.size(); ++i) g[i] += gt[i];
        }
        // diagonal blocks received each (i, i2) and (i2, i) pair of the same pose twice only when two edges share pose and landmark,
        // which a valid graph does not contain; off-diagonal blocks: mirror the upper triangle
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < r; ++c)
                if (r / 6 != c / 6) S[(size_t)r * n + c] = S[(size_t)c * n + r];
        if (n > 0 && !cholesky_solve(S, n, g)) return false;
        dxp.assign((size_t)6 * n_pose, 0.0);
        for (int k = 0; k < n_pose; ++k)
            if (slot[k] >= 0)
                for (int a = 0; a < 6; ++a) dxp[(size_t)6 * k + a] = g[(size_t)6 * slot[k] + a];
        dxl.assign((size_t)3 * n_pt, 0.0);
        for (int j = 0; j < n_pt; ++j) {
            double r[3] = {B.bl[(size_t)3 * j], B.bl[(size_t)3 * j + 1], B.bl[(size_t)3 * j + 2]};
            for (int i = lm_start[j]; i < lm_start[(size_t)j + 1]; ++i) {
                const int e = lm_edges[i];
                const int k = edge_pose[e];
                if (slot[k] < 0) continue;
                const double* W = B.Hpl + (size_t)18 * e;
                const double* d = &dxp[(size_t)6 * k];
                for (int c = 0; c < 3; ++c)
                    r[c] -= ((W[c] * d[0] + W[3 + c] * d[1]) + (W[6 + c] * d[2] + W[9 + c] * d[3])) + (W[12 + c] * d[4] + W[15 + c] * d[5]);
            }
            const double* Hi = &Hinv[(size_t)9 * j];
            for (int c = 0; c < 3; ++c) dxl[(size_t)3 * j + c] = (Hi[3 * c] * r[0] + Hi[3 * c + 1] * r[1]) + Hi[3 * c + 2] * r[2];
        }
        return true;
    }

    // one optimizer.optimize(iters) call. Returns the number of iterations entered.
    ovs_status run_round(std::vector<Pose>& T, std::vector<double>& X, int iters, bool robust, const volatile uint8_t* stop, double* chi_start,
                         double* chi_end, int* n_iter) {
        Blocks cur, trial;
        OVS_HIP_TRY(cur.reserve(n_pose, n_pt, cap_mono + cap_stereo));
        OVS_HIP_TRY(trial.reserve(n_pose, n_pt, cap_mono + cap_stereo));
        ovs_status st = linearize(T, X, robust, cur);
        if (st != OVS_OK) return st;
        double current_chi = cur.chi2[1];
        *chi_start = current_chi;
        *chi_end = current_chi;
        *n_iter = 0;
        if (iters <= 0 || n_edge() == 0) return OVS_OK;
        // computeLambdaInit: tau * the largest diagonal entry of the active vertices' Hessian blocks
        double max_diag = 0;
        for (int k = 0; k < n_pose; ++k)
            if (slot[k] >= 0)
                for (int a = 0; a < 6; ++a) max_diag = std::max(max_diag, std::fabs(cur.Hpp[(size_t)36 * k + 7 * a]));
        for (int j = 0; j < n_pt; ++j)
            if (lm_start[(size_t)j + 1] > lm_start[j])
                for (int a = 0; a < 3; ++a) max_diag = std::max(max_diag, std::fabs(cur.Hll[(size_t)9 * j + 4 * a]));
        double lambda = 1e-5 * max_diag, ni = 2;
        std::vector<double> dxp, dxl, Xn;
        std::vector<Pose> Tn;
        for (int it = 0; it < iters; ++it) {
            if (stop && *stop) break;
            ++*n_iter;
            double rho = 0;
            int qmax = 0;
            do {
                const bool ok = solve(cur, lambda, dxp, dxl);
                double temp_chi = 1.7976931348623157e308;
                double scale = 1e-3;
                if (ok) {
                    Tn = T;
                    Xn = X;
                    for (int k = 0; k < n_pose; ++k)
                        if (slot[k] >= 0) se3_oplus(Tn[k], &dxp[(size_t)6 * k]);
                    for (size_t i = 0; i < Xn.size(); ++i) Xn[i] += dxl[i];
                    st = linearize(Tn, Xn, robust, trial);
                    if (st != OVS_OK) return st;
                    temp_chi = trial.chi2[1];
                    double sc = 0;
                    for (int k = 0; k < n_pose; ++k)
                        if (slot[k] >= 0)
                            for (int a = 0; a < 6; ++a) sc += dxp[(size_t)6 * k + a] * (lambda * dxp[(size_t)6 * k + a] + cur.bp[(size_t)6 * k + a]);
                    for (size_t i = 0; i < dxl.size(); ++i) sc += dxl[i] * (lambda * dxl[i] + cur.bl[i]);
                    scale = sc + 1e-3;
                }
                rho = (current_chi - temp_chi) / scale;
                if (ok && rho > 0 && std::isfinite(temp_chi)) {
                    double alpha = 1.0 - std::pow(2 * rho - 1, 3.0);
                    alpha = std::min(alpha, 2.0 / 3.0);
                    lambda *= std::max(1.0 / 3.0, alpha);
                    ni = 2;
                    current_chi = temp_chi;
                    T.swap(Tn);
                    X.swap(Xn);
                    cur.swap(trial);   // the accepted trial's blocks are the next iteration's system
                } else {
                    lambda *= ni;
                    ni *= 2;
                    if (!std::isfinite(lambda)) break;
                }
                ++qmax;
            } while (rho < 0 && qmax < 10 && !(stop && *stop));
            if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) break;
        }
        *chi_end = current_chi;
        return OVS_OK;
    }
};

}   // namespace

extern "C" {

ovs_status ovs_local_ba_optimize(int32_t device, double* poses, const uint8_t* pose_fixed, int32_t n_pose, double* points, int32_t n_pt,
                                 const ovs_ba_edge* mono, int32_t n_mono, const ovs_ba_edge_stereo* stereo, int32_t n_stereo,
                                 const ovs_ba_cam* cam, double focal_x_baseline, int32_t setup_type, int32_t num_first_iter, int32_t num_second_iter,
                                 const volatile uint8_t* force_stop_flag, uint8_t* mono_outlier, uint8_t* stereo_outlier, double* info) {
    if (!poses || !points || !cam || n_pose < 1 || n_pt < 1 || n_mono < 0 || n_stereo < 0 || (n_mono > 0 && (!mono || !mono_outlier)) ||
        (n_stereo > 0 && (!stereo || !stereo_outlier)) || num_first_iter < 0 || num_second_iter < 0)
        return OVS_ERR_INVALID;
    for (int i = 0; i < n_mono; ++i)
        if (mono[i].pose_idx < 0 || mono[i].pose_idx >= n_pose || mono[i].point_idx < 0 || mono[i].point_idx >= n_pt) return OVS_ERR_INVALID;
    for (int i = 0; i < n_stereo; ++i)
        if (stereo[i].pose_idx < 0 || stereo[i].pose_idx >= n_pose || stereo[i].point_idx < 0 || stereo[i].point_idx >= n_pt) return OVS_ERR_INVALID;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    OVS_HIP_TRY(hipSetDevice(device));
    Lba L;
    L.n_pose = n_pose;
    L.n_pt = n_pt;
    L.fixed = pose_fixed;
    L.cam = *cam;
    L.bf = focal_x_baseline;
    L.setup_type = setup_type;
    L.cap_mono = (size_t)n_mono;
    L.cap_stereo = (size_t)n_stereo;
    L.slot.assign((size_t)n_pose, -1);
    for (int k = 0; k < n_pose; ++k)
        if (!(pose_fixed && pose_fixed[k])) L.slot[k] = L.n_free++;
    ovs_status st = L.init_device();
    if (st != OVS_OK) return st;
    std::vector<Pose> T((size_t)n_pose);
    for (int k = 0; k < n_pose; ++k) {
        quat_to_rot(poses + 7 * (size_t)k + 3, T[k].R);
        for (int a = 0; a < 3; ++a) T[k].t[a] = poses[7 * (size_t)k + a];
    }
    std::vector<double> X(points, points + 3 * (size_t)n_pt);
    double info_l[6] = {0, 0, 0, 0, 0, 0};

    // ---- round 1: all edges, Huber kernels
    L.mono.assign(mono, mono + n_mono);
    L.stereo.assign(stereo, stereo + n_stereo);
    st = L.set_edges();
    if (st != OVS_OK) return st;
    int it1 = 0, it2 = 0;
    st = L.run_round(T, X, num_first_iter, true, force_stop_flag, &info_l[0], &info_l[1], &it1);
    if (st != OVS_OK) return st;
    std::vector<double> chi;
    std::vector<uint8_t> depth;
    st = L.edge_chi2(T, X, chi, depth);
    if (st != OVS_OK) return st;
    std::vector<double> chi_r1 = chi;
    std::vector<uint8_t> out_r1((size_t)n_mono + n_stereo);
    for (int i = 0; i < n_mono; ++i) out_r1[i] = (kChi2D < chi[i]) || !depth[i];
    for (int i = 0; i < n_stereo; ++i) out_r1[(size_t)n_mono + i] = (kChi3D < chi[(size_t)n_mono + i]) || !depth[(size_t)n_mono + i];
    const bool stopped = force_stop_flag && *force_stop_flag;
    std::vector<int> map_m, map_s;   // active edge of round 2 -> original index
    if (!stopped) {
        // ---- round 2: inliers only (outliers go to level 1), no robust kernel
        L.mono.clear();
        L.stereo.clear();
        for (int i = 0; i < n_mono; ++i)
            if (!out_r1[i]) {
                L.mono.push_back(mono[i]);
                map_m.push_back(i);
            }
        for (int i = 0; i < n_stereo; ++i)
            if (!out_r1[(size_t)n_mono + i]) {
                L.stereo.push_back(stereo[i]);
                map_s.push_back(i);
            }
        st = L.set_edges();
        if (st != OVS_OK) return st;
        st = L.run_round(T, X, num_second_iter, false, force_stop_flag, &info_l[2], &info_l[3], &it2);
        if (st != OVS_OK) return st;
    }
    // ---- final outlier flags: an edge optimised in round 2 is judged at the final state; a level-1 edge keeps its round-1 chi2
    //      (g2o does not recompute the error of inactive edges) but its depth test sees the final state
    L.mono.assign(mono, mono + n_mono);
    L.stereo.assign(stereo, stereo + n_stereo);
    st = L.set_edges();
    if (st != OVS_OK) return st;
    st = L.edge_chi2(T, X, chi, depth);
    if (st != OVS_OK) return st;
    for (int i = 0; i < n_mono; ++i) {
        const double c = (!stopped && !out_r1[i]) ? chi[i] : chi_r1[i];
        mono_outlier[i] = (kChi2D < c) || !depth[i];
    }
    for (int i = 0; i < n_stereo; ++i) {
        const size_t e = (size_t)n_mono + i;
        const double c = (!stopped && !out_r1[e]) ? chi[e] : chi_r1[e];
        stereo_outlier[i] = (kChi3D < c) || !depth[e];
    }
    std::vector<double> p7;
    L.pack_poses(T, p7);
    for (int k = 0; k < n_pose; ++k)
        if (L.slot[k] >= 0) std::memcpy(poses + 7 * (size_t)k, &p7[(size_t)7 * k], sizeof(double) * 7);
    std::memcpy(points, X.data(), sizeof(double) * 3 * (size_t)n_pt);
    if (info) {
        info_l[4] = it1;
        info_l[5] = it2;
        std::memcpy(info, info_l, sizeof(info_l));
    }
    return OVS_OK;
}

}   // extern "C"
