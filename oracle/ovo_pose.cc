// ovo_pose.cc -- CPU ORACLE (test infrastructure, see ovo_oracle.h): optimize::pose_optimizer::optimize restated from spec
// (expected: src/openvslam/optimize/pose_optimizer.{h,cc}, optimize/g2o/se3/{pose_opt_edge_wrapper.h, perspective_pose_opt_edge.*,
// shot_vertex.*}; g2o OptimizationAlgorithmLevenberg, RobustKernelHuber, SE3Quat::exp). PARITY UNPINNED (upstream and g2o absent).
//
// One free vertex (the frame's pose, world -> camera), one unary edge per observed landmark (2 residuals, or 3 for a stereo
// keypoint), landmark positions fixed. Upstream's schedule: 4 rounds; every round re-initialises the pose vertex with the frame's
// INITIAL pose, runs 10 Levenberg-Marquardt iterations over the current inlier edges, then re-classifies EVERY edge with the new
// pose (chi2 > 5.99146f mono / 7.81473f stereo -> outlier, excluded from the next round); from round 2 on the Huber kernel is removed; the
// loop stops early when fewer than 10 edges are left in the graph (all edges stay in the graph upstream, so this only triggers for
// n < 10). Result: final pose, outlier flags, number of inliers.
//
// g2o's Levenberg-Marquardt, as restated here (ORACLE_SPEC rule 25): lambda_0 = 1e-5 * max |H_jj| at the first iteration of every
// round; trial step (H + lambda I) dx = b by Cholesky; rho = (chi_cur - chi_new) / (dx . (lambda dx + b) + 1e-3); accepted
// (rho > 0 and finite): lambda *= clamp(1 - (2 rho - 1)^3, 1/3, 2/3), ni = 2; rejected: lambda *= ni, ni *= 2, state restored, up
// to 10 trials per iteration; an iteration that ends rejected (or rho == 0) terminates the round. chi values are the ROBUSTIFIED sums.
// Pose update: T <- exp([omega, upsilon]) * T (g2o SE3Quat::exp, left-multiplicative).
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "ovo_oracle.h"
#include "../include/ovs_detmath.h"

namespace {

struct Pose {
    double R[9], t[3];
};

// g2o SE3Quat::exp(update): omega = update[0..2], upsilon = update[3..5]
void se3_exp(const double* u, Pose& out) {
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = std::sqrt((wx * wx + wy * wy) + wz * wz);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = (O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j]) + O[3 * i + 2] * O[6 + j];
    double V[9];
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        if (theta < 0.00001) {
            out.R[i] = (I + O[i]) + O2[i];
            V[i] = out.R[i];   // g2o: V = R for tiny angles
        } else {
            const double s = std::sin(theta), c = std::cos(theta);
            out.R[i] = (I + s / theta * O[i]) + (1 - c) / (theta * theta) * O2[i];
            V[i] = (I + (1 - c) / (theta * theta) * O[i]) + (theta - s) / (theta * theta * theta) * O2[i];
        }
    }
    for (int i = 0; i < 3; ++i) out.t[i] = (V[3 * i] * u[3] + V[3 * i + 1] * u[4]) + V[3 * i + 2] * u[5];
}

void compose(const Pose& a, const Pose& b, Pose& out) {   // out = a * b
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out.R[3 * i + j] = (a.R[3 * i] * b.R[j] + a.R[3 * i + 1] * b.R[3 + j]) + a.R[3 * i + 2] * b.R[6 + j];
        out.t[i] = ((a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1]) + a.R[3 * i + 2] * b.t[2]) + a.t[i];
    }
}

struct Lin {
    double H[36], b[6], chi_robust;
};

// residual, chi2 and (optionally) the edge's contribution to H, b for one observation
// equirectangular variant (expected: src/openvslam/optimize/g2o/se3/equirectangular_pose_opt_edge.{h,cc}): cam = {cols, rows, -, -},
// e = obs - (cols (1/2 + atan2(x, z) / 2 pi), rows (1/2 + asin(y / |p|) / pi)), no seam wrap-around (ORACLE_SPEC rule 26); the 2 x 6
// Jacobian is the pose part of ovo_ba_linearize_equirect, same operation order. asin / atan2: include/ovs_detmath.h (shared, rule 24).
inline double edge_eval_equirect(const Pose& T, const ovo_pose_obs& o, const double* cam, double delta, Lin* lin) {
    const double x = ((T.R[0] * o.pos_w[0] + T.R[1] * o.pos_w[1]) + T.R[2] * o.pos_w[2]) + T.t[0];
    const double y = ((T.R[3] * o.pos_w[0] + T.R[4] * o.pos_w[1]) + T.R[5] * o.pos_w[2]) + T.t[1];
    const double z = ((T.R[6] * o.pos_w[0] + T.R[7] * o.pos_w[1]) + T.R[8] * o.pos_w[2]) + T.t[2];
    const double kPi = 3.14159265358979323846;
    const double cols = cam[0], rows = cam[1];
    const double L = std::sqrt((x * x + y * y) + z * z);
    const double rxz = x * x + z * z;
    const double theta = ovs_det_atan2(x, z);
    const double phi = -ovs_det_asin(y / L);
    const double e0 = o.obs_x - cols * (0.5 + theta / (2.0 * kPi));
    const double e1 = o.obs_y - rows * (0.5 - phi / kPi);
    const double c2 = o.inv_sigma_sq * (e0 * e0 + e1 * e1);
    if (!lin) return c2;
    double rho0 = c2, rho1 = 1.0;
    const double dsqr = delta * delta;
    if (delta > 0 && c2 > dsqr) {
        const double sq = std::sqrt(c2);
        rho0 = 2 * sq * delta - dsqr;
        rho1 = delta / sq;
    }
    lin->chi_robust += rho0;
    const double a0 = -(cols / (2.0 * kPi)) * (1.0 / rxz);
    const double a1 = -(rows / kPi) * (1.0 / (L * std::sqrt(rxz)));
    double J[2][6];
    auto col = [&](double dx, double dy, double dz, double& j0, double& j1) {
        const double dL = (1.0 / L) * ((x * dx + y * dy) + z * dz);
        j0 = a0 * (z * dx - x * dz);
        j1 = a1 * (L * dy - y * dL);
    };
    col(0.0, -z, y, J[0][0], J[1][0]);
    col(z, 0.0, -x, J[0][1], J[1][1]);
    col(-y, x, 0.0, J[0][2], J[1][2]);
    col(1.0, 0.0, 0.0, J[0][3], J[1][3]);
    col(0.0, 1.0, 0.0, J[0][4], J[1][4]);
    col(0.0, 0.0, 1.0, J[0][5], J[1][5]);
    const double W = rho1 * o.inv_sigma_sq;
    for (int a = 0; a < 6; ++a) {
        for (int b = 0; b < 6; ++b) lin->H[6 * a + b] += W * (J[0][a] * J[0][b] + J[1][a] * J[1][b]);
        lin->b[a] += -(W * (J[0][a] * e0 + J[1][a] * e1));
    }
    return c2;
}

inline double edge_eval(const Pose& T, const ovo_pose_obs& o, const double* cam, double bf, double delta, Lin* lin, int model = 0) {
    if (model == 1) return edge_eval_equirect(T, o, cam, delta, lin);
    const double x = ((T.R[0] * o.pos_w[0] + T.R[1] * o.pos_w[1]) + T.R[2] * o.pos_w[2]) + T.t[0];
    const double y = ((T.R[3] * o.pos_w[0] + T.R[4] * o.pos_w[1]) + T.R[5] * o.pos_w[2]) + T.t[1];
    const double z = ((T.R[6] * o.pos_w[0] + T.R[7] * o.pos_w[1]) + T.R[8] * o.pos_w[2]) + T.t[2];
    const double invz = 1.0 / z, invz2 = invz * invz;
    const double fx = cam[0], fy = cam[1];
    const int D = o.is_stereo ? 3 : 2;
    double e[3];
    const double u = fx * x * invz + cam[2];
    e[0] = o.obs_x - u;
    e[1] = o.obs_y - (fy * y * invz + cam[3]);
    e[2] = o.is_stereo ? o.obs_x_right - (u - bf * invz) : 0.0;
    double ss = e[0] * e[0] + e[1] * e[1];
    if (D == 3) ss = ss + e[2] * e[2];
    const double c2 = o.inv_sigma_sq * ss;
    if (!lin) return c2;
    double rho0 = c2, rho1 = 1.0;
    const double dsqr = delta * delta;
    if (delta > 0 && c2 > dsqr) {
        const double sq = std::sqrt(c2);
        rho0 = 2 * sq * delta - dsqr;
        rho1 = delta / sq;
    }
    lin->chi_robust += rho0;
    double J[3][6];
    J[0][0] = x * y * invz2 * fx;
    J[0][1] = -(1 + x * x * invz2) * fx;
    J[0][2] = y * invz * fx;
    J[0][3] = -invz * fx;
    J[0][4] = 0;
    J[0][5] = x * invz2 * fx;
    J[1][0] = (1 + y * y * invz2) * fy;
    J[1][1] = -x * y * invz2 * fy;
    J[1][2] = -x * invz * fy;
    J[1][3] = 0;
    J[1][4] = -invz * fy;
    J[1][5] = y * invz2 * fy;
    J[2][0] = J[0][0] - bf * y * invz2;
    J[2][1] = J[0][1] + bf * x * invz2;
    J[2][2] = J[0][2];
    J[2][3] = J[0][3];
    J[2][4] = 0;
    J[2][5] = J[0][5] - bf * invz2;
    const double W = rho1 * o.inv_sigma_sq;
    for (int a = 0; a < 6; ++a) {
        for (int b = 0; b < 6; ++b) {
            double s = J[0][a] * J[0][b] + J[1][a] * J[1][b];
            if (D == 3) s = s + J[2][a] * J[2][b];
            lin->H[6 * a + b] += W * s;
        }
        double g = J[0][a] * e[0] + J[1][a] * e[1];
        if (D == 3) g = g + J[2][a] * e[2];
        lin->b[a] += -(W * g);
    }
    return c2;
}

// robustified chi2 of the active edges for pose T (no Jacobians)
double robust_chi(const Pose& T, const ovo_pose_obs* obs, int n, const std::vector<uint8_t>& active, const double* cam, double bf,
                  double delta, int model) {
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        if (!active[i]) continue;
        const double c2 = edge_eval(T, obs[i], cam, bf, 0, nullptr, model);
        double r = c2;
        if (delta > 0 && c2 > delta * delta) r = 2 * std::sqrt(c2) * delta - delta * delta;
        sum += r;
    }
    return sum;
}

// (H + lambda I) x = b by Cholesky (LL^T); false if not positive definite
bool solve6(const double* H, double lambda, const double* b, double* x) {
    double L[36] = {0};
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = H[6 * i + j] + (i == j ? lambda : 0.0);
            for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
            if (i == j) {
                if (!(s > 0)) return false;
                L[6 * i + i] = std::sqrt(s);
            } else {
                L[6 * i + j] = s / L[6 * j + j];
            }
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
        y[i] = s / L[6 * i + i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
        x[i] = s / L[6 * i + i];
    }
    return true;
}

}   // namespace

namespace {
// ORACLE_SPEC rule 25 (iv) as a run-time variant: 0 (default) the frame vertex is initialised ONCE, rounds continue from the previous estimate
// (OpenVSLAM as recalled); 1 it is re-set to the input pose at the start of every round (ORB-SLAM2's Optimizer::PoseOptimization)
int g_pose_reset_each_round = 0;
int pose_optimize_impl(const double* pose_cw_in, const ovo_pose_obs* obs, int n, const double* cam4, double bf, int setup_type, int model,
                       double* pose_cw_out, uint8_t* outlier, int* num_valid) {
    // upstream: sqrt_chi_sq = (frm.camera_->setup_type_ == Monocular) ? sqrt_chi_sq_2D : sqrt_chi_sq_3D -- ONE Huber delta for every
    // edge of the frame, chosen by the rig; the chi-square outlier gates below stay per edge (is_monocular_)
    // upstream: constexpr float chi_sq_2D = 5.99146; const float sqrt_chi_sq_2D = std::sqrt(chi_sq_2D); (3D: 7.81473) -- FLOAT constants widened
    // to double where g2o consumes them (ORACLE_SPEC rule 25); hex literals so no library sqrt is involved
    const float chi_sq_2D = 5.99146f, chi_sq_3D = 7.81473f;
    const float sqrt_chi_sq_2D = std::sqrt(chi_sq_2D), sqrt_chi_sq_3D = std::sqrt(chi_sq_3D);
    const double huber = setup_type == 0 ? (double)sqrt_chi_sq_2D : (double)sqrt_chi_sq_3D;
    Pose T0;
    std::memcpy(T0.R, pose_cw_in, sizeof(double) * 9);
    std::memcpy(T0.t, pose_cw_in + 9, sizeof(double) * 3);
    std::vector<uint8_t> active((size_t)n, 1);
    for (int i = 0; i < n; ++i) outlier[i] = 0;
    Pose T = T0;
    int num_bad = 0;
    if (n >= 5) {   // upstream: if (num_init_obs < 5) return 0;
        for (int trial = 0; trial < 4; ++trial) {
            // `if (trial == num_trials_ - 2) edge->setRobustKernel(nullptr)` runs AFTER trial 2's optimisation: Huber in trials 0, 1, 2.
            // The frame vertex is set to the initial pose once, before the loop (unlike ORB-SLAM2, which resets it every round):
            // each round continues from the previous round's estimate.
            const bool robust = trial < 3;
            if (g_pose_reset_each_round) T = T0;
            Pose Terr = T;       // the state the ACTIVE edges' errors were last computed at (g2o leaves them stale after a rejected step)
            double lambda = 0, ni = 2;
            for (int it = 0; it < 10; ++it) {
                Lin lin;
                std::memset(&lin, 0, sizeof(lin));
                Terr = T;   // solve() starts with computeActiveErrors() at the current estimate
                for (int i = 0; i < n; ++i)
                    if (active[i])
                        edge_eval(T, obs[i], cam4, bf, robust ? huber : 0.0, &lin, model);
                double current_chi = lin.chi_robust;
                if (it == 0) {
                    double max_diag = 0;
                    for (int j = 0; j < 6; ++j) max_diag = std::max(std::fabs(lin.H[7 * j]), max_diag);
                    lambda = 1e-5 * max_diag;
                    ni = 2;
                }
                double rho = 0;
                int qmax = 0;
                do {
                    double dx[6];
                    const bool ok = solve6(lin.H, lambda, lin.b, dx);
                    Pose Tn = T;
                    double temp_chi = std::numeric_limits<double>::max();
                    if (ok) {
                        Pose E;
                        se3_exp(dx, E);
                        compose(E, T, Tn);
                        temp_chi = robust_chi(Tn, obs, n, active, cam4, bf, robust ? huber : 0.0, model);
                        Terr = Tn;   // computeActiveErrors() ran on the trial state
                    }
                    rho = current_chi - temp_chi;
                    double scale = 0;
                    if (ok)
                        for (int j = 0; j < 6; ++j) scale += dx[j] * (lambda * dx[j] + lin.b[j]);
                    scale += 1e-3;
                    rho /= scale;
                    if (rho > 0 && std::isfinite(temp_chi)) {
                        double alpha = 1. - std::pow(2 * rho - 1, 3);
                        alpha = std::min(alpha, 2.0 / 3.0);
                        lambda *= std::max(1.0 / 3.0, alpha);
                        ni = 2;
                        current_chi = temp_chi;
                        T = Tn;
                    } else {
                        lambda *= ni;
                        ni *= 2;
                    }
                    ++qmax;
                } while (rho < 0 && qmax < 10);
                if (qmax == 10 || rho == 0) break;
            }
            // re-classify every edge: upstream calls edge->computeError() only for the edges currently flagged as outliers (they were
            // not part of the optimisation); an inlier's chi2() is whatever the last computeActiveErrors() left, i.e. the last TRIAL
            // state -- equal to the estimate unless the round ended on a rejected step (qmax == 10 or rho == 0).
            num_bad = 0;
            for (int i = 0; i < n; ++i) {
                const double c2 = edge_eval(active[i] ? Terr : T, obs[i], cam4, bf, 0, nullptr, model);
                const double thr = (model == 0 && obs[i].is_stereo) ? (double)chi_sq_3D : (double)chi_sq_2D;
                if (thr < c2) {
                    outlier[i] = 1;
                    active[i] = 0;
                    ++num_bad;
                } else {
                    outlier[i] = 0;
                    active[i] = 1;
                }
            }
            if (n - num_bad < 5) break;   // upstream: if (num_init_obs - num_bad_obs < 5) break;
        }
    }
    std::memcpy(pose_cw_out, T.R, sizeof(double) * 9);
    std::memcpy(pose_cw_out + 9, T.t, sizeof(double) * 3);
    *num_valid = n >= 5 ? n - num_bad : 0;
    return 0;
}
}   // namespace

extern "C" int ovo_pose_set_variant(int which, int value) {
    if (which != 0 || (value != 0 && value != 1)) return -1;
    g_pose_reset_each_round = value;
    return 0;
}

extern "C" int ovo_pose_optimize(const double* pose_cw_in, const ovo_pose_obs* obs, int n, const double* cam4, double bf, int setup_type,
                                 double* pose_cw_out, uint8_t* outlier, int* num_valid) {
    return pose_optimize_impl(pose_cw_in, obs, n, cam4, bf, setup_type, 0, pose_cw_out, outlier, num_valid);
}

// equirectangular frames (upstream: `case camera::model_type_t::Equirectangular` of pose_optimizer::optimize -> equirectangular_pose_opt_edge):
// every edge is monocular, the rig is Monocular (one Huber delta sqrtf(5.99146f), gate 5.99146f); obs[i].is_stereo / obs_x_right are ignored
extern "C" int ovo_pose_optimize_equirect(const double* pose_cw_in, const ovo_pose_obs* obs, int n, int cols, int rows, double* pose_cw_out,
                                          uint8_t* outlier, int* num_valid) {
    const double cam[4] = {(double)cols, (double)rows, 0.0, 0.0};
    return pose_optimize_impl(pose_cw_in, obs, n, cam, 0.0, 0, 1, pose_cw_out, outlier, num_valid);
}
