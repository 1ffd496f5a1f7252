// ovo_ba.cc -- CPU ORACLE (test infrastructure, see ovo_oracle.h): residual / Jacobian / normal-equation blocks of local BA.
// PARITY UNPINNED (upstream and g2o absent). Restates SURVEY.md 8(a) rows B1-B3:
//   B1 optimize::g2o::se3::mono_perspective_reproj_edge::computeError   (expected: src/openvslam/optimize/g2o/se3/perspective_reproj_edge.cc)
//   B2 ...::linearizeOplus (identical to ORB-SLAM2 EdgeSE3ProjectXYZ: vertex 0 = landmark, vertex 1 = pose, pose order omega then upsilon)
//   B3 g2o BaseBinaryEdge::constructQuadraticForm + RobustKernelHuber::robustify (rho'' term dropped) + BlockSolver::buildSystem
// fp64 throughout, fixed evaluation order (edges in input order), no FMA contraction.
#include <cmath>
#include <cstring>

#include "ovo_oracle.h"
#include "../include/ovs_detmath.h"

extern "C" {

typedef struct ovo_ba_cam {
    double fx, fy, cx, cy;
} ovo_ba_cam;

typedef struct ovo_ba_edge {
    int32_t pose_idx, point_idx;
    double obs_x, obs_y;
    double inv_sigma_sq;   // information = inv_level_sigma_sq[octave] * I2
} ovo_ba_edge;

// poses: n_pose x 7 = (tx, ty, tz, qx, qy, qz, qw) (g2o SE3Quat::toVector order), world -> camera.
// Outputs (all zero-initialised here): Hpp n_pose x 36 (row-major 6x6), bp n_pose x 6, Hll n_pt x 9, bl n_pt x 3,
// Hpl n_edge x 18 (row-major 6x3 = Jp^T W Jl; zero for fixed poses), chi2[2] = {sum e^T Omega e, sum rho(e^T Omega e)}.
int ovo_ba_linearize(const double* poses, const uint8_t* pose_fixed, int n_pose, const double* points, int n_pt,
                     const ovo_ba_edge* edges, int n_edge, const ovo_ba_cam* cam, double huber_delta, double* Hpp, double* bp,
                     double* Hll, double* bl, double* Hpl, double* chi2) {
    std::memset(Hpp, 0, sizeof(double) * 36 * n_pose);
    std::memset(bp, 0, sizeof(double) * 6 * n_pose);
    std::memset(Hll, 0, sizeof(double) * 9 * n_pt);
    std::memset(bl, 0, sizeof(double) * 3 * n_pt);
    std::memset(Hpl, 0, sizeof(double) * 18 * n_edge);
    chi2[0] = chi2[1] = 0.0;
    const double dsqr = huber_delta * huber_delta;
    for (int e = 0; e < n_edge; ++e) {
        const ovo_ba_edge& ed = edges[e];
        if (ed.pose_idx < 0 || ed.pose_idx >= n_pose || ed.point_idx < 0 || ed.point_idx >= n_pt) return -1;
        const double* P = poses + 7 * ed.pose_idx;
        const double* X = points + 3 * ed.point_idx;
        // unit quaternion -> rotation matrix (Eigen's toRotationMatrix form)
        const double qx = P[3], qy = P[4], qz = P[5], qw = P[6];
        const double tx2 = 2 * qx, ty2 = 2 * qy, tz2 = 2 * qz;
        const double twx = tx2 * qw, twy = ty2 * qw, twz = tz2 * qw;
        const double txx = tx2 * qx, txy = ty2 * qx, txz = tz2 * qx;
        const double tyy = ty2 * qy, tyz = tz2 * qy, tzz = tz2 * qz;
        const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
        const double x = R[0][0] * X[0] + R[0][1] * X[1] + R[0][2] * X[2] + P[0];
        const double y = R[1][0] * X[0] + R[1][1] * X[1] + R[1][2] * X[2] + P[1];
        const double z = R[2][0] * X[0] + R[2][1] * X[1] + R[2][2] * X[2] + P[2];
        const double invz = 1.0 / z, invz2 = invz * invz;
        // B1: e = z_obs - pi(RX + t)
        const double e0 = ed.obs_x - (cam->fx * x * invz + cam->cx);
        const double e1 = ed.obs_y - (cam->fy * y * invz + cam->cy);
        const double w = ed.inv_sigma_sq;
        const double c2 = w * (e0 * e0 + e1 * e1);
        // B3: Huber weights (g2o RobustKernelHuber::robustify; huber_delta <= 0 means "no robust kernel")
        double rho0 = c2, rho1 = 1.0;
        if (huber_delta > 0 && c2 > dsqr) {
            const double sq = std::sqrt(c2);
            rho0 = 2 * sq * huber_delta - dsqr;
            rho1 = huber_delta / sq;
        }
        chi2[0] += c2;
        chi2[1] += rho0;
        // B2: Jacobians. Jl = -1/z * [fx 0 -fx x/z; 0 fy -fy y/z] * R   (2x3);  Jp (2x6), g2o SE3 order (omega, upsilon)
        double Jl[2][3], Jp[2][6];
        for (int c = 0; c < 3; ++c) {
            Jl[0][c] = -invz * (cam->fx * R[0][c] - cam->fx * x * invz * R[2][c]);
            Jl[1][c] = -invz * (cam->fy * R[1][c] - cam->fy * y * invz * R[2][c]);
        }
        Jp[0][0] = x * y * invz2 * cam->fx;
        Jp[0][1] = -(1 + x * x * invz2) * cam->fx;
        Jp[0][2] = y * invz * cam->fx;
        Jp[0][3] = -invz * cam->fx;
        Jp[0][4] = 0;
        Jp[0][5] = x * invz2 * cam->fx;
        Jp[1][0] = (1 + y * y * invz2) * cam->fy;
        Jp[1][1] = -x * y * invz2 * cam->fy;
        Jp[1][2] = -x * invz * cam->fy;
        Jp[1][3] = 0;
        Jp[1][4] = -invz * cam->fy;
        Jp[1][5] = y * invz2 * cam->fy;
        // weighted information W = rho1 * w * I2; omega_r = -rho1 * w * e
        const double W = rho1 * w;
        const double r0 = -W * e0, r1 = -W * e1;
        double* hl = Hll + 9 * ed.point_idx;
        double* gl = bl + 3 * ed.point_idx;
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) hl[3 * a + b] += W * (Jl[0][a] * Jl[0][b] + Jl[1][a] * Jl[1][b]);
            gl[a] += Jl[0][a] * r0 + Jl[1][a] * r1;
        }
        if (!(pose_fixed && pose_fixed[ed.pose_idx])) {
            double* hp = Hpp + 36 * ed.pose_idx;
            double* gp = bp + 6 * ed.pose_idx;
            double* hpl = Hpl + 18 * (size_t)e;
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) hp[6 * a + b] += W * (Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b]);
                gp[a] += Jp[0][a] * r0 + Jp[1][a] * r1;
                for (int b = 0; b < 3; ++b) hpl[3 * a + b] = W * (Jp[0][a] * Jl[0][b] + Jp[1][a] * Jl[1][b]);
            }
        }
    }
    return 0;
}

// B1/B2 equirectangular variant: optimize::g2o::se3::equirectangular_reproj_edge (expected: src/openvslam/optimize/g2o/se3/
// equirectangular_reproj_edge.cc). pi(p) = (cols (1/2 + atan2(x, z) / 2 pi), rows (1/2 + asin(y / |p|) / pi)); e = obs - pi(RX + t), NO
// wrap-around correction at the seam (ORACLE_SPEC rule 26). With dp the derivative of pos_c w.r.t. one state component:
//   d u = (cols / 2 pi) (z dp_x - x dp_z) / (x^2 + z^2),  d v = (rows / pi) (L dp_y - y dL) / (L sqrt(x^2 + z^2)),  dL = pos_c . dp / L, J = -d(u, v);
// columns: rotation e_k x pos_c, translation e_k, landmark R's columns. Same blocks / conventions as ovo_ba_linearize.
int ovo_ba_linearize_equirect(const double* poses, const uint8_t* pose_fixed, int n_pose, const double* points, int n_pt,
                              const ovo_ba_edge* edges, int n_edge, int cols, int rows, double huber_delta, double* Hpp, double* bp,
                              double* Hll, double* bl, double* Hpl, double* chi2) {
    std::memset(Hpp, 0, sizeof(double) * 36 * n_pose);
    std::memset(bp, 0, sizeof(double) * 6 * n_pose);
    std::memset(Hll, 0, sizeof(double) * 9 * n_pt);
    std::memset(bl, 0, sizeof(double) * 3 * n_pt);
    std::memset(Hpl, 0, sizeof(double) * 18 * n_edge);
    chi2[0] = chi2[1] = 0.0;
    const double kPi = 3.14159265358979323846;
    const double dsqr = huber_delta * huber_delta;
    for (int e = 0; e < n_edge; ++e) {
        const ovo_ba_edge& ed = edges[e];
        if (ed.pose_idx < 0 || ed.pose_idx >= n_pose || ed.point_idx < 0 || ed.point_idx >= n_pt) return -1;
        const double* P = poses + 7 * ed.pose_idx;
        const double* X = points + 3 * ed.point_idx;
        const double qx = P[3], qy = P[4], qz = P[5], qw = P[6];
        const double tx2 = 2 * qx, ty2 = 2 * qy, tz2 = 2 * qz;
        const double twx = tx2 * qw, twy = ty2 * qw, twz = tz2 * qw;
        const double txx = tx2 * qx, txy = ty2 * qx, txz = tz2 * qx;
        const double tyy = ty2 * qy, tyz = tz2 * qy, tzz = tz2 * qz;
        const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
        const double x = R[0][0] * X[0] + R[0][1] * X[1] + R[0][2] * X[2] + P[0];
        const double y = R[1][0] * X[0] + R[1][1] * X[1] + R[1][2] * X[2] + P[1];
        const double z = R[2][0] * X[0] + R[2][1] * X[1] + R[2][2] * X[2] + P[2];
        const double L = std::sqrt((x * x + y * y) + z * z);
        const double rxz = x * x + z * z;
        const double theta = ovs_det_atan2(x, z);
        const double phi = -ovs_det_asin(y / L);
        const double e0 = ed.obs_x - cols * (0.5 + theta / (2.0 * kPi));
        const double e1 = ed.obs_y - rows * (0.5 - phi / kPi);
        const double w = ed.inv_sigma_sq;
        const double c2 = w * (e0 * e0 + e1 * e1);
        double rho0 = c2, rho1 = 1.0;
        if (huber_delta > 0 && c2 > dsqr) {
            const double sq = std::sqrt(c2);
            rho0 = 2 * sq * huber_delta - dsqr;
            rho1 = huber_delta / sq;
        }
        chi2[0] += c2;
        chi2[1] += rho0;
        const double a0 = -((double)cols / (2.0 * kPi)) * (1.0 / rxz);
        const double a1 = -((double)rows / kPi) * (1.0 / (L * std::sqrt(rxz)));
        double Jl[2][3], Jp[2][6];
        auto col = [&](double dx, double dy, double dz, double& j0, double& j1) {
            const double dL = (1.0 / L) * ((x * dx + y * dy) + z * dz);
            j0 = a0 * (z * dx - x * dz);
            j1 = a1 * (L * dy - y * dL);
        };
        col(0.0, -z, y, Jp[0][0], Jp[1][0]);
        col(z, 0.0, -x, Jp[0][1], Jp[1][1]);
        col(-y, x, 0.0, Jp[0][2], Jp[1][2]);
        col(1.0, 0.0, 0.0, Jp[0][3], Jp[1][3]);
        col(0.0, 1.0, 0.0, Jp[0][4], Jp[1][4]);
        col(0.0, 0.0, 1.0, Jp[0][5], Jp[1][5]);
        for (int c = 0; c < 3; ++c) col(R[0][c], R[1][c], R[2][c], Jl[0][c], Jl[1][c]);
        const double W = rho1 * w;
        const double r0 = -W * e0, r1 = -W * e1;
        double* hl = Hll + 9 * ed.point_idx;
        double* gl = bl + 3 * ed.point_idx;
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) hl[3 * a + b] += W * (Jl[0][a] * Jl[0][b] + Jl[1][a] * Jl[1][b]);
            gl[a] += Jl[0][a] * r0 + Jl[1][a] * r1;
        }
        if (!(pose_fixed && pose_fixed[ed.pose_idx])) {
            double* hp = Hpp + 36 * ed.pose_idx;
            double* gp = bp + 6 * ed.pose_idx;
            double* hpl = Hpl + 18 * (size_t)e;
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) hp[6 * a + b] += W * (Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b]);
                gp[a] += Jp[0][a] * r0 + Jp[1][a] * r1;
                for (int b = 0; b < 3; ++b) hpl[3 * a + b] = W * (Jp[0][a] * Jl[0][b] + Jp[1][a] * Jl[1][b]);
            }
        }
    }
    return 0;
}

// per-edge chi2 = e^T Omega e of the equirectangular edge at (poses, points) -- what g2o's edge->chi2() returns after computeError(); used by
// the local-BA restatement (lba.py) for the outlier gates, with the SAME asin / atan2 as the linearisation above
int ovo_ba_edge_chi2_equirect(const double* poses, int n_pose, const double* points, int n_pt, const ovo_ba_edge* edges, int n_edge, int cols,
                              int rows, double* chi2) {
    const double kPi = 3.14159265358979323846;
    for (int e = 0; e < n_edge; ++e) {
        const ovo_ba_edge& ed = edges[e];
        if (ed.pose_idx < 0 || ed.pose_idx >= n_pose || ed.point_idx < 0 || ed.point_idx >= n_pt) return -1;
        const double* P = poses + 7 * ed.pose_idx;
        const double* X = points + 3 * ed.point_idx;
        const double qx = P[3], qy = P[4], qz = P[5], qw = P[6];
        const double tx2 = 2 * qx, ty2 = 2 * qy, tz2 = 2 * qz;
        const double twx = tx2 * qw, twy = ty2 * qw, twz = tz2 * qw;
        const double txx = tx2 * qx, txy = ty2 * qx, txz = tz2 * qx;
        const double tyy = ty2 * qy, tyz = tz2 * qy, tzz = tz2 * qz;
        const double x = (1 - (tyy + tzz)) * X[0] + (txy - twz) * X[1] + (txz + twy) * X[2] + P[0];
        const double y = (txy + twz) * X[0] + (1 - (txx + tzz)) * X[1] + (tyz - twx) * X[2] + P[1];
        const double z = (txz - twy) * X[0] + (tyz + twx) * X[1] + (1 - (txx + tyy)) * X[2] + P[2];
        const double L = std::sqrt((x * x + y * y) + z * z);
        const double theta = ovs_det_atan2(x, z);
        const double phi = -ovs_det_asin(y / L);
        const double e0 = ed.obs_x - cols * (0.5 + theta / (2.0 * kPi));
        const double e1 = ed.obs_y - rows * (0.5 - phi / kPi);
        chi2[e] = ed.inv_sigma_sq * (e0 * e0 + e1 * e1);
    }
    return 0;
}

typedef struct ovo_ba_edge_stereo {
    int32_t pose_idx, point_idx;
    double obs_x, obs_y, obs_x_right;
    double inv_sigma_sq;   // information = inv_level_sigma_sq[octave] * I3
} ovo_ba_edge_stereo;

// B1/B2 stereo variant: optimize::g2o::se3::stereo_perspective_reproj_edge (expected: src/openvslam/optimize/g2o/se3/
// perspective_reproj_edge.cc; identical to ORB-SLAM2 EdgeStereoSE3ProjectXYZ): e = (u, v, u_r) - pi(RX + t) with
// u_r = u - bf / z; row 2 of the Jacobians = row 0 corrected by the bf / z^2 terms. Same outputs / conventions as
// ovo_ba_linearize; `accumulate` != 0 adds into the given blocks instead of zeroing them (mixed mono + stereo graphs).
int ovo_ba_linearize_stereo(const double* poses, const uint8_t* pose_fixed, int n_pose, const double* points, int n_pt,
                            const ovo_ba_edge_stereo* edges, int n_edge, const ovo_ba_cam* cam, double bf, double huber_delta,
                            int accumulate, double* Hpp, double* bp, double* Hll, double* bl, double* Hpl, double* chi2) {
    if (!accumulate) {
        std::memset(Hpp, 0, sizeof(double) * 36 * n_pose);
        std::memset(bp, 0, sizeof(double) * 6 * n_pose);
        std::memset(Hll, 0, sizeof(double) * 9 * n_pt);
        std::memset(bl, 0, sizeof(double) * 3 * n_pt);
        chi2[0] = chi2[1] = 0.0;
    }
    std::memset(Hpl, 0, sizeof(double) * 18 * n_edge);
    const double dsqr = huber_delta * huber_delta;
    for (int e = 0; e < n_edge; ++e) {
        const ovo_ba_edge_stereo& ed = edges[e];
        if (ed.pose_idx < 0 || ed.pose_idx >= n_pose || ed.point_idx < 0 || ed.point_idx >= n_pt) return -1;
        const double* P = poses + 7 * ed.pose_idx;
        const double* X = points + 3 * ed.point_idx;
        const double qx = P[3], qy = P[4], qz = P[5], qw = P[6];
        const double tx2 = 2 * qx, ty2 = 2 * qy, tz2 = 2 * qz;
        const double twx = tx2 * qw, twy = ty2 * qw, twz = tz2 * qw;
        const double txx = tx2 * qx, txy = ty2 * qx, txz = tz2 * qx;
        const double tyy = ty2 * qy, tyz = tz2 * qy, tzz = tz2 * qz;
        const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
        const double x = R[0][0] * X[0] + R[0][1] * X[1] + R[0][2] * X[2] + P[0];
        const double y = R[1][0] * X[0] + R[1][1] * X[1] + R[1][2] * X[2] + P[1];
        const double z = R[2][0] * X[0] + R[2][1] * X[1] + R[2][2] * X[2] + P[2];
        const double invz = 1.0 / z, invz2 = invz * invz;
        const double u = cam->fx * x * invz + cam->cx;
        double er[3];
        er[0] = ed.obs_x - u;
        er[1] = ed.obs_y - (cam->fy * y * invz + cam->cy);
        er[2] = ed.obs_x_right - (u - bf * invz);
        const double w = ed.inv_sigma_sq;
        const double c2 = w * ((er[0] * er[0] + er[1] * er[1]) + er[2] * er[2]);
        double rho0 = c2, rho1 = 1.0;
        if (huber_delta > 0 && c2 > dsqr) {
            const double sq = std::sqrt(c2);
            rho0 = 2 * sq * huber_delta - dsqr;
            rho1 = huber_delta / sq;
        }
        chi2[0] += c2;
        chi2[1] += rho0;
        double Jl[3][3], Jp[3][6];
        for (int c = 0; c < 3; ++c) {
            Jl[0][c] = -invz * (cam->fx * R[0][c] - cam->fx * x * invz * R[2][c]);
            Jl[1][c] = -invz * (cam->fy * R[1][c] - cam->fy * y * invz * R[2][c]);
            Jl[2][c] = Jl[0][c] - bf * R[2][c] * invz2;
        }
        Jp[0][0] = x * y * invz2 * cam->fx;
        Jp[0][1] = -(1 + x * x * invz2) * cam->fx;
        Jp[0][2] = y * invz * cam->fx;
        Jp[0][3] = -invz * cam->fx;
        Jp[0][4] = 0;
        Jp[0][5] = x * invz2 * cam->fx;
        Jp[1][0] = (1 + y * y * invz2) * cam->fy;
        Jp[1][1] = -x * y * invz2 * cam->fy;
        Jp[1][2] = -x * invz * cam->fy;
        Jp[1][3] = 0;
        Jp[1][4] = -invz * cam->fy;
        Jp[1][5] = y * invz2 * cam->fy;
        Jp[2][0] = Jp[0][0] - bf * y * invz2;
        Jp[2][1] = Jp[0][1] + bf * x * invz2;
        Jp[2][2] = Jp[0][2];
        Jp[2][3] = Jp[0][3];
        Jp[2][4] = 0;
        Jp[2][5] = Jp[0][5] - bf * invz2;
        const double W = rho1 * w;
        const double r[3] = {-W * er[0], -W * er[1], -W * er[2]};
        double* hl = Hll + 9 * ed.point_idx;
        double* gl = bl + 3 * ed.point_idx;
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) hl[3 * a + b] += W * ((Jl[0][a] * Jl[0][b] + Jl[1][a] * Jl[1][b]) + Jl[2][a] * Jl[2][b]);
            gl[a] += (Jl[0][a] * r[0] + Jl[1][a] * r[1]) + Jl[2][a] * r[2];
        }
        if (!(pose_fixed && pose_fixed[ed.pose_idx])) {
            double* hp = Hpp + 36 * ed.pose_idx;
            double* gp = bp + 6 * ed.pose_idx;
            double* hpl = Hpl + 18 * (size_t)e;
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) hp[6 * a + b] += W * ((Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b]) + Jp[2][a] * Jp[2][b]);
                gp[a] += (Jp[0][a] * r[0] + Jp[1][a] * r[1]) + Jp[2][a] * r[2];
                for (int b = 0; b < 3; ++b) hpl[3 * a + b] = W * ((Jp[0][a] * Jl[0][b] + Jp[1][a] * Jl[1][b]) + Jp[2][a] * Jl[2][b]);
            }
        }
    }
    return 0;
}

}   // extern "C"
