// ba_host_math.h -- host-side pieces of optimize::local_bundle_adjuster::optimize that stay on the CPU (BASELINE north star: "the sparse
// Schur solve staying on the host"): SE3 update (g2o SE3Quat::exp), rotation <-> quaternion as Eigen / g2o do them, the dense Cholesky of
// the reduced camera system (at most 6 n_pose square).
#pragma once
#include <cmath>
#include <vector>

namespace ovs_ba_host {

struct Pose {   // world -> camera, rotation matrix row-major + translation
    double R[9], t[3];
};

inline void quat_to_rot(const double* q, double* R) {   // q = (x, y, z, w)
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

inline void rot_to_quat(const double* R, double* q) {   // Eigen's Quaternion(Matrix3) branches, then w >= 0 and unit norm (SE3Quat::normalizeRotation)
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
inline void se3_oplus(Pose& T, const double* u) {
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

inline bool inv3_sym(const double* H, double lambda, double* out) {   // (H + lambda I)^-1 by cofactors
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
static inline __attribute__((always_inline)) bool cholesky_solve_body(std::vector<double>& A, int n, std::vector<double>& b) {
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

// The same loop nest compiled twice: for the x86-64 baseline (SSE2) and for AVX2 (4 doubles per operation; products and differences stay
// individually rounded: no FMA, -ffp-contract=off); chosen once at run time. 288 x 288 at config 5: 0.66 -> 0.47 ms per solve. (A panel-blocked version on a 4-thread pool was
// measured too: bit-identical, but no faster on hosts whose cores are shared, so the solve stays single-threaded.)
__attribute__((target("avx2"))) inline bool cholesky_solve_avx2(std::vector<double>& A, int n, std::vector<double>& b) {
    return cholesky_solve_body(A, n, b);
}
inline bool cholesky_solve_generic(std::vector<double>& A, int n, std::vector<double>& b) { return cholesky_solve_body(A, n, b); }
inline bool cholesky_solve(std::vector<double>& A, int n, std::vector<double>& b) {
    static const bool has_avx2 = __builtin_cpu_supports("avx2");
    return has_avx2 ? cholesky_solve_avx2(A, n, b) : cholesky_solve_generic(A, n, b);
}

}   // namespace ovs_ba_host
