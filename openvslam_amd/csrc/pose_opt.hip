// pose_opt.hip -- optimize::pose_optimizer::optimize (expected: src/openvslam/optimize/pose_optimizer.{h,cc},
// optimize/g2o/se3/{perspective_pose_opt_edge.*, shot_vertex.*}; g2o OptimizationAlgorithmLevenberg + RobustKernelHuber): the
// per-frame, pose-only bundle adjustment of the tracking thread. It runs right after the matchers on every
// frame, so it belongs on the device with them: ONE launch does all 4 rounds x 10 Levenberg-Marquardt iterations.
//
// One 256-thread workgroup per frame. The <= 8192 observations stay in HBM/L2 (64 B each) and are re-read every pass; a thread owns
// observations tid, tid + 256, ... and keeps their inlier flags in a 32-bit register mask. Per LM iteration: every thread
// accumulates the 21 + 6 normal-equation terms and the robustified chi2 of its observations, a fixed-order block reduction
// (wave shuffles, then waves 0..3) produces H, b, chi; thread 0 solves the 6x6 system by Cholesky, applies exp(dx) * T and the
// workgroup evaluates the trial pose's chi2 -- g2o's accept / reject / lambda schedule as restated by the CPU checker under oracle/.
// fp64 throughout; per-observation quantities follow the oracle's operation order, the SUMS are associated differently (tree vs
// sequential), so parity with the oracle is to a stated tolerance (tests: pose 1e-9, identical inlier flags away from the chi2 gates).
#include <algorithm>
#include <cmath>
#include <cstring>

#include <cstdlib>

#include <type_traits>

#include "ovs_common.h"

namespace ovs {

struct PoseD {
    double R[9], t[3];
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// a workgroup-uniform double (read from LDS by every lane) moved to scalar registers: the pose of a sweep costs 24 vector registers otherwise,
// which the 512-thread build (256 registers per lane) did not have -- it kept part of the linearisation's state in scratch memory
__device__ __forceinline__ double uniform_d(double v) {
    union {
        double d;
        int i[2];
    } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.d;
}

__device__ void se3_exp_d(const double* u, PoseD& out) {
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = sqrt((wx * wx + wy * wy) + wz * wz);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = (O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j]) + O[3 * i + 2] * O[6 + j];
    double V[9];
    const bool small = theta < 0.00001;
    // the three coefficients once (the expressions of upstream's Sophus-style exp, evaluated per matrix entry there)
    double sn = 0.0, cs = 1.0;
    if (!small) sincos(theta, &sn, &cs);   // one argument reduction, one pass over the two polynomials (sin() and cos() each evaluate both)
    // one division for the three coefficients sin / theta, (1 - cos) / theta^2, (theta - sin) / theta^3 (round 4: three IEEE divides were ~90
    // dependent instructions of the one lane every trial waits for; the products differ from the quotients by an ulp)
    const double it = small ? 0.0 : 1.0 / theta, it2 = it * it;
    const double ka = sn * it, kb = (1 - cs) * it2, kc = (theta - sn) * (it2 * it);
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        if (small) {
            out.R[i] = (I + O[i]) + O2[i];
            V[i] = out.R[i];
        } else {
            out.R[i] = (I + ka * O[i]) + kb * O2[i];
            V[i] = (I + kb * O[i]) + kc * O2[i];
        }
    }
    for (int i = 0; i < 3; ++i) out.t[i] = (V[3 * i] * u[3] + V[3 * i + 1] * u[4]) + V[3 * i + 2] * u[5];
}

__device__ void compose_d(const PoseD& a, const PoseD& b, PoseD& out) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out.R[3 * i + j] = (a.R[3 * i] * b.R[j] + a.R[3 * i + 1] * b.R[3 + j]) + a.R[3 * i + 2] * b.R[6 + j];
        out.t[i] = ((a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1]) + a.R[3 * i + 2] * b.t[2]) + a.t[i];
    }
}

// (H + lambda I) x = b by Cholesky. One thread runs this between two barriers of a one-workgroup kernel, so its dependent chain is the
// kernel's: the 27 divisions by diagonal entries are 6 reciprocal square roots (v_rsq_f64 + Newton, round 4: ~15 dependent operations
// instead of an IEEE sqrt and an IEEE divide of ~55) and 27 multiplications; the products differ from the quotients by an ulp or two, far
// inside the optimiser's 1e-9 tolerance against the oracle.
__device__ bool solve6_d(const double* H, double lambda, const double* b, double* x) {
    double L[36], inv[6];
    for (int i = 0; i < 36; ++i) L[i] = 0;
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = H[6 * i + j] + (i == j ? lambda : 0.0);
            for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
            if (i == j) {
                if (!(s > 0)) return false;
                inv[i] = rsqrt_newton(s);   // (L[i][i] itself is never read: every use below is a multiplication by its reciprocal)
            } else {
                L[6 * i + j] = s * inv[j];
            }
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
        y[i] = s * inv[i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
        x[i] = s * inv[i];
    }
    return true;
}

// chi2 of one observation; with acc != nullptr also its 21 (upper triangle, row-major) + 6 + 1 contributions: H, b, robust chi2
// equirectangular_pose_opt_edge (expected: src/openvslam/optimize/g2o/se3/equirectangular_pose_opt_edge.{h,cc}): cam = {cols, rows, -, -};
// the operation order of the CPU checker (and of the pose part of k_ba_linearize's equirectangular model); asin / atan2 from ovs_detmath.h
__device__ __forceinline__ double pose_edge_equirect(const double* R, const double* t, const ovs_pose_obs& o, const ovs_ba_cam& cam, double delta,
                                                     double* acc) {
    const double x = ((R[0] * o.pos_w[0] + R[1] * o.pos_w[1]) + R[2] * o.pos_w[2]) + t[0];
    const double y = ((R[3] * o.pos_w[0] + R[4] * o.pos_w[1]) + R[5] * o.pos_w[2]) + t[1];
    const double z = ((R[6] * o.pos_w[0] + R[7] * o.pos_w[1]) + R[8] * o.pos_w[2]) + t[2];
    const double kPi = 3.14159265358979323846;
    const double cols = cam.fx, rows = cam.fy;
    const double L = sqrt((x * x + y * y) + z * z);
    const double rxz = x * x + z * z;
    const double theta = ovs_det_atan2(x, z);
    const double phi = -ovs_det_asin(y / L);
    const double e0 = o.obs_x - cols * (0.5 + theta / (2.0 * kPi));
    const double e1 = o.obs_y - rows * (0.5 - phi / kPi);
    const double c2 = o.inv_sigma_sq * (e0 * e0 + e1 * e1);
    if (!acc) return c2;
    double rho0 = c2, rho1 = 1.0;
    const double dsqr = delta * delta;
    if (delta > 0 && c2 > dsqr) {
        const double sq = sqrt(c2);
        rho0 = 2 * sq * delta - dsqr;
        rho1 = delta / sq;
    }
    const double a0 = -(cols / (2.0 * kPi)) * (1.0 / rxz);
    const double a1 = -(rows / kPi) * (1.0 / (L * sqrt(rxz)));
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
    // H += J^T (W J), b -= (W J)^T e: the weighted rows A = W J once (12 products), then fused multiply-adds straight into the accumulators --
    // 54 operations per observation instead of 105 (round 5; separate products and sums until then). The sums associate differently from the
    // oracle's (its per-entry order is W * (J0a J0b + J1a J1b)): differences of an ulp per term, inside the optimiser's 1e-9 tolerance.
    double A0[6], A1[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        A0[a] = W * J[0][a];
        A1[a] = W * J[1][a];
    }
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = a; b < 6; ++b, ++k) acc[k] = __builtin_fma(A0[a], J[0][b], __builtin_fma(A1[a], J[1][b], acc[k]));
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] = __builtin_fma(-A0[a], e0, __builtin_fma(-A1[a], e1, acc[21 + a]));
    acc[27] += rho0;
    return c2;
}

// STEREO = false: the frame holds no stereo observation (every monocular frame), so the third residual row is not even evaluated under a
// false predicate -- hipcc if-converts `if (st)` into ~100 unconditional instructions plus 28 selects per observation, a third of the edge
template <int MODEL, bool STEREO>
__device__ __forceinline__ double pose_edge_impl(const double* R, const double* t, const ovs_pose_obs& o, const ovs_ba_cam& cam, double bf,
                                                 double delta, double* acc) {
    if (MODEL == 1) return pose_edge_equirect(R, t, o, cam, delta, acc);
    const double x = ((R[0] * o.pos_w[0] + R[1] * o.pos_w[1]) + R[2] * o.pos_w[2]) + t[0];
    const double y = ((R[3] * o.pos_w[0] + R[4] * o.pos_w[1]) + R[5] * o.pos_w[2]) + t[1];
    const double z = ((R[6] * o.pos_w[0] + R[7] * o.pos_w[1]) + R[8] * o.pos_w[2]) + t[2];
    const double invz = 1.0 / z, invz2 = invz * invz;
    const bool st = STEREO && o.is_stereo != 0;
    const double u = cam.fx * x * invz + cam.cx;
    const double e0 = o.obs_x - u;
    const double e1 = o.obs_y - (cam.fy * y * invz + cam.cy);
    const double e2 = st ? o.obs_x_right - (u - bf * invz) : 0.0;
    double ss = e0 * e0 + e1 * e1;
    if (st) ss = ss + e2 * e2;
    const double c2 = o.inv_sigma_sq * ss;
    if (!acc) return c2;
    double rho0 = c2, rho1 = 1.0;
    const double dsqr = delta * delta;
    if (delta > 0 && c2 > dsqr) {
        const double sq = sqrt(c2);
        rho0 = 2 * sq * delta - dsqr;
        rho1 = delta / sq;
    }
    double J[3][6];
    J[0][0] = x * y * invz2 * cam.fx;
    J[0][1] = -(1 + x * x * invz2) * cam.fx;
    J[0][2] = y * invz * cam.fx;
    J[0][3] = -invz * cam.fx;
    J[0][4] = 0;
    J[0][5] = x * invz2 * cam.fx;
    J[1][0] = (1 + y * y * invz2) * cam.fy;
    J[1][1] = -x * y * invz2 * cam.fy;
    J[1][2] = -x * invz * cam.fy;
    J[1][3] = 0;
    J[1][4] = -invz * cam.fy;
    J[1][5] = y * invz2 * cam.fy;
    J[2][0] = J[0][0] - bf * y * invz2;
    J[2][1] = J[0][1] + bf * x * invz2;
    J[2][2] = J[0][2];
    J[2][3] = J[0][3];
    J[2][4] = 0;
    J[2][5] = J[0][5] - bf * invz2;
    const double W = rho1 * o.inv_sigma_sq;
    // J[0][4], J[1][3] and J[2][4] are exact zeros: their terms are simply not issued (0 * x is not 0 for x = inf / NaN, which the edges never
    // hold, so the compiler may not drop them itself). H += J^T (W J), b -= (W J)^T e with the weighted rows A = W J formed once and fused
    // multiply-adds straight into the accumulators: 52 operations per monocular observation instead of 105 (round 5; DESIGN 7.3 of round 4).
    // The sums associate differently from the oracle's W * (J0a J0b + J1a J1b): an ulp per term, inside the optimiser's 1e-9 tolerance.
    double A0[6], A1[6], A2[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        A0[a] = W * J[0][a];
        A1[a] = W * J[1][a];
        A2[a] = st ? W * J[2][a] : 0.0;
    }
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = a; b < 6; ++b, ++k) {
            const bool z0 = a == 4 || b == 4, z1 = a == 3 || b == 3;   // row 0 (and row 2 like it) / row 1 contribute nothing
            if (z0 && z1) continue;                                     // (3, 4): every product is a zero, the sum stays +0
            double v = acc[k];
            if (!z1) v = __builtin_fma(A1[a], J[1][b], v);
            if (!z0) v = __builtin_fma(A0[a], J[0][b], v);
            if (st && !z0) v = __builtin_fma(A2[a], J[2][b], v);
            acc[k] = v;
        }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        double v = acc[21 + a];
        if (a != 3) v = __builtin_fma(-A1[a], e1, v);
        if (a != 4) v = __builtin_fma(-A0[a], e0, v);
        if (st && a != 4) v = __builtin_fma(-A2[a], e2, v);
        acc[21 + a] = v;
    }
    acc[27] += rho0;
    return c2;
}

constexpr int kPoseMaxObs = 8192;   // <= 32 observations per thread: the inlier flags of a thread fit one register
// one workgroup per frame, kPoseThreads threads (template parameter: 256 / 512 / 1024, chosen per launch)

// KREG (round 6): 2 = the launcher knows that no thread owns more than two observations (n <= 2 G kPoseThreads: every tracked frame on the
// G > 1 path): a thread then loads ITS two 64-byte records once, before the first round, and every one of the ~50 passes of a call reads
// them from registers -- a pass started with a dependent trip to L2 (~1 us of a ~6 us iteration). 0 = records re-read from memory per pass.
template <int MODEL, int kPoseThreads, int KREG>   // MODEL 0 perspective (mono / stereo edges), 1 equirectangular (mono edges)
__global__ __launch_bounds__(kPoseThreads) void k_pose_optimize(const double* __restrict__ poses_in, const ovs_pose_obs* __restrict__ obs_all,
                                                      const int32_t* __restrict__ obs_offsets, ovs_ba_cam cam, double bf, int setup_type,
                                                      double* __restrict__ poses_out, uint8_t* __restrict__ outlier_all,
                                                      int32_t* __restrict__ num_valid, int reset_each_round, int G,
                                                      unsigned long long* __restrict__ gpart, unsigned int epoch0, int batch_retries,
                                                      int stereo_hint) {
    // G > 1 (round 4, single-frame latency path): the frame's observations are spread over G workgroups (blockIdx.y). Every sum over the
    // observations is then a sum of G workgroup partials exchanged through memory behind a grid-wide barrier; each workgroup adds them in
    // the same order, solves the same 6 x 6 system and takes the same branches, so there is no second barrier and no broadcast.
    // gpart: [frame][2][G][56] 64-bit words (two epochs; see exchange()).
    constexpr int kPoseWaves = kPoseThreads / 64;
    constexpr int kRedPitch = kPoseThreads + 8;   // doubles per row of the reduction scratch
    __shared__ double s_part[kPoseWaves][28];
    extern __shared__ __attribute__((aligned(16))) double s_red[];   // [28 * kRedPitch]: the linearisation's block reduction (dynamic: 116 KB at 512 threads)
    __shared__ double s_sum[28];
    __shared__ double s_sys[28];   // the current iteration's system (copied from s_sum, which the trial reductions overwrite)
    __shared__ PoseD s_T, s_Tn;
    __shared__ PoseD s_Tq[9];              // trial poses of an iteration's retries 1 .. 9 (evaluated together, see below)
    __shared__ double s_okq[9], s_scq[9];  // ... whether the solve succeeded, and the gain ratio's denominator
    __shared__ double s_ctl[4];   // [0] = continue trials of this iteration, [1] = continue iterations of this round
    __shared__ int s_cnt[kPoseWaves];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // G > 1 is launched for ONE frame with 8 G workgroups of which every eighth works: consecutive workgroups are dealt round-robin to the
    // eight XCDs, so the G workers share one XCD's L2 and the barrier's atomics and the partials never cross the fabric
    if (G > 1 && (blockIdx.y & 7u) != 0u) return;
    const int p = blockIdx.x, grp = G > 1 ? (int)(blockIdx.y >> 3) : 0;
    const int gtid = grp * kPoseThreads + tid, gstride = G * kPoseThreads;   // this thread's observations: gtid, gtid + gstride, ...
    unsigned long long* const my_part = gpart ? gpart + (size_t)p * 2 * G * 56 : nullptr;
    unsigned int epoch = epoch0;   // epoch0 + exchanges passed (workgroup-uniform, the same in every workgroup of the frame)
    __shared__ int s_abort;
    if (tid == 0) s_abort = 0;
    // exchange nv <= 28 workgroup sums (in s_sum) between the frame's workgroups: afterwards s_sum holds the sums over all of them, added in
    // workgroup order. Returns false when the barrier was abandoned (a workgroup of the frame never arrived within 50 ms: the launch is
    // reported as failed and the caller falls back to one workgroup per frame).
    auto exchange = [&](int nv) -> bool {
        if (G == 1) return true;
        // Round 6: no arrival counter. Every partial travels as two 64-bit words {flag = epoch : 32 | half of the double : 32} (a 64-bit store is
        // single-copy atomic, so a word whose flag is this epoch carries this epoch's data: no release fence, no second trip to fetch the data
        // after the barrier). Round 4-5: [28 stores, barrier, fetch_add with release, spin on the counter, barrier, load the partials, barrier]
        // = three dependent trips to L2 per exchange, ~2 us of a ~5.5 us iteration. Now: stores, then every thread polls ITS word of the
        // G x 2 nv -- one trip after the last workgroup's store lands. Two buffers by epoch parity: a workgroup writes epoch e + 2 into the
        // buffer of epoch e only after it has read every workgroup's e + 1 words, which those wrote after reading epoch e. Epochs start at the
        // call's epoch0 (host: + 4096 per call), so words left by an earlier call never match.
        ++epoch;
        unsigned long long* const buf = my_part + (size_t)(epoch & 1u) * G * 56;
        if (tid < 2 * nv) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(s_sum[tid >> 1]);
            const unsigned long long half = (tid & 1) ? (bits >> 32) : (bits & 0xffffffffull);
            __hip_atomic_store(&buf[grp * 56 + tid], ((unsigned long long)epoch << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t* const s_w = reinterpret_cast<uint32_t*>(s_red);   // [G][56] halves (s_red is free here: the linearisation's reduction is over)
        const unsigned long long t0 = wall_clock64();
        for (int w = tid; w < 2 * nv * G; w += kPoseThreads) {
            const int g = w / (2 * nv), j = w - g * (2 * nv);
            unsigned long long v;
            unsigned int spins = 0;
            while ((uint32_t)((v = __hip_atomic_load(&buf[g * 56 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch) {
                if ((++spins & 63u) == 0u && wall_clock64() - t0 > 5000000ull) {   // a workgroup of the frame never arrived within 50 ms
                    s_abort = 1;
                    break;
                }
            }
            s_w[g * 56 + j] = (uint32_t)v;
        }
        __syncthreads();
        if (s_abort) return false;
        if (tid < nv) {   // the sums in workgroup order, the same in every workgroup
            const double* const s_d = reinterpret_cast<const double*>(s_w);
            double v = s_d[tid];
            for (int g = 1; g < G; ++g) v += s_d[g * 28 + tid];
            s_sum[tid] = v;
        }
        __syncthreads();
        return true;
    };
#define POSE_EXCHANGE(nv)                                        \
    if (!exchange(nv)) {                                         \
        if (tid == 0 && grp == 0) num_valid[p] = -2;             \
        return;                                                  \
    }
    const int o0 = obs_offsets[p], n = obs_offsets[p + 1] - o0;
    const ovs_pose_obs* obs = obs_all + o0;
    uint8_t* outlier = outlier_all + o0;
    // this thread's observations gtid, gtid + gstride, ...: body(k, i, record). KREG: the first two from registers (and there are no others)
    ovs_pose_obs o_r0 = {}, o_r1 = {};
    if (KREG) {
        if (gtid < n) o_r0 = obs[gtid];
        if (gtid + gstride < n) o_r1 = obs[gtid + gstride];
    }
    auto for_each_obs = [&](auto&& body) __attribute__((always_inline)) {
        if (KREG) {
            if (gtid < n) body(0, gtid, o_r0);
            if (gtid + gstride < n) body(1, gtid + gstride, o_r1);
        } else {
            for (int k = 0, i = gtid; i < n; i += gstride, ++k) body(k, i, obs[i]);
        }
    };
    // upstream: ONE Huber delta per frame, chosen by the rig (Monocular -> sqrt_chi_sq_2D, otherwise sqrt_chi_sq_3D); the chi-square
    // outlier gates stay per edge
    // upstream: constexpr float chi_sq_2D = 5.99146; const float sqrt_chi_sq_2D = std::sqrt(chi_sq_2D); (3D: 7.81473) -- FLOAT constants widened
    // to double where g2o consumes them (ORACLE_SPEC rule 25); hex literals so no library sqrt is involved
    const double kChi2D = 0x1.7f7414p+2, kChi3D = 0x1.f4248ap+2, kSqrtChi2D = 0x1.394fbcp+1, kSqrtChi3D = 0x1.65d26ap+1;
    const double huber = setup_type == 0 ? kSqrtChi2D : kSqrtChi3D;
    if (n > kPoseMaxObs) {   // the per-thread inlier mask holds 32 observations: refuse instead of aliasing flags
        if (tid == 0) num_valid[p] = -1;
        return;
    }

    // fixed-order block reduction of the 28 per-thread sums of a linearisation into s_sum, through LDS: 28 separate wave reductions (six
    // dependent cross-lane steps each) were a quarter of an iteration's time on this one-workgroup, latency-bound kernel. Every thread
    // stores its 28 values (row i = value i, one column per thread), then thread (i, part) = (tid >> 3, tid & 7) adds the 32 columns
    // part, part + 8, ... of row i in that order and the eight parts are combined by three exchange steps. Rows are kRedPitch doubles apart:
    // 528 dwords = 16 banks, so the four rows a 32-lane group reads fall on disjoint banks.
    auto reduce28_finish = [&]() {   // (the caller has stored its 28 values: s_red[i * kRedPitch + tid] = v[i])
        __syncthreads();
        if (tid < 28 * 8) {
            const int i = tid >> 3, part = tid & 7;
            const double* row = s_red + i * kRedPitch + part;
            // sixteen LDS reads in flight per batch: left to itself hipcc (at the 256-register ceiling of the 512-thread build) reused one
            // register pair for every read, i.e. 32 dependent LDS round trips -- 8 960 cycles per reduction, a quarter of the kernel
            double s = 0;
#pragma unroll
            for (int k0 = 0; k0 < kPoseThreads / 8; k0 += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = row[8 * (k0 + u)];
                asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                             "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
#pragma unroll
                for (int u = 0; u < 16; ++u) s = (k0 + u == 0) ? v[u] : s + v[u];   // (the same order as ever: columns part, part + 8, ...)
            }
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            if (part == 0) s_sum[i] = s;
        }
        __syncthreads();
    };
    // fixed-order block reduction of NV per-thread values into s_sum
    auto reduce = [&](const double* v, int nv) {
        for (int i = 0; i < nv; ++i) {
            const double s = wave_sum_d(v[i]);
            if (lane == 0) s_part[wv][i] = s;
        }
        __syncthreads();
        if (tid < nv) {   // fixed order: wave 0, 1, 2, ...
            double s = s_part[0][tid];
            for (int w = 1; w < kPoseWaves; ++w) s += s_part[w][tid];
            s_sum[tid] = s;
        }
        __syncthreads();
    };

    // the input pose is read where it is needed (here and, under reset_each_round, at the start of a round): held in registers across the
    // kernel it cost 24 of them for nothing -- the 512-thread build kept it in scratch memory
    auto load_input_pose = [&]() {
        for (int i = 0; i < 9; ++i) s_T.R[i] = poses_in[12 * (size_t)p + i];
        for (int i = 0; i < 3; ++i) s_T.t[i] = poses_in[12 * (size_t)p + 9 + i];
    };
    uint32_t active = 0xFFFFFFFFu;   // bit k <-> observation gtid + gstride * k
    int st_any = 0;
    for (int i = gtid; i < n; i += gstride) outlier[i] = 0;
    // stereo_hint: -1 = find out (every workgroup scans the whole frame: the flag must be the same in all); 0 / 1 = the host has looked (one-frame
    // calls: with the records in host memory G scans of them would cross PCIe)
    if (MODEL == 0 && stereo_hint < 0)
        for (int i = tid; i < n; i += kPoseThreads) st_any |= obs[i].is_stereo;
    const bool has_stereo = MODEL == 0 && (stereo_hint >= 0 ? stereo_hint != 0 : __builtin_amdgcn_readfirstlane(__syncthreads_or(st_any)) != 0);
    if (tid == 0) load_input_pose();
    __syncthreads();
    int num_bad = 0;
    if (n >= 5) {
        for (int trial = 0; trial < 4; ++trial) {
            // Huber in rounds 0..2 (`if (trial == num_trials_ - 2) setRobustKernel(nullptr)` runs after round 2's optimisation); the
            // estimate carries over from round to round (the frame vertex is initialised once, before the loop)
            const bool robust = trial < 3;
            if (reset_each_round) {   // rule 25 (iv)'s alternative (ORB-SLAM2): every round starts from the input pose again (kernel argument: uniform)
                if (tid == 0) load_input_pose();
                __syncthreads();
            }
            double lambda = 0, ni = 2;   // held identically by every thread (all control flow below is workgroup-uniform)
            bool err_at_trial = false;   // active edges' errors were last computed at s_Tn (g2o leaves them stale after a rejected step)
            // An iteration's FIRST trial evaluates the whole system at the trial state, not just its chi2: when the step is accepted -- the
            // common case -- that IS the next iteration's linearisation (same state, same operations, same bits), which then starts without a
            // pass over the observations and without its reduction. Retries after a rejection evaluate chi2 only.
            bool have_lin = false;       // s_sum holds the system at s_T
            auto linearise_at = [&](bool at_trial) __attribute__((always_inline)) -> bool {   // state s_Tn / s_T -> s_sum[0 .. 27] (H upper triangle, b, robust chi2)
                double acc[28];
#pragma unroll
                for (int i = 0; i < 28; ++i) acc[i] = 0;
                double R[9], t[3];
                for (int i = 0; i < 9; ++i) R[i] = uniform_d(at_trial ? s_Tn.R[i] : s_T.R[i]);
                for (int i = 0; i < 3; ++i) t[i] = uniform_d(at_trial ? s_Tn.t[i] : s_T.t[i]);
                auto sweep = [&](auto stereo_tag) __attribute__((always_inline)) {
                    for_each_obs([&](int k, int, const ovs_pose_obs& o) __attribute__((always_inline)) {
                        if ((active >> k) & 1u) pose_edge_impl<MODEL, decltype(stereo_tag)::value>(R, t, o, cam, bf, robust ? huber : 0.0, acc);
                    });
                };
                if (has_stereo) sweep(std::true_type{});   // (workgroup-uniform)
                else sweep(std::false_type{});
#pragma unroll
                for (int i = 0; i < 28; ++i) s_red[i * kRedPitch + tid] = acc[i];
                reduce28_finish();
                return exchange(28);
            };
            for (int it = 0; it < 10; ++it) {
                err_at_trial = false;    // solve() starts with computeActiveErrors() at the current estimate
                if (!have_lin && !linearise_at(false)) {
                    if (tid == 0 && grp == 0) num_valid[p] = -2;
                    return;
                }
                have_lin = false;
                // the system stays in LDS (s_sys: H upper triangle row-major, b, chi2): only thread 0 needs it, for the solve, and 42 doubles
                // held by every thread across the trial's linearisation would not fit the register file of a 512-thread workgroup
                double current_chi = s_sum[27];
                if (tid < 28) s_sys[tid] = s_sum[tid];
                if (it == 0) {
                    double max_diag = 0;
                    const int dg[6] = {0, 6, 11, 15, 18, 20};   // (a, a) in the packed upper triangle
                    for (int j = 0; j < 6; ++j) max_diag = fmax(fabs(s_sum[dg[j]]), max_diag);
                    lambda = 1e-5 * max_diag;
                    ni = 2;
                }
                __syncthreads();   // s_sum is rewritten by the trial reductions below
                double rho = 0;
                int qmax = 0;
                do {
                    // (the 512-thread build never batches: the host picks 512 threads only for one-workgroup frames of 768 or more
                    // observations, where the one-by-one form measured faster, and without this block the build needs no scratch memory)
                    if (kPoseThreads == 256 && qmax == 1 && batch_retries) {
                        // ---- The first trial was rejected. g2o would now retry up to nine times with lambda * nu, nu doubling -- a sequence
                        // that does not depend on the outcomes, and (in a converged round, where the gain ratio is rounding noise) usually runs
                        // to the end: nine times [one lane's solve, a pass over the observations, a reduction, four barriers]. Round 4: the nine
                        // solves run on nine lanes at once, ONE pass evaluates the nine trial poses, one reduction returns the nine chi2, and
                        // every thread then replays g2o's accept / reject decisions over them in order. Per trial the arithmetic is the
                        // sequential form's, so are the results, bit for bit (OVS_POSE_BATCH_RETRIES=0 keeps the sequential form; tests compare).
                        if (tid < 9) {
                            double lam = lambda, nn = ni;
                            for (int j = 0; j < tid; ++j) {
                                lam *= nn;
                                nn *= 2;
                            }
                            double H[36], b[6], dxq[6] = {0, 0, 0, 0, 0, 0};
                            {
                                int k = 0;
                                for (int a = 0; a < 6; ++a)
                                    for (int c = a; c < 6; ++c) {
                                        H[6 * a + c] = s_sys[k];
                                        H[6 * c + a] = s_sys[k];
                                        ++k;
                                    }
                                for (int a = 0; a < 6; ++a) b[a] = s_sys[21 + a];
                            }
                            const bool okq = solve6_d(H, lam, b, dxq);
                            if (okq) {
                                PoseD E;
                                se3_exp_d(dxq, E);
                                compose_d(E, s_T, s_Tq[tid]);
                            }
                            s_okq[tid] = okq ? 1.0 : 0.0;
                            double scale = 0;
                            if (okq)
                                for (int j = 0; j < 6; ++j) scale += dxq[j] * (lam * dxq[j] + b[j]);
                            s_scq[tid] = scale + 1e-3;
                        }
                        __syncthreads();
                        double part[9];
#pragma unroll
                        for (int q = 0; q < 9; ++q) part[q] = 0;
#pragma unroll 1
                        for (int q = 0; q < 9; ++q) {
                            if (s_okq[q] == 0.0) continue;   // (workgroup-uniform)
                            double R[9], t[3];
                            for (int i = 0; i < 9; ++i) R[i] = s_Tq[q].R[i];
                            for (int i = 0; i < 3; ++i) t[i] = s_Tq[q].t[i];
                            double pq = 0;
                            auto sweep = [&](auto stereo_tag) __attribute__((always_inline)) {
                                for_each_obs([&](int k, int, const ovs_pose_obs& o) __attribute__((always_inline)) {
                                    if ((active >> k) & 1u) {
                                        const double c2 = pose_edge_impl<MODEL, decltype(stereo_tag)::value>(R, t, o, cam, bf, 0.0, nullptr);
                                        const double delta = robust ? huber : 0.0;
                                        double r = c2;
                                        if (delta > 0 && c2 > delta * delta) r = 2 * sqrt(c2) * delta - delta * delta;
                                        pq += r;
                                    }
                                });
                            };
                            if (has_stereo) sweep(std::true_type{});
                            else sweep(std::false_type{});
#pragma unroll
                            for (int u = 0; u < 9; ++u) part[u] = u == q ? pq : part[u];   // (no dynamic index into the register array)
                        }
                        reduce(part, 9);
                        POSE_EXCHANGE(9)
                        double lam = lambda, nn = ni;
                        int last_ok = -1;
                        bool accepted = false;
                        for (int q = 0; q < 9; ++q) {
                            const bool okq = s_okq[q] != 0.0;
                            const double temp_chi = okq ? s_sum[q] : 1.7976931348623157e308;
                            if (okq) last_ok = q;
                            rho = (current_chi - temp_chi) / s_scq[q];
                            ++qmax;
                            if (rho > 0 && isfinite(temp_chi)) {
                                const double tr = 2 * rho - 1;
                                double alpha = 1. - tr * tr * tr;   // (g2o: pow(2 rho - 1, 3); the library pow was ~150 instructions per lane for a cube)
                                alpha = fmin(alpha, 2.0 / 3.0);
                                lambda = lam * fmax(1.0 / 3.0, alpha);
                                ni = 2;
                                current_chi = temp_chi;
                                accepted = true;
                                break;
                            }
                            lam *= nn;
                            nn *= 2;
                            lambda = lam;
                            ni = nn;
                            if (!(rho < 0)) break;   // rho == 0 (or NaN): the sequential loop stops retrying here as well
                        }
                        __syncthreads();   // every thread has read s_sum / s_T before thread 0 moves the poses
                        if (last_ok >= 0) {
                            err_at_trial = true;   // the active edges' errors were last computed at this trial's pose
                            if (tid == 0) {
                                s_Tn = s_Tq[last_ok];
                                if (accepted) s_T = s_Tn;
                            }
                        }
                        have_lin = false;
                        __syncthreads();
                        break;
                    }
                    // thread 0: solve, trial pose
                    double dx[6] = {0, 0, 0, 0, 0, 0};
                    if (tid == 0) {
                        double H[36], b[6];
                        {
                            int k = 0;
                            for (int a = 0; a < 6; ++a)
                                for (int c = a; c < 6; ++c) {
                                    H[6 * a + c] = s_sys[k];
                                    H[6 * c + a] = s_sys[k];
                                    ++k;
                                }
                            for (int a = 0; a < 6; ++a) b[a] = s_sys[21 + a];
                        }
                        const bool ok = solve6_d(H, lambda, b, dx);
                        if (ok) {
                            PoseD E;
                            se3_exp_d(dx, E);
                            compose_d(E, s_T, s_Tn);
                        }
                        s_ctl[0] = ok ? 1.0 : 0.0;
                        double scale = 0;
                        if (ok)
                            for (int j = 0; j < 6; ++j) scale += dx[j] * (lambda * dx[j] + b[j]);
                        s_ctl[1] = scale + 1e-3;
                    }
                    __syncthreads();
                    const bool ok = s_ctl[0] != 0.0;
                    const double scale = s_ctl[1];
                    double temp_chi = 1.7976931348623157e308;
                    const bool full = ok && qmax == 0;   // (workgroup-uniform)
                    if (full) {
                        if (!linearise_at(true)) {
                            if (tid == 0 && grp == 0) num_valid[p] = -2;
                            return;
                        }
                        temp_chi = s_sum[27];
                        err_at_trial = true;
                    } else if (ok) {
                        double R[9], t[3];
                        for (int i = 0; i < 9; ++i) R[i] = s_Tn.R[i];
                        for (int i = 0; i < 3; ++i) t[i] = s_Tn.t[i];
                        double part = 0;
                        auto sweep = [&](auto stereo_tag) __attribute__((always_inline)) {
                            for_each_obs([&](int k, int, const ovs_pose_obs& o) __attribute__((always_inline)) {
                                if ((active >> k) & 1u) {
                                    const double c2 = pose_edge_impl<MODEL, decltype(stereo_tag)::value>(R, t, o, cam, bf, 0.0, nullptr);
                                    const double delta = robust ? huber : 0.0;
                                    double r = c2;
                                    if (delta > 0 && c2 > delta * delta) r = 2 * sqrt(c2) * delta - delta * delta;
                                    part += r;
                                }
                            });
                        };
                        if (has_stereo) sweep(std::true_type{});
                        else sweep(std::false_type{});
                        reduce(&part, 1);
                        POSE_EXCHANGE(1)
                        temp_chi = s_sum[0];
                        err_at_trial = true;
                    }
                    __syncthreads();
                    rho = (current_chi - temp_chi) / scale;
                    if (rho > 0 && isfinite(temp_chi)) {
                        const double tr = 2 * rho - 1;
                                double alpha = 1. - tr * tr * tr;   // (g2o: pow(2 rho - 1, 3); the library pow was ~150 instructions per lane for a cube)
                        alpha = fmin(alpha, 2.0 / 3.0);
                        lambda *= fmax(1.0 / 3.0, alpha);
                        ni = 2;
                        current_chi = temp_chi;
                        if (tid == 0) s_T = s_Tn;
                        have_lin = full;   // the system just evaluated at s_Tn is the one the next iteration needs
                    } else {
                        lambda *= ni;
                        ni *= 2;
                    }
                    ++qmax;
                    __syncthreads();
                } while (rho < 0 && qmax < 10);
                if (qmax == 10 || rho == 0) break;
            }
            // ---- re-classify every observation: outliers of the previous round at the estimate (upstream calls computeError() for
            // them), inliers at the state their errors were last computed at (the last trial state: the estimate itself unless the
            // round ended on a rejected step)
            {
                double R[9], t[3], Re[9], te[3];
                for (int i = 0; i < 9; ++i) R[i] = s_T.R[i];
                for (int i = 0; i < 3; ++i) t[i] = s_T.t[i];
                for (int i = 0; i < 9; ++i) Re[i] = err_at_trial ? s_Tn.R[i] : R[i];
                for (int i = 0; i < 3; ++i) te[i] = err_at_trial ? s_Tn.t[i] : t[i];
                int bad = 0;
                const uint32_t was_active = active;
                active = 0;
                for_each_obs([&](int k, int i, const ovs_pose_obs& o) __attribute__((always_inline)) {
                    const bool wa = (was_active >> k) & 1u;
                    const double c2 = pose_edge_impl<MODEL, true>(wa ? Re : R, wa ? te : t, o, cam, bf, 0.0, nullptr);
                    const bool out = ((MODEL == 0 && o.is_stereo) ? kChi3D : kChi2D) < c2;
                    outlier[i] = out ? 1 : 0;
                    if (out) ++bad;
                    else active |= 1u << k;
                });
                // block sum of bad (integers: exact)
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
                if (lane == 0) s_cnt[wv] = bad;
                __syncthreads();
                num_bad = 0;
                for (int w = 0; w < kPoseWaves; ++w) num_bad += s_cnt[w];
                __syncthreads();
                if (G > 1) {   // (counts below 2^53 are exact as doubles)
                    if (tid == 0) s_sum[0] = (double)num_bad;
                    __syncthreads();
                    POSE_EXCHANGE(1)
                    num_bad = (int)s_sum[0];
                    __syncthreads();
                }
            }
            if (n - num_bad < 5) break;   // upstream: if (num_init_obs - num_bad_obs < 5) break;
        }
    }
    if (grp == 0) {
        if (tid < 9) poses_out[12 * (size_t)p + tid] = s_T.R[tid];
        if (tid < 3) poses_out[12 * (size_t)p + 9 + tid] = s_T.t[tid];
        if (tid == 0) num_valid[p] = n >= 5 ? n - num_bad : 0;
    }
#undef POSE_EXCHANGE
}

}   // namespace ovs

using namespace ovs;

// OVS_POSE_BATCH_RETRIES: the retries 1 .. 9 of an iteration in one pass -- 1 always, 0 never, unset (-1): by work per thread (see
// pose_optimize_host). Read once per process, like the switches of ovs::tuning().
static int pose_batch_retries_env() {
    static const int v = [] {
        const char* e = std::getenv("OVS_POSE_BATCH_RETRIES");
        return e && e[0] ? (std::atoi(e) != 0 ? 1 : 0) : -1;
    }();
    return v;
}

// OVS_POSE_OBS_REGS=0: observations re-read from memory in every pass (the form of rounds 3-5; same bits, for A/B timing and the test that
// compares the two forms)
static bool tuning_pose_obs_regs() {
    static const bool v = [] {
        const char* e = std::getenv("OVS_POSE_OBS_REGS");
        return !(e && e[0] == '0');
    }();
    return v;
}

// ovs_pose_set_variant(OVS_POSE_VARIANT_RESET_EACH_ROUND, 0 | 1): process-wide, read at every launch
static std::atomic<int> g_pose_reset_each_round{0};

extern "C" {

static ovs_status pose_optimize_batch_dev(int model, const double* d_poses_in, const ovs_pose_obs* d_obs, const int32_t* d_obs_offsets, int32_t batch,
                                          const ovs_ba_cam& cam, double focal_x_baseline, int32_t setup_type, double* d_poses_out,
                                          uint8_t* d_outlier, int32_t* d_num_valid, void* stream, int threads_default = 256, int groups = 1,
                                          unsigned long long* d_gpart = nullptr, unsigned int epoch0 = 0, int batch_retries_auto = 0, bool obs_in_regs = false, int stereo_hint = -1) {
    if (!d_poses_in || !d_obs || !d_obs_offsets || !d_poses_out || !d_outlier || !d_num_valid || batch < 1) return OVS_ERR_INVALID;
    // workgroup size: the kernel is one latency-bound workgroup per frame; more waves hide the f64 latency of the per-observation work
    // but pay in barriers (measured per 2000-observation frame in DESIGN.md section 3.6)
    const int threads_env = tuning().pose_threads;
    const int T = threads_env == 256 || threads_env == 512 ? threads_env : threads_default;
    const size_t lds = sizeof(double) * 28 * (size_t)(T + 8);
#define OVS_POSE_LAUNCH(MODEL, TT, KR, BF, ST)                                                                                              \
    do {                                                                                                                              \
        static LdsAttrCache configured; /* per device: a second device needs the attribute too (116 KB of dynamic LDS at 512 threads) */  \
        OVS_HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void*>(k_pose_optimize<MODEL, TT, KR>), sizeof(double) * 28 * (TT + 8), configured)); \
        hipLaunchKernelGGL((k_pose_optimize<MODEL, TT, KR>), dim3(batch, groups > 1 ? 8 * groups : 1), dim3(TT), lds, (hipStream_t)stream, d_poses_in, d_obs, d_obs_offsets, \
                           cam, BF, ST, d_poses_out, d_outlier, d_num_valid, g_pose_reset_each_round.load(std::memory_order_relaxed), groups, d_gpart, \
                           epoch0, pose_batch_retries_env() < 0 ? batch_retries_auto : pose_batch_retries_env(), stereo_hint);      \
    } while (0)
    // obs_in_regs: the caller knows every frame of the launch has at most 2 * groups * 256 observations (KREG = 2, 256-thread workgroups only)
    const bool kreg = obs_in_regs && T == 256 && tuning_pose_obs_regs();
    if (model == 1) {
        if (T == 512) OVS_POSE_LAUNCH(1, 512, 0, 0.0, 0);
        else if (kreg) OVS_POSE_LAUNCH(1, 256, 2, 0.0, 0);
        else OVS_POSE_LAUNCH(1, 256, 0, 0.0, 0);
    } else {
        if (T == 512) OVS_POSE_LAUNCH(0, 512, 0, focal_x_baseline, (int)setup_type);
        else if (kreg) OVS_POSE_LAUNCH(0, 256, 2, focal_x_baseline, (int)setup_type);
        else OVS_POSE_LAUNCH(0, 256, 0, focal_x_baseline, (int)setup_type);
    }
#undef OVS_POSE_LAUNCH
    OVS_HIP_TRY(hipGetLastError());
    return OVS_OK;
}

ovs_status ovs_pose_set_variant(int32_t which, int32_t value) {
    if (which != OVS_POSE_VARIANT_RESET_EACH_ROUND || (value != 0 && value != 1)) return OVS_ERR_INVALID;
    g_pose_reset_each_round.store(value, std::memory_order_relaxed);
    return OVS_OK;
}

ovs_status ovs_pose_optimize_batch_dev(const double* d_poses_in, const ovs_pose_obs* d_obs, const int32_t* d_obs_offsets, int32_t batch,
                                       const ovs_ba_cam* cam, double focal_x_baseline, int32_t setup_type, double* d_poses_out,
                                       uint8_t* d_outlier, int32_t* d_num_valid, void* stream) {
    if (!cam) return OVS_ERR_INVALID;
    return pose_optimize_batch_dev(0, d_poses_in, d_obs, d_obs_offsets, batch, *cam, focal_x_baseline, setup_type, d_poses_out, d_outlier, d_num_valid,
                                   stream);
}

ovs_status ovs_pose_optimize_equirect_batch_dev(const double* d_poses_in, const ovs_pose_obs* d_obs, const int32_t* d_obs_offsets, int32_t batch,
                                                int32_t cols, int32_t rows, double* d_poses_out, uint8_t* d_outlier, int32_t* d_num_valid,
                                                void* stream) {
    if (cols < 1 || rows < 1) return OVS_ERR_INVALID;
    const ovs_ba_cam cam = {(double)cols, (double)rows, 0.0, 0.0};
    return pose_optimize_batch_dev(1, d_poses_in, d_obs, d_obs_offsets, batch, cam, 0.0, 0, d_poses_out, d_outlier, d_num_valid, stream);
}

static ovs_status pose_optimize_host(int model, int32_t device, const double* pose_cw_in, const ovs_pose_obs* obs, int32_t n_obs,
                                     const ovs_ba_cam* cam, double focal_x_baseline, int32_t setup_type, double* pose_cw_out,
                                     uint8_t* outlier_flags, int32_t* num_valid) {
    if (!pose_cw_in || !cam || !pose_cw_out || !num_valid || n_obs < 0 || (n_obs > 0 && (!obs || !outlier_flags))) return OVS_ERR_INVALID;
    if (n_obs > kPoseMaxObs) return OVS_ERR_CAPACITY;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    OVS_HIP_TRY(hipSetDevice(device));
    const size_t no = (size_t)std::max(n_obs, 1);
    // one device block and its pinned host mirror: inputs [pose 12 f64 | offsets 2 i32 | pad | observations] go up in ONE copy, outputs
    // [pose 12 f64 | num_valid | pad | outlier flags] come back in ONE copy, one wait (was three synchronous copies each way: ~60 us of a
    // 0.7 ms call)
    const size_t off_off = 96, off_sync = 112, off_obs = 128, in_bytes = off_obs + sizeof(ovs_pose_obs) * no;
    const size_t off_out = (in_bytes + 255) & ~(size_t)255, off_nv = off_out + 96, off_fl = off_out + 128, out_bytes = 128 + no;
    constexpr int kMaxGroups = 8;
    const size_t off_part = (off_out + out_bytes + 255) & ~(size_t)255, total = off_part + sizeof(unsigned long long) * 2 * kMaxGroups * 56;
    // per-thread, per-device staging that only grows: this runs once per tracked frame, an allocation per call would cost more than
    // the optimisation itself
    struct Scratch {
        unsigned char *p = nullptr, *h = nullptr;
        hipStream_t stream = nullptr;
        size_t cap = 0;
        int device = -1;
        unsigned int epoch0 = 0;   // the exchange words' flags of call c are epoch0 + 1 ..: + 4096 per call, so an earlier call's words never match
        void release() {
            if (p) (void)hipFree(p);
            if (h) (void)hipHostFree(h);
            if (stream) (void)hipStreamDestroy(stream);
            p = h = nullptr;
            stream = nullptr;
            cap = 0;
            device = -1;
        }
        ~Scratch() { release(); }
    };
    static thread_local Scratch scratch;
    if (scratch.device != device || scratch.cap < total) {
        scratch.release();
        const size_t want = std::max<size_t>(total, (size_t)1 << 20);
        OVS_HIP_TRY(hipMalloc(&scratch.p, want));
        OVS_HIP_TRY(hipMemset(scratch.p, 0, want));   // (exchange flags of a fresh block: zero, which no epoch equals)
        OVS_HIP_TRY(hipHostMalloc(&scratch.h, want, hipHostMallocDefault));
        OVS_HIP_TRY(hipStreamCreateWithFlags(&scratch.stream, hipStreamNonBlocking));
        scratch.cap = want;
        scratch.device = device;
    }
    unsigned char *d = scratch.p, *h = scratch.h;
    const int32_t offs[2] = {0, n_obs};
    std::memcpy(h, pose_cw_in, sizeof(double) * 12);
    std::memcpy(h + off_off, offs, sizeof(offs));
    std::memset(h + off_sync, 0, 16);   // (unused since round 6: the exchange has no arrival counter)
    if (n_obs) std::memcpy(h + off_obs, obs, sizeof(ovs_pose_obs) * (size_t)n_obs);
    // one latency-bound workgroup: 512 threads hide the f64 latency of the per-observation work when there is enough of it; measured,
    // 256 / 512 threads: perspective 2000 observations 0.509 / 0.512 ms, 1000: 0.438 / 0.422, 500: 0.286 / 0.293; equirectangular
    // 2000: 1.39 / 1.15, 1000: 0.759 / 0.786, 500: 0.427 / 0.588
    // Round 4: a frame with 1200 or more observations is spread over four workgroups of 256 threads on one XCD: the per-iteration pass over
    // the observations shrinks to a quarter, at the price of one grid-wide barrier per pass (~1.5 us: arrival counter + partial sums through
    // that XCD's L2). Means over 8 synthetic frames, 1 / 4 workgroups (profiles/r04w_pose_groups.txt; r04u_pose_groups.txt has 2 and 8
    // too): 300 observations 0.253 / 0.345 ms, 700: 0.280 / 0.327, 1000: 0.392 / 0.397, 1300: 0.408 / 0.367, 2000: 0.452 / 0.381,
    // 4000: 0.614 / 0.378. A frame's time also depends on how many rejected trials its converged rounds end on (+-20 % between frames of
    // one size), hence means, and hence a threshold a little above the crossover (tracked 1080p frames carry ~1300 matches: mean of four frame pairs 0.400 -> 0.376 ms). OVS_POSE_GROUPS=g forces g; the one-workgroup form is also the
    // fallback if a barrier is ever abandoned (num_valid == -2).
    const int groups_env = tuning().pose_groups;
    // round 5 (after the fused multiply-add accumulation made the per-observation work a third lighter; profiles/r05s_pose_groups.txt, means over 8
    // frames, 1 / 2 / 4 / 8 workgroups): 300 observations 0.235 / 0.259 / 0.267 / 0.285 ms, 700: 0.275 / 0.283 / 0.285 / 0.302, 1000: 0.320 / 0.322 /
    // 0.277 / 0.297, 1300: 0.335 / 0.323 / 0.298 / 0.288, 2000: 0.373 / 0.382 / 0.324 / 0.301, 4000: 0.541 / 0.466 / 0.378 / 0.325
    // round 6 (observations in registers, exchange without an arrival counter; profiles/r06w_pose_groups.txt, 1 / 2 / 4 / 8 workgroups): 300 observations
    // 0.222 / 0.234 / 0.238 / 0.251 ms, 700: 0.284 / 0.251 / 0.257 / 0.264, 1000: 0.332 / 0.284 / 0.249 / 0.258, 1300: 0.345 / 0.312 / 0.255 / 0.251,
    // 2000: 0.383 / 0.370 / 0.280 / 0.262, 4000: 0.553 / 0.461 / 0.363 / 0.277
    int groups = groups_env > 0 ? std::min(groups_env, kMaxGroups) : (n_obs >= 1600 ? 8 : (n_obs >= 850 ? 4 : (n_obs >= 500 ? 2 : 1)));
    int stereo_hint = 0;
    if (model == 0)
        for (int32_t i = 0; i < n_obs; ++i) stereo_hint |= obs[i].is_stereo != 0;
    // Round 6, zero copy: when the kernel reads every record exactly once (KREG form), it reads them -- and the pose -- straight from the pinned block
    // and writes pose, count and flags straight into it: no H2D copy before the launch, no D2H copy after it (two runtime copy commands and their
    // dependencies, ~20 us of a ~0.25 ms call). The exchange words stay in device memory. OVS_POSE_ZERO_COPY=0: the copies of rounds 3-5.
    static const bool zero_copy_env = [] {
        const char* e = std::getenv("OVS_POSE_ZERO_COPY");
        return !(e && e[0] == '0');
    }();
    bool first = true;
    for (;;) {
        // retries 1 .. 9 of an iteration in ONE pass (k_pose_optimize): nine trial poses per observation pay where a thread holds few
        // observations -- means over 8 frames, one pass / one by one (profiles/r04aj_pose_batched_retries.txt): four workgroups 1300
        // observations 0.322 / 0.347 ms, 2000: 0.337 / 0.370, 300: 0.271 / 0.322; one workgroup 300: 0.245 / 0.256 but 1300: 0.408 / 0.394,
        // 2000: 0.478 / 0.439 (a sequence that accepts its second or third trial has then evaluated seven poses for nothing). Same bits.
        const int batch_retries = (groups > 1 || n_obs <= 512) ? 1 : 0;
        const int threads = groups > 1 ? 256 : ((model == 1 ? n_obs >= 1500 : n_obs >= 768) ? 512 : 256);
        const bool in_regs = n_obs <= 2 * groups * 256 && threads == 256 && tuning_pose_obs_regs() && tuning().pose_threads != 512;
        const bool zero_copy = in_regs && zero_copy_env && first;
        unsigned char* const io = zero_copy ? h : d;   // where the kernel finds its inputs and leaves its outputs
        if (!zero_copy) OVS_HIP_TRY(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, scratch.stream));
        const ovs_status st = pose_optimize_batch_dev(model, reinterpret_cast<double*>(io), reinterpret_cast<ovs_pose_obs*>(io + off_obs),
                                                      reinterpret_cast<int32_t*>(io + off_off), 1, *cam, focal_x_baseline, setup_type,
                                                      reinterpret_cast<double*>(io + off_out), io + off_fl, reinterpret_cast<int32_t*>(io + off_nv),
                                                      scratch.stream, threads, groups, reinterpret_cast<unsigned long long*>(d + off_part),
                                                      (scratch.epoch0 += 4096u), batch_retries, in_regs, stereo_hint);
        if (st != OVS_OK) {
            (void)hipStreamSynchronize(scratch.stream);   // the upload may still be reading the pinned block the next call overwrites
            return st;
        }
        if (!zero_copy) OVS_HIP_TRY(hipMemcpyAsync(h + off_out, d + off_out, out_bytes, hipMemcpyDeviceToHost, scratch.stream));
        OVS_HIP_TRY(hipStreamSynchronize(scratch.stream));
        int32_t nv = 0;
        std::memcpy(&nv, h + off_nv, sizeof(nv));
        if (nv != -2 || groups == 1) break;
        groups = 1;   // a workgroup of the frame was not scheduled within 50 ms: run the frame in one workgroup (inputs through device memory)
        first = false;
    }
    std::memcpy(pose_cw_out, h + off_out, sizeof(double) * 12);
    std::memcpy(num_valid, h + off_nv, sizeof(int32_t));
    if (n_obs) std::memcpy(outlier_flags, h + off_fl, (size_t)n_obs);
    return OVS_OK;
}

ovs_status ovs_pose_optimize(int32_t device, const double* pose_cw_in, const ovs_pose_obs* obs, int32_t n_obs, const ovs_ba_cam* cam,
                             double focal_x_baseline, int32_t setup_type, double* pose_cw_out, uint8_t* outlier_flags, int32_t* num_valid) {
    return pose_optimize_host(0, device, pose_cw_in, obs, n_obs, cam, focal_x_baseline, setup_type, pose_cw_out, outlier_flags, num_valid);
}

ovs_status ovs_pose_optimize_equirect(int32_t device, const double* pose_cw_in, const ovs_pose_obs* obs, int32_t n_obs, int32_t cols, int32_t rows,
                                      double* pose_cw_out, uint8_t* outlier_flags, int32_t* num_valid) {
    if (cols < 1 || rows < 1) return OVS_ERR_INVALID;
    const ovs_ba_cam cam = {(double)cols, (double)rows, 0.0, 0.0};
    return pose_optimize_host(1, device, pose_cw_in, obs, n_obs, &cam, 0.0, 0, pose_cw_out, outlier_flags, num_valid);
}

}   // extern "C"
