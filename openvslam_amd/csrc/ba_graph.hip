// ba_graph.hip -- B1-B4 without atomics: a local-BA graph resident in HBM (edges indexed by landmark and by keyframe once per edge set),
// deterministic linearisation, landmark elimination (Schur complement) and back-substitution on the device
// (expected: src/openvslam/optimize/local_bundle_adjuster.cc; g2o BlockSolver_6_3::buildSystem / ::solve, OptimizationAlgorithmLevenberg).
//
// Why: k_ba_linearize issued 1.2 M scattered fp64 atomics per linearisation (0.022 of HBM, sums that change from
// run to run), and ovs_local_ba_optimize downloaded 16 MB of blocks per Levenberg-Marquardt trial to eliminate the landmarks on the host.
// Here every sum has a fixed order:
//   k_linearize      (round 5: ONE launch for the two independent halves below, which ran one after the other until then)
//     lin_landmark   one lane per landmark walks ITS edges in ascending index (mono first, then stereo -- the oracle's association):
//                    Hll | bl exactly as a sequential loop would add them, Hpl per edge, per-landmark chi2 partials;
//     lin_pose       one workgroup per keyframe over ITS edges: 21 + 6 pose-block terms per lane, fixed-shape tree reduction;
//   k_reduce_scalars chi2 (and the Levenberg-Marquardt start damping: max |H_jj|) by one workgroup, fixed tree; at the end of an LM trial
//                    also the sum of the landmarks' gain-ratio terms (the former k_sum_1024);
//   k_lm_prepare     (Hll + lambda I)^-1 per landmark and Y_e = W_e Hll^-1 per edge;
//   k_schur          (round 5: one launch)
//     schur_pair     one workgroup per pair of free keyframes (a, b >= a): S_ab = [a == b](Hpp_a + lambda I) - sum over the landmarks both
//                    observe of Y_ea W_eb^T; the common landmarks are found on the device (k_edge_table: keyframe x landmark -> edge);
//     schur_rhs      g_a = bp_a - sum over a's edges of Y_e bl_j, one workgroup per free keyframe;
//   ba_solve.hip     Cholesky of the reduced camera system (round 4; ovs_local_ba_set_solver(1): on the host as in rounds 1-3 and in
//                    BASELINE's north star -- then 0.7 MB come down and 5 KB go up per trial);
//   k_trial_update   (round 5: one launch)
//     pose_update    T <- exp(dx) T per free keyframe (g2o SE3Quat::exp as in ba_host_math.h se3_oplus), the 7-double records the
//                    linearisation reads, the keyframes' part of the gain ratio's denominator -- one workgroup;
//     backsub_landmark  dxl_j = Hll^-1 (bl_j - sum W_e^T dxp), the trial points X + dxl and the landmark's term of g2o's gain-ratio scale
//                    (k_backsub: the same on increments uploaded by the host solver).
// Round 6 (DESIGN.md section 3.6 has the measurements): one lane per EDGE on both sides of the linearisation and in the back-substitution; both
// halves of a linearisation in ONE launch again (k_linearize2: the keyframe side fell from 218 to 78 VGPRs with one entry per thread), the
// keyframes' blocks finished by extra workgroups of k_reduce_scalars; every 144-byte record (Hpl, Y) leaves through LDS as whole lines -- a lane
// storing its own record issued nine partial-line requests, and those, not bytes, were what the kernels queued behind; the pairs' common
// landmarks listed once per graph (k_pair_lists) and k_schur_l gathering its records cooperatively, whole rows of the pair table per XCD; the
// chi-square gates (k_edge_gate) and the graph build's landmark-order pass (k_edges_by_slot, k_dup_check) on the device.
// Round 5's three mergers are side-by-side only (a workgroup runs one of the two bodies, chosen by its index): the arithmetic of every block,
// landmark and keyframe is what it was, the results are the same bits, and a trial is 332 instead of 386 us of kernels (profiles/r05ai_*).
// Per-edge arithmetic is the expression sequence of k_ba_linearize / the oracle (every product individually rounded), so Hpl, Hll and bl
// are bit-identical to the CPU oracle; Hpp, bp and chi2 are tree sums (1e-15 relative), identical from run to run.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "ovs_common.h"

#define OVS_LAUNCH_TRY(name)                                  \
    do {                                                      \
        hipError_t _e = hipGetLastError();                    \
        if (_e != hipSuccess) {                               \
            ovs::set_last_error("launch of " name, _e);       \
            return OVS_ERR_HIP;                               \
        }                                                     \
    } while (0)

namespace ovs {

// ba_solve.hip: the padded layout of the reduced camera system
int dense_solve_pad(int n);
size_t dense_solve_doubles(int n);

struct GEdge {   // mono and stereo observations in one record; stereo iff index >= n_mono
    int32_t pose, pt;
    double ox, oy, oxr, w;
};

// Jacobians are carried as [3][6] arrays; the third row is zero for a mono edge (never read: dot3 stops at two rows)
__device__ __forceinline__ double dot3(const double (&A)[3][6], int a, const double (&B)[3][6], int b, bool stereo) {
    double s = A[0][a] * B[0][b];
    s = s + A[1][a] * B[1][b];
    if (stereo) s = s + A[2][a] * B[2][b];
    return s;
}

// residual, Jacobians, Huber weight of one perspective edge -- the operation order of k_ba_linearize<2|3> (model 0) and of the CPU oracle
__device__ __forceinline__ void edge_lin(const double* __restrict__ P, const double* __restrict__ X, const GEdge& ed, bool stereo,
                                         const ovs_ba_cam& cam, double bf, double huber, double (&Jl)[3][6], double (&Jp)[3][6], double (&r)[3],
                                         double& W, double& c2, double& rho0) {
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
    double er[3] = {0.0, 0.0, 0.0};
    const double u = cam.fx * x * invz + cam.cx;
    er[0] = ed.ox - u;
    er[1] = ed.oy - (cam.fy * y * invz + cam.cy);
    double ss = er[0] * er[0] + er[1] * er[1];
    if (stereo) {
        er[2] = ed.oxr - (u - bf * invz);
        ss = ss + er[2] * er[2];
    }
    const double w = ed.w;
    c2 = w * ss;
    rho0 = c2;
    double rho1 = 1.0;
    const double dsqr = huber * huber;
    if (huber > 0 && c2 > dsqr) {
        const double sq = sqrt(c2);
        rho0 = 2 * sq * huber - dsqr;
        rho1 = huber / sq;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) Jl[0][c] = Jl[1][c] = Jl[2][c] = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Jl[0][c] = -invz * (cam.fx * R[0][c] - cam.fx * x * invz * R[2][c]);
        Jl[1][c] = -invz * (cam.fy * R[1][c] - cam.fy * y * invz * R[2][c]);
        Jl[2][c] = stereo ? Jl[0][c] - bf * R[2][c] * invz2 : 0.0;
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
    if (stereo) {
        Jp[2][0] = Jp[0][0] - bf * y * invz2;
        Jp[2][1] = Jp[0][1] + bf * x * invz2;
        Jp[2][2] = Jp[0][2];
        Jp[2][3] = Jp[0][3];
        Jp[2][4] = 0;
        Jp[2][5] = Jp[0][5] - bf * invz2;
    } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) Jp[2][c] = 0.0;
    }
    W = rho1 * w;
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = -W * er[k];
}

// the equirectangular edge (model 1; expected: src/openvslam/optimize/g2o/se3/equirectangular_reproj_edge.{h,cc}): the operation order of
// k_ba_linearize's equirectangular model and of the CPU checker; cam = {cols, rows, -, -}; mono edges only (rows 2 stay zero)
__device__ __forceinline__ void edge_lin_equirect(const double* __restrict__ P, const double* __restrict__ X, const GEdge& ed, const ovs_ba_cam& cam,
                                                  double huber, double (&Jl)[3][6], double (&Jp)[3][6], double (&r)[3], double& W, double& c2,
                                                  double& rho0) {
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
    const double kPi = 3.14159265358979323846;
    const double cols = cam.fx, rows = cam.fy;
    const double L = sqrt((x * x + y * y) + z * z);
    const double rxz = x * x + z * z;
    const double theta = ovs_det_atan2(x, z);
    const double phi = -ovs_det_asin(y / L);
    const double e0 = ed.ox - cols * (0.5 + theta / (2.0 * kPi));
    const double e1 = ed.oy - rows * (0.5 - phi / kPi);
    const double w = ed.w;
    c2 = w * (e0 * e0 + e1 * e1);
    rho0 = c2;
    double rho1 = 1.0;
    const double dsqr = huber * huber;
    if (huber > 0 && c2 > dsqr) {
        const double sq = sqrt(c2);
        rho0 = 2 * sq * huber - dsqr;
        rho1 = huber / sq;
    }
    const double a0 = -(cols / (2.0 * kPi)) * (1.0 / rxz);
    const double a1 = -(rows / kPi) * (1.0 / (L * sqrt(rxz)));
    auto col = [&](double dx, double dy, double dz, double& j0, double& j1) {
        const double dL = (1.0 / L) * ((x * dx + y * dy) + z * dz);
        j0 = a0 * (z * dx - x * dz);
        j1 = a1 * (L * dy - y * dL);
    };
#pragma unroll
    for (int c = 0; c < 6; ++c) Jl[0][c] = Jl[1][c] = Jl[2][c] = Jp[2][c] = 0.0;
    col(0.0, -z, y, Jp[0][0], Jp[1][0]);
    col(z, 0.0, -x, Jp[0][1], Jp[1][1]);
    col(-y, x, 0.0, Jp[0][2], Jp[1][2]);
    col(1.0, 0.0, 0.0, Jp[0][3], Jp[1][3]);
    col(0.0, 1.0, 0.0, Jp[0][4], Jp[1][4]);
    col(0.0, 0.0, 1.0, Jp[0][5], Jp[1][5]);
#pragma unroll
    for (int c = 0; c < 3; ++c) col(R[0][c], R[1][c], R[2][c], Jl[0][c], Jl[1][c]);
    W = rho1 * w;
    r[0] = -W * e0;
    r[1] = -W * e1;
    r[2] = 0.0;
}

struct GraphDev {   // device views shared by the kernels
    const GEdge* edges;
    int n_mono, n_edge, n_pose, n_pt;
    const int32_t* lm_start;     // [n_pt + 1] into lm_edges; per landmark: mono edges ascending, then stereo edges ascending
    const int32_t* lm_edges;
    const int32_t* lm_nmono;     // [n_pt] how many of the landmark's edges are mono
    const int32_t* pose_start;   // [n_pose + 1] into pose_edges
    const int32_t* pose_edges;
    const int32_t* pose_pt;      // [n_edge] landmark of every entry of pose_edges
    const uint8_t* fixed;        // [n_pose]
    const uint8_t* active;       // [n_edge] 0 = the edge is at g2o level 1 (an outlier of round 1): it contributes nothing
    // work partition of k_linearize (round 6): one lane per EDGE on both sides
    const GEdge* ledges;         // [n_edge] the edge records in lm_edges' order (built on the device when the graph is created): a landmark
                                 // workgroup streams its records instead of gathering 48 of every 128 bytes it fetches
    const int32_t* lm_of_slot;   // [n_edge] landmark of every entry of lm_edges
    const int32_t* lm_wg_first;  // [n_lm_wg + 1] first landmark of every landmark workgroup: whole landmarks, at most 256 edges and 256 landmarks
                                 // (a landmark with more than 256 edges is a workgroup of its own)
    const int32_t* chunk_kf;     // [n_chunks] keyframe of every chunk of kPoseChunk entries of pose_edges
    const int32_t* chunk_start;  // [n_pose + 1] first chunk of every keyframe
    double* pose_part;           // [n_chunks x 27] the chunks' sums of the 21 + 6 pose-block terms
    int n_lm_wg, n_chunks;
    ovs_ba_cam cam;        // model 1: {cols, rows, -, -}
    double bf;
    int model;             // 0 perspective (mono / stereo edges), 1 equirectangular (mono edges)
};

// ---- linearisation ----------------------------------------------------------------------------------------------------------
// Round 6: one lane per EDGE on both sides (rounds 2-5: one lane per landmark walking its edges, one workgroup per keyframe walking 256 of
// its edges at a time -- chains of dependent gathers, 1.6 k waves for a million edges, 73 % of the wave cycles waiting). The SUMS keep their
// order: a landmark's contributions are added in ascending edge index, mono before stereo, exactly as the sequential loop did (Hll, bl, chi2
// partials: the same bits as rounds 2-5 and as the oracle); a keyframe's 27 terms are a fixed-shape tree per chunk of 512 edges, the chunks
// added in ascending order by k_reduce_scalars (deterministic; not the tree of rounds 2-5: last-bit differences in Hpp / bp).
constexpr int kPoseChunk = 512;     // entries of pose_edges per keyframe workgroup (two per thread)
constexpr int kLmSlots = 256;       // edges per landmark workgroup

// sum of x over the wave in lane 63: row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, then row_bcast 15 / 31 -- data-parallel moves in the
// vector ALU (two per double and step) instead of ds_bpermute round trips through the LDS crossbar (round 2-5's xor-shuffle tree: 324 of
// them per wave for the keyframe side's 27 sums). A fixed shape: the same bits from run to run.
__device__ __forceinline__ double wave_sum_lane63(double x) {
#define OVS_DPP_STEP(ctrl, row_mask, bound)                                                                                  \
    {                                                                                                                        \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, row_mask, 0xf, bound);                        \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, row_mask, 0xf, bound);                        \
        x = x + __hiloint2double(hi, lo);                                                                                    \
    }
    OVS_DPP_STEP(0x111, 0xf, true)    // row_shr:1
    OVS_DPP_STEP(0x112, 0xf, true)    // row_shr:2
    OVS_DPP_STEP(0x114, 0xf, true)    // row_shr:4
    OVS_DPP_STEP(0x118, 0xf, true)    // row_shr:8
    OVS_DPP_STEP(0x142, 0xa, false)   // row_bcast:15 into rows 1, 3
    OVS_DPP_STEP(0x143, 0xc, false)   // row_bcast:31 into rows 2, 3
#undef OVS_DPP_STEP
    return x;
}

// fixed-shape reduction of NV per-thread values over a 256-thread workgroup: the lanes of a wave (above), then the four waves in order
template <int NV>
__device__ __forceinline__ void block_sum_256(double (&v)[NV], double (*s_part)[NV]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double x = wave_sum_lane63(v[i]);
        if (lane == 63) s_part[wv][i] = x;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ((s_part[0][i] + s_part[1][i]) + s_part[2][i]) + s_part[3][i];
    __syncthreads();
}

// chunk c of keyframe k by one 256-thread workgroup: thread t takes entries t and t + 256 of the chunk, one after the other
template <int kModel, bool kStage>
__device__ __forceinline__ void lin_pose_chunk(const GraphDev& g, const int chunk, const double* __restrict__ poses, const double* __restrict__ points,
                                               double huber_mono, double huber_stereo, double* __restrict__ Hpl, double (*s_part)[27],
                                               double2* __restrict__ s_h) {
    const int k = g.chunk_kf[chunk];
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    const int e1 = g.pose_start[k + 1];
    const int i0 = g.pose_start[k] + (chunk - g.chunk_start[k]) * kPoseChunk + (int)threadIdx.x;
    auto zero_hpl = [&](int e) {
        double2* const h = reinterpret_cast<double2*>(Hpl + 18 * (size_t)e);
#pragma unroll
        for (int a = 0; a < 9; ++a) h[a] = double2{0.0, 0.0};
    };
    if (g.fixed[k]) {   // (workgroup-uniform) a fixed keyframe's block and its edges' Hpl are zero
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (i0 + 256 * u < e1) zero_hpl(g.pose_edges[i0 + 256 * u]);
    } else {
        const double* P = poses + 7 * (size_t)k;
        // both entries' gathers (index -> record -> landmark) are issued before either edge is worked on, one after the other
        int ee[2];
        GEdge edd[2];
        double XX[2][3];
        int pt[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool in = i0 + 256 * u < e1;
            ee[u] = in ? g.pose_edges[i0 + 256 * u] : -1;
            pt[u] = in ? g.pose_pt[i0 + 256 * u] : 0;   // the landmark without going through the edge record: one dependent load less
        }
        // (the records' 40-byte gathers were staged through LDS the same way as the stores below -- five coalesced loads per 64 consecutive
        // records -- and that changed nothing: 0.110 against 0.108 ms at a million edges; the loads are not what the kernel waits for)
#pragma unroll
        for (int u = 0; u < 2; ++u) edd[u] = g.edges[max(ee[u], 0)];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const double* x = points + 3 * (size_t)pt[u];
            XX[u][0] = x[0];
            XX[u][1] = x[1];
            XX[u][2] = x[2];
        }
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
            const int e = u ? ee[1] : ee[0];
            if (!kStage) {
                if (e < 0) break;
                if (!g.active[e]) {   // exact zeros for an edge at g2o level 1
                    zero_hpl(e);
                    continue;
                }
            }
            const bool live = kStage ? (e >= 0 && g.active[e]) : true;
            double2 hv[9];
            if (kStage) {
#pragma unroll
                for (int a = 0; a < 9; ++a) hv[a] = double2{0.0, 0.0};
            }
            if (live) {
                const GEdge ed = u ? edd[1] : edd[0];
                const double X[3] = {u ? XX[1][0] : XX[0][0], u ? XX[1][1] : XX[0][1], u ? XX[1][2] : XX[0][2]};
                const bool stereo = e >= g.n_mono;
                double Jl[3][6], Jp[3][6], r[3], W, c2, rho0;
                if (kModel == 1) edge_lin_equirect(P, X, ed, g.cam, huber_mono, Jl, Jp, r, W, c2, rho0);
                else edge_lin(P, X, ed, stereo, g.cam, g.bf, stereo ? huber_stereo : huber_mono, Jl, Jp, r, W, c2, rho0);
                int t = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
#pragma unroll
                    for (int b = a; b < 6; ++b) acc[t++] += W * dot3(Jp, a, Jp, b, stereo);
                    double gq = Jp[0][a] * r[0];
                    gq = gq + Jp[1][a] * r[1];
                    if (stereo) gq = gq + Jp[2][a] * r[2];
                    acc[t++] += gq;
                }
                // W_e = W Jp^T Jl: this side walks the edges in ascending index, so a wave's 64 records are 9 KB of consecutive bytes
                double2* const h = reinterpret_cast<double2*>(Hpl + 18 * (size_t)e);
#pragma unroll
                for (int a = 0; a < 6; a += 2) {   // rows a, a + 1: entries 3 a .. 3 a + 5
                    const double h0 = W * dot3(Jp, a, Jl, 0, stereo), h1 = W * dot3(Jp, a, Jl, 1, stereo), h2 = W * dot3(Jp, a, Jl, 2, stereo);
                    const double h3 = W * dot3(Jp, a + 1, Jl, 0, stereo), h4 = W * dot3(Jp, a + 1, Jl, 1, stereo), h5 = W * dot3(Jp, a + 1, Jl, 2, stereo);
                    if (kStage) {
                        hv[3 * (a >> 1)] = double2{h0, h1};
                        hv[3 * (a >> 1) + 1] = double2{h2, h3};
                        hv[3 * (a >> 1) + 2] = double2{h4, h5};
                    } else {
                        h[3 * (a >> 1)] = double2{h0, h1};
                        h[3 * (a >> 1) + 1] = double2{h2, h3};
                        h[3 * (a >> 1) + 2] = double2{h4, h5};
                    }
                }
            }
            if (kStage) {
                // Round 6, late: a lane storing its own 144-byte record writes nine 16-byte pieces, each store instruction touching 64 different
                // 128-byte lines (9 M partial-line write requests per million edges). When the wave's 64 edges are consecutive (the usual case:
                // edge lists arrive keyframe by keyframe) the records pass through LDS and every store instruction writes 1 KB of consecutive bytes.
                const int lane = (int)threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
                const int e_first = __shfl(e, 0);
                const bool contig = __ballot(e >= 0 && e == e_first + lane) == ~0ull;   // (wave-uniform)
                if (contig) {
                    double2* const sh = s_h + (size_t)wv * (64 * 9);
#pragma unroll
                    for (int a = 0; a < 9; ++a) sh[lane * 9 + a] = hv[a];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    double2* const h = reinterpret_cast<double2*>(Hpl + 18 * (size_t)e_first);
#pragma unroll
                    for (int a = 0; a < 9; ++a) h[a * 64 + lane] = sh[a * 64 + lane];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();   // (the next entry's records overwrite the buffer)
                } else if (e >= 0) {
                    double2* const h = reinterpret_cast<double2*>(Hpl + 18 * (size_t)e);
#pragma unroll
                    for (int a = 0; a < 9; ++a) h[a] = hv[a];
                }
            }
        }
    }
    block_sum_256<27>(acc, s_part);
    if (threadIdx.x < 27) {
        double v = acc[0];
#pragma unroll
        for (int i = 1; i < 27; ++i) v = (int)threadIdx.x == i ? acc[i] : v;
        g.pose_part[27 * (size_t)chunk + threadIdx.x] = v;
    }
}

// Round 6, late: HALF a chunk (256 entries of a keyframe's edge list) by a 256-thread workgroup, ONE entry per thread -- the keyframe side of
// k_linearize2. Two entries per thread (lin_pose_chunk) keep 27 running sums alive across both edges' Jacobians: 218 VGPRs, a budget the landmark
// side would have to share in a merged launch (round 5's merged k_linearize paid exactly that). With one entry a term goes from the Jacobians
// straight into its wave sum: 78 VGPRs. By itself that bought nothing (1 M edges: 0.119 against 0.122 ms with six instead of two waves per SIMD --
// the kernel is bound by its 144 bytes of Hpl per edge, not by latency); what it allows is the merged launch. A half chunk's sum is its four
// waves' sums in wave order; k_reduce_scalars adds the halves in ascending order (a fixed shape; it differs from the two-entry form's in the
// last bits of Hpp / bp, like every change of the tree so far; Hpl is per edge and keeps its bits).
template <int kModel>
__device__ __forceinline__ void lin_pose_half(const GraphDev& g, const int half_chunk, const double* __restrict__ poses, const double* __restrict__ points,
                                              double huber_mono, double huber_stereo, double* __restrict__ Hpl, double (*s_part)[27],   // [4][27]
                                              double2* __restrict__ s_h) {   // [4 waves][64 x 9]: the records on their way to coalesced stores (lin_pose_chunk)
    const int chunk = half_chunk >> 1;
    const int k = g.chunk_kf[chunk];
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
    const int e1 = g.pose_start[k + 1];
    const int i = g.pose_start[k] + (chunk - g.chunk_start[k]) * kPoseChunk + (half_chunk & 1) * 256 + (int)threadIdx.x;
    double Jl[3][6], Jp[3][6], r[3] = {0.0, 0.0, 0.0}, W = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) Jl[a][c] = Jp[a][c] = 0.0;
    bool stereo = false;
    const int e = i < e1 ? g.pose_edges[i] : -1;
    double2 hv[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) hv[a] = double2{0.0, 0.0};   // (a fixed keyframe's edges and the edges at g2o level 1: exact zeros)
    if (e >= 0 && !g.fixed[k]) {   // (fixed: workgroup-uniform)
        const int pt = g.pose_pt[i];
        const GEdge ed = g.edges[e];
        const double* x = points + 3 * (size_t)pt;
        const double X[3] = {x[0], x[1], x[2]};
        if (g.active[e]) {
            stereo = e >= g.n_mono;
            double c2, rho0;
            if (kModel == 1) edge_lin_equirect(poses + 7 * (size_t)k, X, ed, g.cam, huber_mono, Jl, Jp, r, W, c2, rho0);
            else edge_lin(poses + 7 * (size_t)k, X, ed, stereo, g.cam, g.bf, stereo ? huber_stereo : huber_mono, Jl, Jp, r, W, c2, rho0);
#pragma unroll
            for (int a = 0; a < 6; a += 2) {   // rows a, a + 1: entries 3 a .. 3 a + 5
                const double h0 = W * dot3(Jp, a, Jl, 0, stereo), h1 = W * dot3(Jp, a, Jl, 1, stereo), h2 = W * dot3(Jp, a, Jl, 2, stereo);
                const double h3 = W * dot3(Jp, a + 1, Jl, 0, stereo), h4 = W * dot3(Jp, a + 1, Jl, 1, stereo), h5 = W * dot3(Jp, a + 1, Jl, 2, stereo);
                hv[3 * (a >> 1)] = double2{h0, h1};
                hv[3 * (a >> 1) + 1] = double2{h2, h3};
                hv[3 * (a >> 1) + 2] = double2{h4, h5};
            }
        }
    }
    {   // the records leave through LDS when the wave's 64 edges are consecutive (see lin_pose_chunk)
        const int e_first = __shfl(e, 0);
        const bool contig = __ballot(e >= 0 && e == e_first + lane) == ~0ull;   // (wave-uniform)
        if (contig) {
            double2* const sh = s_h + (size_t)__builtin_amdgcn_readfirstlane(wv) * (64 * 9);
#pragma unroll
            for (int a = 0; a < 9; ++a) sh[lane * 9 + a] = hv[a];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double2* const h = reinterpret_cast<double2*>(Hpl + 18 * (size_t)e_first);
#pragma unroll
            for (int a = 0; a < 9; ++a) h[a * 64 + lane] = sh[a * 64 + lane];
        } else if (e >= 0) {
            double2* const h = reinterpret_cast<double2*>(Hpl + 18 * (size_t)e);
#pragma unroll
            for (int a = 0; a < 9; ++a) h[a] = hv[a];
        }
    }
    int t = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = a; b < 6; ++b) {
            const double x = wave_sum_lane63(W * dot3(Jp, a, Jp, b, stereo));   // (a lane without a live edge: W = 0 and Jp = 0)
            if (lane == 63) s_part[wv][t] = x;
            ++t;
        }
        double gq = Jp[0][a] * r[0];
        gq = gq + Jp[1][a] * r[1];
        if (stereo) gq = gq + Jp[2][a] * r[2];
        const double x = wave_sum_lane63(gq);
        if (lane == 63) s_part[wv][t] = x;
        ++t;
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        const int q = (int)threadIdx.x;
        g.pose_part[27 * (size_t)half_chunk + q] = ((s_part[0][q] + s_part[1][q]) + s_part[2][q]) + s_part[3][q];
    }
}

// landmarks [j0, j1) by one 256-thread workgroup: one lane per edge computes the edge's terms, then one lane per (landmark, term) adds a
// landmark's terms in its edges' order. s_c: [14][kLmSlots] terms of the slots (9 Hll, 3 bl, chi2, rho), s_d: [5][256] the sums' diagonal, chi2 and robustified chi2 per landmark.
template <int kModel>
__device__ __forceinline__ void lin_landmark_wg(const GraphDev& g, const int wg, const double* __restrict__ poses, const double* __restrict__ points,
                                                double huber_mono, double huber_stereo, double* __restrict__ Hll, double* __restrict__ bl,
                                                double* __restrict__ lm_chi, double (*s_c)[kLmSlots + 1], double (*s_d)[256]) {   // s_d: [5][256]
    const int tid = (int)threadIdx.x;
    const int j0 = g.lm_wg_first[wg], j1 = g.lm_wg_first[wg + 1], n_lm = j1 - j0;
    const int s0 = g.lm_start[j0], s1 = g.lm_start[j1];
    const bool big = s1 - s0 > kLmSlots;   // (workgroup-uniform) then n_lm == 1: the landmark's edges pass in pieces, threads 0 .. 13 carry its sums
    double run_m = 0.0, run_s = 0.0;

    for (int base = s0; base == s0 || base < s1; base += kLmSlots) {
        const int s = base + tid;
        if (s < s1) {
            const int e = g.lm_edges[s], j = g.lm_of_slot[s];
            if (!g.active[e]) {   // exact zeros: the sums of the remaining edges keep their order and value
#pragma unroll
                for (int q = 0; q < 14; ++q) s_c[q][tid] = 0.0;
            } else {
                const bool stereo = e >= g.n_mono;
                const GEdge ed = g.ledges[s];
                double Jl[3][6], Jp[3][6], r[3], W, c2, rho0;
                const double* X = points + 3 * (size_t)j;
                if (kModel == 1) edge_lin_equirect(poses + 7 * (size_t)ed.pose, X, ed, g.cam, huber_mono, Jl, Jp, r, W, c2, rho0);
                else edge_lin(poses + 7 * (size_t)ed.pose, X, ed, stereo, g.cam, g.bf, stereo ? huber_stereo : huber_mono, Jl, Jp, r, W, c2, rho0);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int b = 0; b < 3; ++b) s_c[3 * a + b][tid] = W * dot3(Jl, a, Jl, b, stereo);
                    double t = Jl[0][a] * r[0];
                    t = t + Jl[1][a] * r[1];
                    if (stereo) t = t + Jl[2][a] * r[2];
                    s_c[9 + a][tid] = t;
                }
                s_c[12][tid] = c2;
                s_c[13][tid] = rho0;
            }
        }
        __syncthreads();
        if (!big) {
            for (int u = tid; u < 14 * n_lm; u += 256) {
                const int l = (int)(((unsigned)u * 0x4925u) >> 18), q = u - 14 * l;   // u / 14 for u < 14 * 256 (0x4925 = ceil(2^18 / 14))
                const int j = j0 + l;
                const int a = g.lm_start[j] - base, nm = g.lm_nmono[j], b = g.lm_start[j + 1] - base;
                double hm = 0.0, hs = 0.0;   // mono and stereo sums kept apart: the oracle adds (mono total) + (stereo total)
                for (int i = a; i < a + nm; ++i) hm += s_c[q][i];
                for (int i = a + nm; i < b; ++i) hs += s_c[q][i];
                const double v = hm + hs;
                if (q < 9) {
                    Hll[9 * (size_t)j + q] = v;
                    if (q == 0 || q == 4 || q == 8) s_d[q >> 2][l] = v;
                } else if (q < 12) {
                    bl[3 * (size_t)j + (q - 9)] = v;
                } else {
                    s_d[q - 9][l] = v;   // rows 3, 4: the landmark's chi2 and robustified chi2
                }
            }
        } else if (tid < 14) {
            const int a = g.lm_start[j0], nm = g.lm_nmono[j0];
            const int m_end = min(a + nm, base + kLmSlots), s_end = min(s1, base + kLmSlots);
            for (int i = max(a, base); i < m_end; ++i) run_m += s_c[tid][i - base];
            for (int i = max(a + nm, base); i < s_end; ++i) run_s += s_c[tid][i - base];
        }
        __syncthreads();
    }
    if (big && tid < 14) {
        const double v = run_m + run_s;
        if (tid < 9) {
            Hll[9 * (size_t)j0 + tid] = v;
            if (tid == 0 || tid == 4 || tid == 8) s_d[tid >> 2][0] = v;
        } else if (tid < 12) {
            bl[3 * (size_t)j0 + (tid - 9)] = v;
        } else {
            s_d[tid - 9][0] = v;
        }
    }
    __syncthreads();
    // the workgroup's share of the scalars k_reduce_scalars finishes: chi2, robustified chi2 (sums over its landmarks, fixed shape) and the
    // largest |diagonal entry| of its landmarks' blocks (0 for a landmark without edges) -- three numbers per workgroup instead of three per landmark
    double a = 0.0, b = 0.0, m = 0.0;
    if (tid < n_lm) {
        const int j = j0 + tid;
        a = s_d[3][tid];
        b = s_d[4][tid];
        m = g.lm_start[j + 1] > g.lm_start[j] ? fmax(fmax(fabs(s_d[0][tid]), fabs(s_d[1][tid])), fabs(s_d[2][tid])) : 0.0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
        m = fmax(m, __shfl_xor(m, off));
    }
    __syncthreads();   // s_d's readers are done
    if ((tid & 63) == 0) {
        s_d[0][tid >> 6] = a;
        s_d[1][tid >> 6] = b;
        s_d[2][tid >> 6] = m;
    }
    __syncthreads();
    if (tid == 0) {
        lm_chi[3 * (size_t)wg] = ((s_d[0][0] + s_d[0][1]) + s_d[0][2]) + s_d[0][3];
        lm_chi[3 * (size_t)wg + 1] = ((s_d[1][0] + s_d[1][1]) + s_d[1][2]) + s_d[1][3];
        lm_chi[3 * (size_t)wg + 2] = fmax(fmax(s_d[2][0], s_d[2][1]), fmax(s_d[2][2], s_d[2][3]));
    }
}

// the edge records in lm_edges' order and the landmark of every slot, once per graph (round 6: the slots' landmarks were written by the host's
// landmark-order pass; together with the duplicate check below that pass was 0.17 of the graph build's 0.47 ms at config 5)
__global__ __launch_bounds__(256) void k_edges_by_slot(const GEdge* __restrict__ edges, const int32_t* __restrict__ lm_edges, int n_edge,
                                                      GEdge* __restrict__ ledges, int32_t* __restrict__ lm_of_slot,
                                                      unsigned long long* __restrict__ dup_word) {
    const int s = (int)(blockIdx.x * 256 + threadIdx.x);
    if (s == 0) *dup_word = ~0ull;   // (k_dup_check is the next launch on the stream)
    if (s < n_edge) {
        const GEdge e = edges[lm_edges[s]];
        ledges[s] = e;
        lm_of_slot[s] = e.pt;
    }
}
// A keyframe observes a landmark at most once: slot s reports {landmark : 32 | keyframe : 32} when an earlier slot of its landmark has the same
// keyframe; the smallest report wins, i.e. the first such landmark and its lowest such keyframe, whatever the order the lanes run in.
__global__ __launch_bounds__(256) void k_dup_check(const GEdge* __restrict__ ledges, const int32_t* __restrict__ lm_start, int n_edge,
                                                  unsigned long long* __restrict__ dup_word) {
    const int s = (int)(blockIdx.x * 256 + threadIdx.x);
    if (s >= n_edge) return;
    const int j = ledges[s].pt, k = ledges[s].pose;
    bool dup = false;
    for (int i = lm_start[j]; i < s; ++i) dup |= ledges[i].pose == k;
    if (dup) atomicMin(dup_word, ((unsigned long long)(uint32_t)j << 32) | (uint32_t)k);
}

// The two halves of a linearisation are independent (different outputs, the same inputs). Round 5 ran them side by side in ONE launch; as
// two launches each gets its own register budget: the keyframe side needs ~200 VGPRs (27 running sums beside both Jacobians: two waves per
// SIMD), the landmark side fewer than 128 (four waves per SIMD) -- merged, every workgroup paid the larger figure.
template <int kModel, bool kStage>
__global__ __launch_bounds__(256) void k_lin_pose(GraphDev g, const double* __restrict__ poses, const double* __restrict__ points, double huber_mono,
                                                 double huber_stereo, double* __restrict__ Hpl) {
    __shared__ double s_part[4][27];
    __shared__ double2 s_h[kStage ? 4 * 64 * 9 : 1];   // the four waves' 64 records of 144 bytes on their way to coalesced stores
    lin_pose_chunk<kModel, kStage>(g, (int)blockIdx.x, poses, points, huber_mono, huber_stereo, Hpl, s_part, s_h);
}
template <int kModel>
__global__ __launch_bounds__(256) void k_lin_landmark(GraphDev g, const double* __restrict__ poses, const double* __restrict__ points, double huber_mono,
                                                     double huber_stereo, double* __restrict__ Hpp, double* __restrict__ bp, double* __restrict__ Hll,
                                                     double* __restrict__ bl, double* __restrict__ lm_chi) {
    __shared__ double s_c[14][kLmSlots + 1];   // + 1: the fourteen terms of a slot in fourteen different banks
    __shared__ double s_d[5][256];
    // first, the keyframes' blocks (k_lin_pose, the launch before this one, has left a keyframe's 27 terms as one sum per chunk of its edges):
    // the chunks are added in ascending order; Hpp symmetric, bp
    for (int k = (int)blockIdx.x; k < g.n_pose; k += (int)gridDim.x)
        if (threadIdx.x < 27) {
            const int t = (int)threadIdx.x;
            double v = 0.0;
            for (int c = g.chunk_start[k]; c < g.chunk_start[k + 1]; ++c) v += g.pose_part[27 * (size_t)c + t];
            // term t of the upper triangle's rows (a, a .. 5), each followed by the row's right-hand side entry
            int a = 0, rem = t;
            while (rem >= 7 - a) {
                rem -= 7 - a;
                ++a;
            }
            if (rem == 6 - a) {
                bp[6 * (size_t)k + a] = v;
            } else {
                const int b = a + rem;
                Hpp[36 * (size_t)k + 6 * a + b] = v;
                Hpp[36 * (size_t)k + 6 * b + a] = v;
            }
        }
    lin_landmark_wg<kModel>(g, (int)blockIdx.x, poses, points, huber_mono, huber_stereo, Hll, bl, lm_chi, s_c, s_d);
}

// Both halves in ONE launch again (round 6, late; round 5's merged k_linearize was split because the keyframe side's 218 VGPRs set the budget of
// every workgroup -- lin_pose_half needs 78, the landmark side 70): 2 n_chunks keyframe workgroups (half a chunk each) and n_lm_wg landmark
// workgroups, interleaved one by one while both kinds last so that the keyframe side's stream of Hpl stores (144 bytes per edge: it is bound by
// them) and the landmark side's LDS reductions share the chip instead of taking turns. The keyframes' blocks are finished (halves added in
// ascending order) by k_reduce_scalars, the launch behind this one.
template <int kModel>
__global__ __launch_bounds__(256) void k_linearize2(GraphDev g, const double* __restrict__ poses, const double* __restrict__ points, double huber_mono,
                                                   double huber_stereo, double* __restrict__ Hpl, double* __restrict__ Hll, double* __restrict__ bl,
                                                   double* __restrict__ lm_chi, double* __restrict__ chi2) {
    // chi2[2], the largest |diagonal entry|, is a maximum the workgroups of k_reduce_scalars (the next launch) put together with atomics: cleared here
    if (blockIdx.x == 0 && threadIdx.x == 0) chi2[2] = 0.0;
    // one buffer, two views: the landmark side's s_c[14][kLmSlots + 1] | s_d[5][256]; the keyframe side's s_h[4][64 x 9] (double2) | s_part[4][27]
    constexpr int kRaw = 14 * (kLmSlots + 1) + 5 * 256;
    static_assert(kRaw >= 2 * 4 * 64 * 9 + 4 * 27, "the keyframe side's view fits the landmark side's");
    __shared__ __attribute__((aligned(16))) double s_raw[kRaw];
    const int A = 2 * g.n_chunks, B = g.n_lm_wg, lo = min(A, B), b = (int)blockIdx.x;
    const bool paired = b < 2 * lo;
    const bool pose_side = paired ? (b & 1) == 0 : A > B;   // (workgroup-uniform)
    const int idx = paired ? b >> 1 : b - lo;
    if (pose_side)
        lin_pose_half<kModel>(g, idx, poses, points, huber_mono, huber_stereo, Hpl, reinterpret_cast<double (*)[27]>(s_raw + 2 * 4 * 64 * 9),
                              reinterpret_cast<double2*>(s_raw));
    else
        lin_landmark_wg<kModel>(g, idx, poses, points, huber_mono, huber_stereo, Hll, bl, lm_chi, reinterpret_cast<double (*)[kLmSlots + 1]>(s_raw),
                                reinterpret_cast<double (*)[256]>(s_raw + 14 * (kLmSlots + 1)));
}

// term t of keyframe k: its half chunks' sums (k_linearize2) added in ascending order, up to 24 loads in flight (a keyframe of 5000 observations
// has 20 half chunks: with eight in flight the launch took 15.7 us at 200 keyframes x 5000 observations, three dependent round trips per sum)
__device__ __forceinline__ double pose_term_sum(const GraphDev& g, const int k, const int t) {
    double v = 0.0;
    const int c1 = 2 * g.chunk_start[k + 1];
    for (int c = 2 * g.chunk_start[k]; c < c1; c += 24) {
        double p[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) p[u] = c + u < c1 ? g.pose_part[27 * (size_t)(c + u) + t] : 0.0;
#pragma unroll
        for (int u = 0; u < 24; ++u)
            if (c + u < c1) v += p[u];
    }
    return v;
}

// What the ten-level LDS tree (s[t] += s[t + w], w = 512 .. 1, a barrier per level) leaves in s[0], with ONE barrier: lane t of wave 0 folds
// s[t + 64 j], j = 0 .. 15, in the tree's own pairing (levels 512 .. 64 pair j with j + 8, 4, 2, 1), the last six levels are lane shifts
// (lane t < w adds lane t + w): the same additions on the same operands, the same bits; twenty barriers of sixteen waves less per launch.
// All 1024 threads call it; the result is thread 0's. The caller puts a barrier before it reuses s.
__device__ __forceinline__ double tree_sum_1024(double* __restrict__ s, const double v) {
    s[threadIdx.x] = v;
    __syncthreads();
    double x = 0.0;
    if (threadIdx.x < 64) {
        double u[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) u[j] = s[threadIdx.x + 64 * j];
#pragma unroll
        for (int h = 8; h >= 1; h >>= 1)
#pragma unroll
            for (int j = 0; j < h; ++j) u[j] = u[j] + u[j + h];
        x = u[0];
#pragma unroll
        for (int w = 32; w >= 1; w >>= 1) x = x + __shfl_down(x, w);
    }
    return x;
}

// chi2[0..1] = sum of the per-landmark partials; chi2[2] = max |diagonal| over free pose blocks and landmarks with edges (g2o's
// computeLambdaInit); one workgroup, fixed order. With `lm_scale` (a Levenberg-Marquardt trial: the landmarks' terms of the gain ratio's
// denominator, written by the back-substitution) their sum goes to scale_sum[0] -- the additions of the former k_sum_1024, in its order.
__global__ __launch_bounds__(1024) void k_reduce_scalars(GraphDev g, const double* __restrict__ lm_chi, const double* Hpp,   // (Hpp may be Hpp_out: no restrict)
                                                        const double* __restrict__ Hll, double* __restrict__ chi2, double* __restrict__ mirror,
                                                        const double* __restrict__ lm_scale, double* __restrict__ scale_sum,
                                                        unsigned long long* __restrict__ host_ll, unsigned int seq, const int32_t* __restrict__ fail2,
                                                        double* Hpp_out = nullptr, double* __restrict__ bp_out = nullptr) {
    // host_ll (round 6, the device solver's LM trials): the trial's outcome -- the gain ratio's two parts, the chi2 triple, the two failure words --
    // also goes straight into a page-locked block as twelve 64-bit words {seq : 32 | half of a double : 32}; the host polls them instead of
    // enqueuing a 264-byte D2H copy and waiting for the stream (a copy kernel, its launch gap and the wait's wake-up per trial). A word whose
    // upper half is this trial's sequence number carries this trial's data: no fence (a system-scope release here would write back every dirty
    // line of the L2 -- the ~50 us per trial that sank "one kernel writes the values into the page-locked block" in round 4).
    __shared__ double s0[1024], s1[1024], s2[1024];
    if (Hpp_out && blockIdx.x > 0) {
        // behind k_linearize2, workgroups 1 ..: the keyframes' blocks from the half chunks' sums (halves in ascending order); Hpp symmetric, bp.
        // (One workgroup doing this as well was a serial tail of 27 n_pose sums of up to 40 dependent loads: 1 M edges 0.162 against 0.124 ms.)
        // The free keyframes' diagonal entries also go into chi2[2] = the largest |diagonal entry| (g2o's computeLambdaInit), as an atomic maximum
        // on the bit pattern (non-negative doubles order like integers; a maximum does not depend on the order of its operands): workgroup 0 used
        // to add up those 6 n_pose sums a second time for itself -- a third of this launch's 13.6 us at 200 keyframes x 5000 observations.
        double dmax = 0.0;
        for (int item = ((int)blockIdx.x - 1) * 1024 + (int)threadIdx.x; item < 27 * g.n_pose; item += ((int)gridDim.x - 1) * 1024) {
            const int k = item / 27, t = item - 27 * k;
            const double v = pose_term_sum(g, k, t);
            // term t of the upper triangle's rows (a, a .. 5), each followed by the row's right-hand side entry
            int a = 0, rem = t;
            while (rem >= 7 - a) {
                rem -= 7 - a;
                ++a;
            }
            if (rem == 6 - a) {
                bp_out[6 * (size_t)k + a] = v;
            } else {
                const int b = a + rem;
                Hpp_out[36 * (size_t)k + 6 * a + b] = v;
                Hpp_out[36 * (size_t)k + 6 * b + a] = v;
                if (rem == 0 && !g.fixed[k]) dmax = fmax(dmax, fabs(v));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off));
        if ((threadIdx.x & 63) == 0 && dmax > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(chi2 + 2), (unsigned long long)__double_as_longlong(dmax));
        return;
    }
    if (lm_scale) {   // (uniform) before the chi2 sums: s0 is reused
        double a = 0;
        for (int j0 = threadIdx.x; j0 < g.n_pt; j0 += 4 * 1024) {   // four loads in flight, added in the plain loop's order
            double c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = j0 + 1024 * u < g.n_pt ? lm_scale[j0 + 1024 * u] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + 1024 * u < g.n_pt) a += c[u];
        }
        const double tot = tree_sum_1024(s0, a);
        if (threadIdx.x == 0) scale_sum[0] = tot;
        __syncthreads();   // (s0 is reused below)
    }
    double a = 0, b = 0, m = 0;
    // the landmark workgroups' shares (k_lin_landmark), a thread's in ascending order
    for (int w0 = threadIdx.x; w0 < g.n_lm_wg; w0 += 8 * 1024) {   // eight triples in flight (a million edges: 4000 workgroups' shares, four
        double ca[8], cb[8], cm[8];                                 // dependent round trips with one at a time), added in the plain loop's order
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = w0 + 1024 * u < g.n_lm_wg;
            const size_t w = in ? (size_t)(w0 + 1024 * u) : 0;
            ca[u] = in ? lm_chi[3 * w] : 0.0;
            cb[u] = in ? lm_chi[3 * w + 1] : 0.0;
            cm[u] = in ? lm_chi[3 * w + 2] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (w0 + 1024 * u < g.n_lm_wg) {
                a += ca[u];
                b += cb[u];
                m = fmax(m, cm[u]);
            }
    }
    if (Hpp_out) {   // (uniform) the keyframes' diagonal entries reach chi2[2] from the workgroups that add them up (above): the landmarks' part here
    } else {
        for (int k = threadIdx.x; k < g.n_pose; k += 1024)
            if (!g.fixed[k])
                for (int d = 0; d < 6; ++d) m = fmax(m, fabs(Hpp[36 * (size_t)k + 7 * d]));
    }
    s1[threadIdx.x] = b;
    s2[threadIdx.x] = m;
    const double ta = tree_sum_1024(s0, a);   // (its barrier also covers s1 and s2)
    double tb = 0.0, tm = 0.0;
    if (threadIdx.x < 64) {
        double u[16], x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            u[j] = s1[threadIdx.x + 64 * j];
            x[j] = s2[threadIdx.x + 64 * j];
        }
#pragma unroll
        for (int h = 8; h >= 1; h >>= 1)
#pragma unroll
            for (int j = 0; j < h; ++j) {
                u[j] = u[j] + u[j + h];
                x[j] = fmax(x[j], x[j + h]);
            }
        tb = u[0];
        tm = x[0];
#pragma unroll
        for (int w = 32; w >= 1; w >>= 1) {
            tb = tb + __shfl_down(tb, w);
            tm = fmax(tm, __shfl_down(tm, w));
        }
    }
    if (threadIdx.x == 0) {
        s0[0] = ta;   // (read by the result words below)
        s1[0] = tb;
        s2[0] = tm;
        chi2[0] = ta;
        chi2[1] = tb;
        if (Hpp_out) {
            if (tm > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(chi2 + 2), (unsigned long long)__double_as_longlong(tm));
        } else {
            chi2[2] = tm;
        }
        if (mirror) {   // a second copy next to the solver's scalars: ONE download brings a Levenberg-Marquardt trial's outcome back
            mirror[0] = ta;
            mirror[1] = tb;
            mirror[2] = tm;   // (behind k_linearize2: the landmarks' part only -- no reader: an LM trial consumes [0], [1] and the gain ratio's parts)
        }
    }
    __syncthreads();
    if (host_ll && threadIdx.x < 12) {
        // values: [0] landmarks' / [1] keyframes' gain-ratio parts (scale_sum[0] was written by thread 0 above: same workgroup, behind barriers;
        // scale_sum[1] by k_trial_update), [2..4] chi2 triple, [5] the two failure words
        const int i = threadIdx.x >> 1;
        unsigned long long bits;
        if (i == 0) bits = (unsigned long long)__double_as_longlong(scale_sum ? scale_sum[0] : 0.0);
        else if (i == 1) bits = (unsigned long long)__double_as_longlong(scale_sum ? scale_sum[1] : 0.0);
        else if (i == 2) bits = (unsigned long long)__double_as_longlong(s0[0]);
        else if (i == 3) bits = (unsigned long long)__double_as_longlong(s1[0]);
        else if (i == 4) bits = (unsigned long long)__double_as_longlong(s2[0]);
        else bits = (unsigned long long)(uint32_t)fail2[0] | ((unsigned long long)(uint32_t)fail2[1] << 32);
        const unsigned long long half = (threadIdx.x & 1) ? (bits >> 32) : (bits & 0xffffffffull);
        __hip_atomic_store(&host_ll[threadIdx.x], ((unsigned long long)seq << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- landmark elimination ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool inv3_sym_d(const double* H, double lambda, double* out) {   // (H + lambda I)^-1 by cofactors
    const double a = H[0] + lambda, b = H[1], c = H[2], d = H[4] + lambda, e = H[5], f = H[8] + lambda;
    const double A = d * f - e * e, B = c * e - b * f, Cc = b * e - c * d;
    const double det = (a * A + b * B) + c * Cc;
    if (!(fabs(det) > 0.0) || !isfinite(det)) return false;
    const double id = 1.0 / det;
    out[0] = A * id;
    out[1] = out[3] = B * id;
    out[2] = out[6] = Cc * id;
    out[4] = (a * f - c * c) * id;
    out[5] = out[7] = (b * c - a * e) * id;
    out[8] = (a * d - b * b) * id;
    return true;
}

// One thread per EDGE (round 4; a thread per landmark walking its ~5 edges left the launch at 157 workgroups and 33 us): every edge inverts its
// landmark's damped block itself (30 flops on 72 bytes that sit in L2) and the landmark's first edge stores it; threads t < n_pt also cover the
// landmarks without edges, whose inverse k_backsub still reads. The same expressions as before: identical bits.
__global__ __launch_bounds__(256) void k_lm_prepare(GraphDev g, const double* __restrict__ Hll, const double* __restrict__ Hpl, double lambda,
                                                   double* __restrict__ Hinv, double* __restrict__ Y, int32_t* __restrict__ fail) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < g.n_pt && g.lm_start[t + 1] == g.lm_start[t]) {
        double Hi[9];
        if (!inv3_sym_d(Hll + 9 * (size_t)t, lambda, Hi)) *fail = 1;   // benign race: every writer stores 1
        else
#pragma unroll
            for (int i = 0; i < 9; ++i) Hinv[9 * (size_t)t + i] = Hi[i];
    }
    // Round 6, late: Y leaves through LDS -- a lane storing its own 144-byte record writes nine 16-byte pieces, each store instruction touching 64
    // different lines (0.9 M partial-line write requests per launch at config 5); a wave's 64 records are 9216 consecutive bytes: 14.0 -> 10.4 us.
    __shared__ double2 s_y[4 * 64 * 9];
    const int lane = (int)threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int e = t;
    bool ok = e < g.n_edge;
    double2 yv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) yv[i] = double2{0.0, 0.0};
    if (ok) {
        const int j = g.edges[e].pt;
        double Hi[9];
        if (!inv3_sym_d(Hll + 9 * (size_t)j, lambda, Hi)) {
            *fail = 1;
            ok = false;
        } else {
            if (g.lm_edges[g.lm_start[j]] == e) {
#pragma unroll
                for (int i = 0; i < 9; ++i) Hinv[9 * (size_t)j + i] = Hi[i];
            }
            const double* W = Hpl + 18 * (size_t)e;
            double y[18];
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) y[3 * a + c] = (W[3 * a] * Hi[c] + W[3 * a + 1] * Hi[3 + c]) + W[3 * a + 2] * Hi[6 + c];
#pragma unroll
            for (int i = 0; i < 9; ++i) yv[i] = double2{y[2 * i], y[2 * i + 1]};
        }
    }
    if (__ballot(ok) == ~0ull) {   // (wave-uniform) the usual case: 64 records, all computed
        double2* const sh = s_y + (size_t)wv * (64 * 9);
#pragma unroll
        for (int i = 0; i < 9; ++i) sh[lane * 9 + i] = yv[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double2* const out = reinterpret_cast<double2*>(Y + 18 * (size_t)(e - lane));
#pragma unroll
        for (int i = 0; i < 9; ++i) out[i * 64 + lane] = sh[i * 64 + lane];
    } else if (ok) {   // (a record whose landmark block could not be inverted stays unwritten, as before: the trial is reported failed)
        double2* const out = reinterpret_cast<double2*>(Y + 18 * (size_t)e);
#pragma unroll
        for (int i = 0; i < 9; ++i) out[i] = yv[i];
    }
}

// edge_of[s * n_pt + j] = the edge of free keyframe s (slot order) to landmark j, or -1: what k_schur_pairs intersects two keyframes'
// observations with (a keyframe observes a landmark at most once: checked at graph creation)
__global__ __launch_bounds__(256) void k_edge_table(const GEdge* __restrict__ edges, int n_edge, const int32_t* __restrict__ slot_of_pose, int n_pt,
                                                   int32_t* __restrict__ edge_of) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edge) return;
    const int s = slot_of_pose[edges[e].pose];
    if (s >= 0) edge_of[(size_t)s * n_pt + edges[e].pt] = e;
}

// S block (a, b), a <= b in slot order: S_ab = [a == b](Hpp_a + lambda I) - sum over the landmarks both keyframes observe of Y_ea W_eb^T.
// One workgroup per pair, no host-built pair lists (round 4; until then they cost 3 ms of every graph build and 2.4 MB of its upload at
// config 5): each of the four waves walks a contiguous quarter of keyframe a's observations 64 at a time, looks each landmark up in
// keyframe b's row of edge_of, and queues the common ones in LDS in list order; whenever 64 are queued every lane takes one (108
// multiply-adds on two 18-double records), so the arithmetic runs with all lanes busy whatever the overlap of the two keyframes. Lane l
// accumulates queue entries l, l + 64, ...; the 64 partial blocks of a wave are folded by a fixed xor tree, the four waves' blocks in wave
// order: the same bits from run to run. (One wave per pair walked the whole list in 32 dependent steps: 107 us per launch at config 5.)
__device__ __forceinline__ void schur_pair(const int pr, const int32_t* __restrict__ pose_start, const int32_t* __restrict__ pose_edges,
                                           const int32_t* __restrict__ pose_pt, const int32_t* __restrict__ pair_ab,
                                           const int32_t* __restrict__ slot_pose, const int32_t* __restrict__ edge_of, int n_pt,
                                           const double* __restrict__ Hpp, const double* __restrict__ Hpl, const double* __restrict__ Y,
                                           double lambda, int pitch, double* __restrict__ S, int2 (*s_queue)[128], double (*s_part)[36]) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int sa = pair_ab[2 * pr], sb = pair_ab[2 * pr + 1];
    const int ka = slot_pose[sa];
    const int32_t* const tb = edge_of + (size_t)sb * n_pt;
    int2* const q = s_queue[wave];
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.0;
    auto take = [&](int cnt) {   // lanes < cnt: one queued (edge of a, edge of b) each
        if (lane < cnt) {
            const int2 en = q[lane];
            const double* y = Y + 18 * (size_t)en.x;
            const double* W2 = Hpl + 18 * (size_t)en.y;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) acc[6 * a + b] += (y[3 * a] * W2[3 * b] + y[3 * a + 1] * W2[3 * b + 1]) + y[3 * a + 2] * W2[3 * b + 2];
        }
    };
    int qn = 0;
    const int i_lo = pose_start[ka], i_hi = pose_start[ka + 1];
    const int quarter = ((i_hi - i_lo + 255) >> 8) << 6;   // a multiple of 64
    const int i0 = i_lo + wave * quarter, i1 = min(i0 + quarter, i_hi);
    for (int base = i0; base < i1; base += 64) {
        const int i = base + lane;
        int ea = 0, eb = -1;
        if (i < i1) {
            ea = pose_edges[i];
            eb = tb[pose_pt[i]];
        }
        const unsigned long long bal = __ballot(eb >= 0);
        if (eb >= 0) q[qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = int2{ea, eb};
        qn += __popcll(bal);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (qn >= 64) {
            take(64);
            const int2 rest = q[64 + lane];   // (lanes >= qn - 64 read stale entries that nobody uses)
            __builtin_amdgcn_wave_barrier();
            q[lane] = rest;
            qn -= 64;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    take(qn);
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        const double x = wave_sum_lane63(acc[i]);   // (round 6: data-parallel moves instead of 12 ds_bpermute per value -- 432 of a wave's 461 LDS instructions)
        if (lane == 63) s_part[wave][i] = x;
    }
    __syncthreads();
    if (threadIdx.x < 36) {
        const int t = threadIdx.x, a = t / 6, b = t - 6 * a;
        double v = -(((s_part[0][t] + s_part[1][t]) + s_part[2][t]) + s_part[3][t]);
        if (sa == sb) v = (Hpp[36 * (size_t)ka + 6 * a + b] + (a == b ? lambda : 0.0)) + v;
        S[(size_t)(6 * sa + a) * pitch + 6 * sb + b] = v;
        if (sa != sb) S[(size_t)(6 * sb + b) * pitch + 6 * sa + a] = v;
    }
}

// g_s = bp_k - sum over keyframe k's edges of Y_e bl_j (k = the keyframe of block s), one workgroup
__device__ __forceinline__ void schur_rhs(const GraphDev& g, const int s, const int32_t* __restrict__ slot_pose, const double* __restrict__ bp,
                                          const double* __restrict__ bl, const double* __restrict__ Y, double* __restrict__ rhs, double (*s_part)[6]) {
    const int k = slot_pose[s];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = g.pose_start[k] + (int)threadIdx.x; i < g.pose_start[k + 1]; i += 256) {
        const int e = g.pose_edges[i];
        const double* y = Y + 18 * (size_t)e;
        const double* b = bl + 3 * (size_t)g.edges[e].pt;
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[a] += (y[3 * a] * b[0] + y[3 * a + 1] * b[1]) + y[3 * a + 2] * b[2];
    }
    block_sum_256<6>(acc, s_part);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 6; ++a) rhs[6 * s + a] = bp[6 * (size_t)k + a] - acc[a];
    }
}

// The reduced camera system in ONE launch (round 5): workgroups [0, n_free) form the right-hand side rows, the others one block (a, b) of S each.
// Both read Y and write disjoint parts of the system; as two launches the 48 workgroups of the right-hand side had the chip to themselves for 15 us.
__global__ __launch_bounds__(256) void k_schur(GraphDev g, int n_free, const int32_t* __restrict__ pose_pt, const int32_t* __restrict__ pair_ab,
                                              const int32_t* __restrict__ slot_pose, const int32_t* __restrict__ edge_of, const double* __restrict__ Hpp,
                                              const double* __restrict__ bp, const double* __restrict__ bl, const double* __restrict__ Hpl,
                                              const double* __restrict__ Y, double lambda, int pitch, double* __restrict__ S, double* __restrict__ rhs) {
    __shared__ int2 s_queue[4][128];
    __shared__ double s_part[4][36];
    __shared__ double s_part6[4][6];
    if ((int)blockIdx.x < n_free)   // (workgroup-uniform)
        schur_rhs(g, (int)blockIdx.x, slot_pose, bp, bl, Y, rhs, s_part6);
    else
        schur_pair((int)blockIdx.x - n_free, g.pose_start, g.pose_edges, pose_pt, pair_ab, slot_pose, edge_of, g.n_pt, Hpp, Hpl, Y, lambda, pitch, S,
                   s_queue, s_part);
}

// ---- round 6, late: the pairs' common landmarks found ONCE per graph ----------------------------------------------------------------
// schur_pair's scan (every wave walks a quarter of keyframe a's observations and looks each landmark up in keyframe b's row of edge_of) does
// not depend on the trial: 23 of k_schur's 61 us at config 5 went into repeating it fifteen times per call. k_pair_lists runs it twice when the
// solver work space is created -- counting, then (behind an exclusive scan of the 4 n_pairs counts) writing the (edge of a, edge of b) entries of
// every (pair, wave) in the order the queue would have met them --, and k_schur_l's lane l of wave w takes entries l, l + 64, ... of ITS list:
// the entries the queue handed that lane, in that order, so S keeps its bits.
template <bool kFill>
__global__ __launch_bounds__(256) void k_pair_lists(const int32_t* __restrict__ pose_start, const int32_t* __restrict__ pose_edges,
                                                   const int32_t* __restrict__ pose_pt, const int32_t* __restrict__ pair_ab,
                                                   const int32_t* __restrict__ slot_pose, const int32_t* __restrict__ edge_of, int n_pt,
                                                   int32_t* __restrict__ cnt, const int32_t* __restrict__ off, int2* __restrict__ ent) {
    const int pr = (int)blockIdx.x, lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int sa = pair_ab[2 * pr], sb = pair_ab[2 * pr + 1];
    const int ka = slot_pose[sa];
    const int32_t* const tb = edge_of + (size_t)sb * n_pt;
    const int i_lo = pose_start[ka], i_hi = pose_start[ka + 1];
    const int quarter = ((i_hi - i_lo + 255) >> 8) << 6;   // (schur_pair's split)
    const int i0 = i_lo + wave * quarter, i1 = min(i0 + quarter, i_hi);
    int qn = 0;
    const int at = kFill ? off[4 * pr + wave] : 0;
    for (int base = i0; base < i1; base += 64) {
        const int i = base + lane;
        int ea = 0, eb = -1;
        if (i < i1) {
            ea = pose_edges[i];
            eb = tb[pose_pt[i]];
        }
        const unsigned long long bal = __ballot(eb >= 0);
        if (kFill && eb >= 0)
            ent[at + qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = int2{ea, eb};
        qn += __popcll(bal);
    }
    if (!kFill && lane == 0) cnt[4 * pr + wave] = qn;
}

// off[i] = cnt[0] + ... + cnt[i - 1], i = 0 .. n (one workgroup; n = 4 n_pairs is a few thousand)
__global__ __launch_bounds__(1024) void k_scan_i32(const int32_t* __restrict__ cnt, int n, int32_t* __restrict__ off) {
    __shared__ int32_t s_sum[1024];
    const int tid = (int)threadIdx.x, per = (n + 1023) / 1024, lo = min(tid * per, n), hi = min(lo + per, n);
    int32_t t = 0;
    for (int i = lo; i < hi; ++i) t += cnt[i];
    s_sum[tid] = t;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int32_t v = tid >= d ? s_sum[tid - d] : 0;
        __syncthreads();
        s_sum[tid] += v;
        __syncthreads();
    }
    int32_t run = s_sum[tid] - t;
    for (int i = lo; i < hi; ++i) {
        off[i] = run;
        run += cnt[i];
    }
    if (tid == 1023) off[n] = s_sum[1023];
}

// schur_pair on the lists
template <bool kCoop>
__device__ __forceinline__ void schur_pair_l(const int pr, const int32_t* __restrict__ pair_ab, const int32_t* __restrict__ slot_pose,
                                             const int32_t* __restrict__ off, const int2* __restrict__ ent, const double* __restrict__ Hpp,
                                             const double* __restrict__ Hpl, const double* __restrict__ Y, double lambda, int pitch,
                                             double* __restrict__ S, double (*s_part)[36], double2* __restrict__ s_rec) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int sa = pair_ab[2 * pr], sb = pair_ab[2 * pr + 1];
    const int ka = slot_pose[sa];
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.0;
    const int s1 = off[4 * pr + wave + 1];
    if (!kCoop) {
        for (int idx = off[4 * pr + wave] + lane; idx < s1; idx += 64) {
            const int2 en = ent[idx];
            const double* y = Y + 18 * (size_t)en.x;
            const double* W2 = Hpl + 18 * (size_t)en.y;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) acc[6 * a + b] += (y[3 * a] * W2[3 * b] + y[3 * a + 1] * W2[3 * b + 1]) + y[3 * a + 2] * W2[3 * b + 2];
        }
    } else {
        // A lane reading its own two 144-byte records issues 18 loads that each touch 64 different lines: the address unit of the CU takes a line
        // per clock, 2300 clocks per batch of 64 entries -- and a diagonal pair (a, a) has all of a's ~2000 edges, eight batches per wave on ONE
        // CU: the launch waited for those 48 workgroups. Here the wave fetches the batch's 2 x 64 records as 2 x 576 16-byte pieces, consecutive
        // lanes taking consecutive pieces of a record (seven records per instruction, ~14 lines), through LDS; every lane then reads its own
        // records from there (stride 144 bytes: conflict-free for 16-byte reads). The products and their order are those of the loop above.
        double2* const buf = s_rec + (size_t)wave * (64 * 9);
        const double2* const Y2 = reinterpret_cast<const double2*>(Y);
        const double2* const H2 = reinterpret_cast<const double2*>(Hpl);
        const int s0 = off[4 * pr + wave];
        int2 en_next = int2{0, 0};
        if (s0 + lane < s1) en_next = ent[s0 + lane];
        for (int base = s0; base < s1; base += 64) {
            const int cnt = min(64, s1 - base);
            const int2 en = en_next;
            if (base + 64 + lane < s1) en_next = ent[base + 64 + lane];   // (the next batch's entries travel under this batch's records: a round trip less per batch)
            double2 ly[9], lw[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int p = 64 * r + lane, rec = (p * 7282) >> 16, part = p - 9 * rec;   // rec = p / 9 for p < 576
                const int ea = __shfl(en.x, rec), eb = __shfl(en.y, rec);
                ly[r] = double2{0.0, 0.0};
                lw[r] = double2{0.0, 0.0};
                if (rec < cnt) {
                    ly[r] = Y2[(size_t)ea * 9 + part];
                    lw[r] = H2[(size_t)eb * 9 + part];
                }
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) buf[64 * r + lane] = ly[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double2 y2[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) y2[i] = buf[9 * lane + i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 9; ++r) buf[64 * r + lane] = lw[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < cnt) {
                const double y[18] = {y2[0].x, y2[0].y, y2[1].x, y2[1].y, y2[2].x, y2[2].y, y2[3].x, y2[3].y, y2[4].x, y2[4].y,
                                      y2[5].x, y2[5].y, y2[6].x, y2[6].y, y2[7].x, y2[7].y, y2[8].x, y2[8].y};
#pragma unroll
                for (int bp = 0; bp < 3; ++bp) {   // rows 2 bp, 2 bp + 1 of W: doubles 6 bp .. 6 bp + 5
                    const double2 w0 = buf[9 * lane + 3 * bp], w1 = buf[9 * lane + 3 * bp + 1], w2 = buf[9 * lane + 3 * bp + 2];
                    const double w[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        acc[6 * a + 2 * bp] += (y[3 * a] * w[0] + y[3 * a + 1] * w[1]) + y[3 * a + 2] * w[2];
                        acc[6 * a + 2 * bp + 1] += (y[3 * a] * w[3] + y[3 * a + 1] * w[4]) + y[3 * a + 2] * w[5];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // (the next batch overwrites the buffer)
        }
    }
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        const double x = wave_sum_lane63(acc[i]);
        if (lane == 63) s_part[wave][i] = x;
    }
    __syncthreads();
    if (threadIdx.x < 36) {
        const int t = threadIdx.x, a = t / 6, b = t - 6 * a;
        double v = -(((s_part[0][t] + s_part[1][t]) + s_part[2][t]) + s_part[3][t]);
        if (sa == sb) v = (Hpp[36 * (size_t)ka + 6 * a + b] + (a == b ? lambda : 0.0)) + v;
        S[(size_t)(6 * sa + a) * pitch + 6 * sb + b] = v;
        if (sa != sb) S[(size_t)(6 * sb + b) * pitch + 6 * sa + a] = v;
    }
}

// (212 VGPRs, two waves per SIMD; three forced by amdgpu_waves_per_eu: 172 bytes of scratch)
template <bool kCoop>
__global__ __launch_bounds__(256) void k_schur_l(GraphDev g, int n_free, const int32_t* __restrict__ pair_ab, const int32_t* __restrict__ slot_pose,
                                                const int32_t* __restrict__ off, const int2* __restrict__ ent, const double* __restrict__ Hpp,
                                                const double* __restrict__ bp, const double* __restrict__ bl, const double* __restrict__ Hpl,
                                                const double* __restrict__ Y, double lambda, int pitch, double* __restrict__ S,
                                                double* __restrict__ rhs, const int32_t* __restrict__ wg_pair, int n_pair_wg) {
    __shared__ double s_part[4][36];
    __shared__ double s_part6[4][6];
    __shared__ double2 s_rec[kCoop ? 4 * 64 * 9 : 1];   // a batch's 64 records per wave
    // grid: n_pair_wg pair workgroups, then n_free right-hand-side workgroups. wg_pair (graph_create) names the pair of every workgroup, -1 = none:
    // consecutive workgroups go to different XCDs (eight L2 caches), and the table gives every XCD whole ROWS a of the pair table, dealt so that
    // the eight get equal work -- the workgroups running side by side on an XCD then gather the same keyframe's Y records (and W records that
    // recur row after row) from ITS L2: in launch order k_schur_l fetched 94-190 MB per launch for 29 MB of distinct records.
    const int q = (int)blockIdx.x;
    if (q >= n_pair_wg) {   // (workgroup-uniform)
        schur_rhs(g, q - n_pair_wg, slot_pose, bp, bl, Y, rhs, s_part6);
    } else {
        const int pr = wg_pair ? wg_pair[q] : q;
        if (pr >= 0) schur_pair_l<kCoop>(pr, pair_ab, slot_pose, off, ent, Hpp, Hpl, Y, lambda, pitch, S, s_part, s_rec);
    }
}

// dxl_j = Hll^-1 (bl_j - sum_e W_e^T dxp[pose(e)]); X_trial = X + dxl; lm_scale[j] = dxl . (lambda dxl + bl_j)
// `dx`: the keyframes' increments, six per reduced block when `slot_of_pose` is given (the device solver's solution vector, read in place),
// else six per keyframe (the host solver's upload)
__device__ __forceinline__ void backsub_landmark(const GraphDev& g, const int j, const double* __restrict__ Hinv, const double* __restrict__ Hpl,
                                                 const double* __restrict__ bl, const double* __restrict__ dx,
                                                 const int32_t* __restrict__ slot_of_pose, double lambda, const double* __restrict__ X,
                                                 double* __restrict__ Xn, double* __restrict__ lm_scale) {
    double r[3] = {bl[3 * (size_t)j], bl[3 * (size_t)j + 1], bl[3 * (size_t)j + 2]};
    for (int i = g.lm_start[j]; i < g.lm_start[j + 1]; ++i) {
        const int e = g.lm_edges[i];
        const int k = g.edges[e].pose;
        if (g.fixed[k]) continue;
        const double* W = Hpl + 18 * (size_t)e;
        const double* d = dx + 6 * (size_t)(slot_of_pose ? slot_of_pose[k] : k);   // (k is free here: its slot is >= 0)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            r[c] -= ((W[c] * d[0] + W[3 + c] * d[1]) + (W[6 + c] * d[2] + W[9 + c] * d[3])) + (W[12 + c] * d[4] + W[15 + c] * d[5]);
    }
    const double* Hi = Hinv + 9 * (size_t)j;
    double sc = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double dl = (Hi[3 * c] * r[0] + Hi[3 * c + 1] * r[1]) + Hi[3 * c + 2] * r[2];
        Xn[3 * (size_t)j + c] = X[3 * (size_t)j + c] + dl;
        sc += dl * (lambda * dl + bl[3 * (size_t)j + c]);
    }
    lm_scale[j] = sc;
}

__global__ __launch_bounds__(128) void k_backsub(GraphDev g, const double* __restrict__ Hinv, const double* __restrict__ Hpl,
                                                const double* __restrict__ bl, const double* __restrict__ dxp, double lambda,
                                                const double* __restrict__ X, double* __restrict__ Xn, double* __restrict__ lm_scale) {
    const int j = (int)blockIdx.x * 128 + (int)threadIdx.x;
    if (j < g.n_pt) backsub_landmark(g, j, Hinv, Hpl, bl, dxp, nullptr, lambda, X, Xn, lm_scale);
}

// ---- LM trial state of the keyframes ------------------------------------------------------------------------------------------------
namespace {

__device__ void dev_rot_to_quat(const double* R, double* q) {   // ba_host_math.h rot_to_quat
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double s = sqrt(tr + 1.0);
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
        double s = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        double qq[4];
        qq[i] = 0.5 * s;
        s = 0.5 / s;
        qq[3] = (R[3 * k + j] - R[3 * j + k]) * s;
        qq[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        qq[k] = (R[3 * k + i] + R[3 * i + k]) * s;
        for (int a = 0; a < 4; ++a) q[a] = qq[a];
    }
    if (q[3] < 0)
        for (int a = 0; a < 4; ++a) q[a] = -q[a];
    const double nn = sqrt((q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]));
    for (int a = 0; a < 4; ++a) q[a] /= nn;
}

__device__ void dev_se3_oplus(const double* TR, const double* Tt, const double* u, double* nR, double* nt) {   // ba_host_math.h se3_oplus
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = sqrt((wx * wx + wy * wy) + wz * wz);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9], E[9], V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = (O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j]) + O[3 * i + 2] * O[6 + j];
    const bool small = theta < 0.00001;
    const double s = small ? 0.0 : sin(theta), c = small ? 1.0 : cos(theta);
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        if (small) {
            E[i] = (I + O[i]) + O2[i];
            V[i] = E[i];
        } else {
            E[i] = (I + s / theta * O[i]) + (1 - c) / (theta * theta) * O2[i];
            V[i] = (I + (1 - c) / (theta * theta) * O[i]) + (theta - s) / (theta * theta * theta) * O2[i];
        }
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) nR[3 * i + j] = (E[3 * i] * TR[j] + E[3 * i + 1] * TR[3 + j]) + E[3 * i + 2] * TR[6 + j];
        const double te = (V[3 * i] * u[3] + V[3 * i + 1] * u[4]) + V[3 * i + 2] * u[5];
        nt[i] = ((E[3 * i] * Tt[0] + E[3 * i + 1] * Tt[1]) + E[3 * i + 2] * Tt[2]) + te;
    }
}

}   // namespace

// T (12 doubles per keyframe: R row-major | t) -> Tn, the 7-double records p7n (t | quaternion x y z w), dxp (6 per keyframe, zeros for
// fixed ones) and scal_pose = sum over free keyframes, in keyframe order, of dx . (lambda dx + bp)  (the keyframes' part of g2o's
// computeScale; the landmarks' part comes from k_backsub)
__device__ __forceinline__ void pose_update(const double* __restrict__ T, const int32_t* __restrict__ slot_of_pose, int n_pose,
                                            const double* __restrict__ x, const double* __restrict__ bp, double lambda, double* __restrict__ Tn,
                                            double* __restrict__ p7n, double* __restrict__ dxp, double* __restrict__ scal_pose,
                                            double (*s_term)[7]) {
    double sc = 0;
    for (int base = 0; base < n_pose; base += 256) {
        const int k = base + (int)threadIdx.x;
        if (k < n_pose) {
            const int sl = slot_of_pose[k];
            double u[6] = {0, 0, 0, 0, 0, 0};
            double nR[9], nt[3];
            if (sl >= 0) {
                for (int a = 0; a < 6; ++a) u[a] = x[6 * (size_t)sl + a];
                dev_se3_oplus(T + 12 * (size_t)k, T + 12 * (size_t)k + 9, u, nR, nt);
            } else {
                for (int a = 0; a < 9; ++a) nR[a] = T[12 * (size_t)k + a];
                for (int a = 0; a < 3; ++a) nt[a] = T[12 * (size_t)k + 9 + a];
            }
            for (int a = 0; a < 9; ++a) Tn[12 * (size_t)k + a] = nR[a];
            for (int a = 0; a < 3; ++a) {
                Tn[12 * (size_t)k + 9 + a] = nt[a];
                p7n[7 * (size_t)k + a] = nt[a];
            }
            dev_rot_to_quat(nR, p7n + 7 * (size_t)k + 3);
            for (int a = 0; a < 6; ++a) {
                dxp[6 * (size_t)k + a] = u[a];
                s_term[threadIdx.x][a] = u[a] * (lambda * u[a] + bp[6 * (size_t)k + a]);
            }
            s_term[threadIdx.x][6] = sl >= 0 ? 1.0 : 0.0;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int cnt = min(256, n_pose - base);
            for (int i = 0; i < cnt; ++i)
                if (s_term[i][6] != 0.0)
                    for (int a = 0; a < 6; ++a) sc += s_term[i][a];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *scal_pose = sc;
}


// Back-substitution, one lane per EDGE (round 6): the landmarks of workgroup `wg` of k_lin_landmark's partition (whole landmarks, at most kLmSlots
// edges). A lane per landmark walking its edges (backsub_landmark) is a chain of four dependent loads per edge and 313 waves at config 5: 23 us
// of latency per Levenberg-Marquardt trial. Here every lane computes ITS slot's W_e^T dx (two dependent loads), and the landmark's lane then
// subtracts its slots' terms from bl_j in list order -- backsub_landmark's subtractions, in its order: the same bits (an edge of a fixed
// keyframe, which that loop skips, subtracts +0.0 here, which changes no value).
__device__ __forceinline__ void backsub_finish(const int j, const double (&r)[3], const double* __restrict__ Hinv, const double* __restrict__ bl,
                                               double lambda, const double* __restrict__ X, double* __restrict__ Xn, double* __restrict__ lm_scale) {
    const double* Hi = Hinv + 9 * (size_t)j;
    double sc = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double dl = (Hi[3 * c] * r[0] + Hi[3 * c + 1] * r[1]) + Hi[3 * c + 2] * r[2];
        Xn[3 * (size_t)j + c] = X[3 * (size_t)j + c] + dl;
        sc += dl * (lambda * dl + bl[3 * (size_t)j + c]);
    }
    lm_scale[j] = sc;
}
__device__ __forceinline__ void backsub_wg(const GraphDev& g, const int wg, const double* __restrict__ Hinv, const double* __restrict__ Hpl,
                                           const double* __restrict__ bl, const double* __restrict__ dx, const int32_t* __restrict__ slot_of_pose,
                                           double lambda, const double* __restrict__ X, double* __restrict__ Xn, double* __restrict__ lm_scale,
                                           double (*s_t)[kLmSlots + 1]) {
    const int tid = (int)threadIdx.x;
    const int j0 = g.lm_wg_first[wg], j1 = g.lm_wg_first[wg + 1], n_lm = j1 - j0;
    const int s0 = g.lm_start[j0], s1 = g.lm_start[j1];
    const bool big = s1 - s0 > kLmSlots;   // (workgroup-uniform) then n_lm == 1: the landmark's edges pass in pieces, thread 0 carries its residual
    double run[3] = {0.0, 0.0, 0.0};
    if (big && tid == 0)
        for (int c = 0; c < 3; ++c) run[c] = bl[3 * (size_t)j0 + c];
    for (int base = s0; base == s0 || base < s1; base += kLmSlots) {
        const int s = base + tid;
        if (s < s1) {
            const int sl = slot_of_pose[g.ledges[s].pose];
            double t[3] = {0.0, 0.0, 0.0};
            if (sl >= 0) {
                const double* W = Hpl + 18 * (size_t)g.lm_edges[s];
                const double* d = dx + 6 * (size_t)sl;
#pragma unroll
                for (int c = 0; c < 3; ++c) t[c] = ((W[c] * d[0] + W[3 + c] * d[1]) + (W[6 + c] * d[2] + W[9 + c] * d[3])) + (W[12 + c] * d[4] + W[15 + c] * d[5]);
            }
            s_t[0][tid] = t[0];
            s_t[1][tid] = t[1];
            s_t[2][tid] = t[2];
        }
        __syncthreads();
        if (!big) {
            if (tid < n_lm) {
                const int j = j0 + tid;
                double r[3] = {bl[3 * (size_t)j], bl[3 * (size_t)j + 1], bl[3 * (size_t)j + 2]};
                for (int i = g.lm_start[j] - base; i < g.lm_start[j + 1] - base; ++i) {
                    r[0] -= s_t[0][i];
                    r[1] -= s_t[1][i];
                    r[2] -= s_t[2][i];
                }
                backsub_finish(j, r, Hinv, bl, lambda, X, Xn, lm_scale);
            }
        } else if (tid == 0) {
            const int n = min(kLmSlots, s1 - base);
            for (int i = 0; i < n; ++i) {
                run[0] -= s_t[0][i];
                run[1] -= s_t[1][i];
                run[2] -= s_t[2][i];
            }
        }
        __syncthreads();
    }
    if (big && tid == 0) backsub_finish(j0, run, Hinv, bl, lambda, X, Xn, lm_scale);
}

// The trial state of a Levenberg-Marquardt step in ONE launch (round 5): workgroup 0 advances the keyframes, the others back-substitute the
// landmarks (kByEdge: one workgroup of k_lin_landmark's partition each, one lane per edge -- round 6; else 128 or 256 landmarks each, one lane per
// landmark). The landmarks read the keyframes' increments from the solver's solution vector (slot order) instead of the dxp array the
// keyframe workgroup writes, so the two halves are independent; as two launches the single keyframe workgroup held the queue for 9.5 us.
template <bool kByEdge>
__global__ __launch_bounds__(256) void k_trial_update(GraphDev g, const double* __restrict__ T, const int32_t* __restrict__ slot_of_pose,
                                                     const double* __restrict__ x, const double* __restrict__ bp, double lambda,
                                                     double* __restrict__ Tn, double* __restrict__ p7n, double* __restrict__ dxp,
                                                     double* __restrict__ scal_pose, const double* __restrict__ Hinv, const double* __restrict__ Hpl,
                                                     const double* __restrict__ bl, const double* __restrict__ X, double* __restrict__ Xn,
                                                     double* __restrict__ lm_scale, int32_t* __restrict__ next_fail, int lm_per_wg) {
    __shared__ double s_term[256][7];   // a keyframe's six products (and whether it is free): thread 0 adds them in keyframe order
    if (blockIdx.x == 0) {   // (workgroup-uniform)
        // the NEXT trial's failure word (the two words alternate): its last reader, the host, consumed it a trial ago -- a memset launch less per trial
        if (threadIdx.x == 0) *next_fail = 0;
        pose_update(T, slot_of_pose, g.n_pose, x, bp, lambda, Tn, p7n, dxp, scal_pose, s_term);
    } else if (kByEdge) {
        backsub_wg(g, (int)blockIdx.x - 1, Hinv, Hpl, bl, x, slot_of_pose, lambda, X, Xn, lm_scale,
                   reinterpret_cast<double (*)[kLmSlots + 1]>(&s_term[0][0]));   // (3 x 257 of the 1792 doubles)
    } else {
        const int j = ((int)blockIdx.x - 1) * lm_per_wg + (int)threadIdx.x;   // (landmarks_per_workgroup, as in k_linearize)
        if ((int)threadIdx.x < lm_per_wg && j < g.n_pt) backsub_landmark(g, j, Hinv, Hpl, bl, x, slot_of_pose, lambda, X, Xn, lm_scale);
    }
}

// per-edge chi2 = e^T Omega e (no kernel) and the sign of the depth (reproj_edge_wrapper::depth_is_positive)
__global__ __launch_bounds__(256) void k_edge_chi2(GraphDev g, const double* __restrict__ poses, const double* __restrict__ points,
                                                  double* __restrict__ chi2, uint8_t* __restrict__ depth_pos) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= g.n_edge) return;
    const GEdge ed = g.edges[e];
    const bool stereo = e >= g.n_mono;
    const double* P = poses + 7 * (size_t)ed.pose;
    const double* X = points + 3 * (size_t)ed.pt;
    const double qx = P[3], qy = P[4], qz = P[5], qw = P[6];
    const double tx2 = 2 * qx, ty2 = 2 * qy, tz2 = 2 * qz;
    const double twx = tx2 * qw, twy = ty2 * qw, twz = tz2 * qw;
    const double txx = tx2 * qx, txy = ty2 * qx, txz = tz2 * qx;
    const double tyy = ty2 * qy, tyz = tz2 * qy, tzz = tz2 * qz;
    const double x = (1 - (tyy + tzz)) * X[0] + (txy - twz) * X[1] + (txz + twy) * X[2] + P[0];
    const double y = (txy + twz) * X[0] + (1 - (txx + tzz)) * X[1] + (tyz - twx) * X[2] + P[1];
    const double z = (txz - twy) * X[0] + (tyz + twx) * X[1] + (1 - (txx + tyy)) * X[2] + P[2];
    if (g.model == 1) {   // equirectangular: reproj_edge_wrapper::depth_is_positive() is true for this camera model
        const double kPi = 3.14159265358979323846;
        const double L = sqrt((x * x + y * y) + z * z);
        const double theta = ovs_det_atan2(x, z);
        const double phi = -ovs_det_asin(y / L);
        const double q0 = ed.ox - g.cam.fx * (0.5 + theta / (2.0 * kPi));
        const double q1 = ed.oy - g.cam.fy * (0.5 - phi / kPi);
        chi2[e] = ed.w * (q0 * q0 + q1 * q1);
        depth_pos[e] = 1;
        return;
    }
    const double invz = 1.0 / z;
    const double u = g.cam.fx * x * invz + g.cam.cx;
    const double e0 = ed.ox - u, e1 = ed.oy - (g.cam.fy * y * invz + g.cam.cy);
    double ss = e0 * e0 + e1 * e1;
    if (stereo) {
        const double e2 = ed.oxr - (u - g.bf * invz);
        ss = ss + e2 * e2;
    }
    chi2[e] = ed.w * ss;
    depth_pos[e] = z > 0.0 ? 1 : 0;
}

// local_bundle_adjuster's chi-square gates on the device (round 6; until then both per-edge arrays came down -- 0.9 MB twice per call --, the host
// judged 100 k edges in two loops and sent the active mask back up). An edge is an outlier when thr < chi2 or its depth is not positive
// (upstream: `chi_sq_2D < edge->chi2() || !edge->depth_is_positive()`), thr by edge kind; the same double comparisons, so the same flags.
//   after round 1 (chi_r1 == nullptr):  out[e] from chi[e] / depth[e]; active[e] = !out[e] (level-1 edges are masked, ba_graph_set_active);
//                                        *n_active += the inliers (integer atomics: any order gives the same count);
//   final (chi_r1 != nullptr):           an edge optimised in round 2 (`use_final` and not a round-1 outlier) is judged by chi[e], any other by its
//                                        round-1 value chi_r1[e] (g2o does not recompute the error of an inactive edge); depth[e] is the final state's.
__global__ __launch_bounds__(256) void k_edge_gate(int n_edge, int n_mono, double thr_mono, double thr_stereo, const double* __restrict__ chi,
                                                  const uint8_t* __restrict__ depth, const double* __restrict__ chi_r1,
                                                  const uint8_t* __restrict__ out1, int use_final, uint8_t* __restrict__ out,
                                                  uint8_t* __restrict__ active, int32_t* __restrict__ n_active) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    bool inlier = false;
    if (e < n_edge) {
        const double thr = e < n_mono ? thr_mono : thr_stereo;
        const double c = chi_r1 ? ((use_final && !out1[e]) ? chi[e] : chi_r1[e]) : chi[e];
        const bool o = (thr < c) || !depth[e];
        out[e] = o ? 1 : 0;
        if (active) active[e] = o ? 0 : 1;
        inlier = !o;
    }
    if (n_active) {   // one atomic per workgroup (one per wave -- 1563 on one address at config 5 -- made this launch 20 us)
        __shared__ int32_t s_cnt[4];
        const unsigned long long b = __ballot(inlier);
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = (int32_t)__popcll(b);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int32_t c = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
            if (c) atomicAdd(n_active, c);
        }
    }
}

}   // namespace ovs

using namespace ovs;

// ---------------------------------------------------------------------------------------------------------------------------
// The graph handle: device copies of the edge set and its indices; no per-call allocation on the linearisation path.
// ---------------------------------------------------------------------------------------------------------------------------
struct ovs_ba_graph {
    int device = 0;
    int n_pose = 0, n_pt = 0, n_mono = 0, n_stereo = 0, n_free = 0;
    ovs_ba_cam cam{};
    double bf = 0;
    int model = 0;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> slot, slot_pose;   // pose -> reduced-system block (-1 fixed); block -> pose
    // device: ONE allocation + ONE upload per graph (a dozen hipMalloc / hipMemcpy pairs cost more than the kernels of a whole LM trial)
    unsigned char* d_arena = nullptr;
    unsigned char* d_solver_arena = nullptr;
    size_t arena_cap = 0, solver_cap = 0;   // allocation sizes (the arenas come from / go back to g_ba_pool)
    uint8_t* d_active = nullptr;
    GEdge* d_edges = nullptr;
    int32_t *d_lm_start = nullptr, *d_lm_edges = nullptr, *d_lm_nmono = nullptr, *d_pose_start = nullptr, *d_pose_edges = nullptr;
    uint8_t* d_fixed = nullptr;
    int32_t *d_pose_pt = nullptr, *d_pair_ab = nullptr, *d_slot_pose = nullptr, *d_slot_of_pose = nullptr, *d_fail = nullptr;
    int32_t *d_lm_of_slot = nullptr, *d_lm_wg_first = nullptr, *d_chunk_kf = nullptr, *d_chunk_start = nullptr;   // k_linearize's work partition
    double* d_pose_part = nullptr;
    GEdge* d_ledges = nullptr;
    int n_lm_wg = 0, n_chunks = 0;
    int32_t* d_edge_of = nullptr;   // [n_free x n_pt], solver arena
    // the pairs' common landmarks (k_pair_lists): [4 n_pairs + 1] offsets, one (edge of a, edge of b) per entry; pl_bound = sum over the landmarks of
    // n (n + 1) / 2, n = the landmark's edges (graph_create): no list can be longer
    int32_t *d_pl_cnt = nullptr, *d_pl_off = nullptr;
    int2* d_pl_ent = nullptr;
    size_t pl_bound = 0;
    bool pl_ready = false;   // the lists exist (solver work space created, bound within the cap)
    int n_pairs = 0;
    int32_t* d_wg_pair = nullptr;   // [n_pair_wg] k_schur_l's work order: whole rows of the pair table per XCD, -1 = no pair
    int n_pair_wg = 0;
    double* d_lm_tmp = nullptr;   // [4 n_pt] per-landmark partials: chi2 pair, max |diagonal|, the gain ratio's scale term
    // solver work space (allocated on first use: ovs_ba_graph_linearize_dev alone does not need it)
    double *d_Hinv = nullptr, *d_Y = nullptr, *d_S = nullptr, *d_rhs = nullptr, *d_dxp = nullptr, *d_scal = nullptr;
    int s_pitch = 0;   // doubles per row of d_S

    GraphDev view() const {
        GraphDev g;
        g.edges = d_edges;
        g.n_mono = n_mono;
        g.n_edge = n_mono + n_stereo;
        g.n_pose = n_pose;
        g.n_pt = n_pt;
        g.lm_start = d_lm_start;
        g.lm_edges = d_lm_edges;
        g.lm_nmono = d_lm_nmono;
        g.pose_start = d_pose_start;
        g.pose_edges = d_pose_edges;
        g.pose_pt = d_pose_pt;
        g.fixed = d_fixed;
        g.active = d_active;
        g.ledges = d_ledges;
        g.lm_of_slot = d_lm_of_slot;
        g.lm_wg_first = d_lm_wg_first;
        g.chunk_kf = d_chunk_kf;
        g.chunk_start = d_chunk_start;
        g.pose_part = d_pose_part;
        g.n_lm_wg = n_lm_wg;
        g.n_chunks = n_chunks;
        g.cam = cam;
        g.bf = bf;
        g.model = model;
        return g;
    }
    int n_edge() const { return n_mono + n_stereo; }
};

namespace {

// landmarks per 256-thread workgroup of k_linearize / k_trial_update: 128 (the upper two waves leave at once) while the launch then still fits the
// chip in one go -- a landmark is a chain of dependent loads, so twice the workgroups on twice the compute units finish sooner --, 256 for larger maps
static int landmarks_per_workgroup(int n_pose, int n_pt) {
    static const int forced = [] {
        const char* e = std::getenv("OVS_BA_LM_PER_WG");   // A/B switch: 128 | 256
        return e ? std::atoi(e) : 0;
    }();
    if (forced == 128 || forced == 256) return forced;
    return n_pose + (n_pt + 127) / 128 <= 512 ? 128 : 256;
}

// `trial_scale`: the linearisation closes a Levenberg-Marquardt trial -- the landmarks' gain-ratio terms the back-substitution left in
// d_lm_tmp[3 n_pt ..) are summed into d_scal[0] by the same launch that sums chi2
ovs_status graph_linearize(ovs_ba_graph* g, const double* d_poses, const double* d_points, double huber_mono, double huber_stereo, double* d_Hpp,
                           double* d_bp, double* d_Hll, double* d_bl, double* d_Hpl, double* d_chi3, hipStream_t s, double* d_chi_mirror = nullptr,
                           bool trial_scale = false, unsigned long long* host_ll = nullptr, unsigned int seq = 0) {
    const GraphDev v = g->view();
    // One launch for both halves (k_linearize2): config 5 0.0325 -> 0.0249 ms, a million edges 0.124 -> 0.1005 ms per linearisation (a launch and
    // its gap less; the Hpl records leave through LDS as whole lines; the keyframe side's stores and the landmark side's LDS reductions overlap).
    // Before the stores were coalesced the merged launch LOST at a million edges (0.131 against 0.125 ms): the partial-line write requests of
    // the keyframe side were what both halves queued behind. OVS_BA_LIN_MERGED=0: k_lin_pose (two entries per thread) and k_lin_landmark as two
    // launches (0.110 ms at a million edges with the staged stores, 0.124 without: OVS_BA_HPL_STAGE=0). The two forms differ in the last bits
    // of Hpp / bp (another summation tree), not in anything per edge or per landmark.
    static const bool merged = [] {
        const char* e = std::getenv("OVS_BA_LIN_MERGED");
        return !(e && e[0] == '0');
    }();
    if (merged) {
        const unsigned n_wg = (unsigned)(2 * g->n_chunks + g->n_lm_wg);
        if (g->model == 1)
            hipLaunchKernelGGL(k_linearize2<1>, dim3(n_wg), dim3(256), 0, s, v, d_poses, d_points, huber_mono, huber_stereo, d_Hpl, d_Hll, d_bl, g->d_lm_tmp, d_chi3);
        else
            hipLaunchKernelGGL(k_linearize2<0>, dim3(n_wg), dim3(256), 0, s, v, d_poses, d_points, huber_mono, huber_stereo, d_Hpl, d_Hll, d_bl, g->d_lm_tmp, d_chi3);
        OVS_LAUNCH_TRY("k_linearize2");
    } else {
        if (g->n_chunks > 0) {
            static const bool stage = [] {   // OVS_BA_HPL_STAGE=0: every lane stores its own record (round 6's first form)
                const char* e = std::getenv("OVS_BA_HPL_STAGE");
                return !(e && e[0] == '0');
            }();
            if (stage) {
                if (g->model == 1) hipLaunchKernelGGL((k_lin_pose<1, true>), dim3(g->n_chunks), dim3(256), 0, s, v, d_poses, d_points, huber_mono, huber_stereo, d_Hpl);
                else hipLaunchKernelGGL((k_lin_pose<0, true>), dim3(g->n_chunks), dim3(256), 0, s, v, d_poses, d_points, huber_mono, huber_stereo, d_Hpl);
            } else if (g->model == 1) hipLaunchKernelGGL((k_lin_pose<1, false>), dim3(g->n_chunks), dim3(256), 0, s, v, d_poses, d_points, huber_mono, huber_stereo, d_Hpl);
            else hipLaunchKernelGGL((k_lin_pose<0, false>), dim3(g->n_chunks), dim3(256), 0, s, v, d_poses, d_points, huber_mono, huber_stereo, d_Hpl);
            OVS_LAUNCH_TRY("k_lin_pose");
        }
        if (g->model == 1)
            hipLaunchKernelGGL(k_lin_landmark<1>, dim3(g->n_lm_wg), dim3(256), 0, s, v, d_poses, d_points, huber_mono, huber_stereo, d_Hpp, d_bp, d_Hll, d_bl, g->d_lm_tmp);
        else
            hipLaunchKernelGGL(k_lin_landmark<0>, dim3(g->n_lm_wg), dim3(256), 0, s, v, d_poses, d_points, huber_mono, huber_stereo, d_Hpp, d_bp, d_Hll, d_bl, g->d_lm_tmp);
        OVS_LAUNCH_TRY("k_lin_landmark");
    }
    hipLaunchKernelGGL(k_reduce_scalars, dim3(1 + (merged ? (27 * g->n_pose + 1023) / 1024 : 0)), dim3(1024), 0, s, v, g->d_lm_tmp, d_Hpp, d_Hll, d_chi3, d_chi_mirror,
                       trial_scale ? g->d_lm_tmp + 3 * (size_t)g->n_pt : (const double*)nullptr, trial_scale ? g->d_scal : (double*)nullptr,
                       trial_scale ? host_ll : (unsigned long long*)nullptr, seq, (const int32_t*)g->d_fail, merged ? d_Hpp : (double*)nullptr,
                       merged ? d_bp : (double*)nullptr);
    OVS_LAUNCH_TRY("k_reduce_scalars");
    return OVS_OK;
}

}   // namespace

// The two device arenas of a graph (5.8 MB + 18 MB at config 5) are recycled: mapping_module builds one graph per new keyframe, and a hipMalloc /
// hipFree pair per arena cost ~0.3 ms of a 7.7 ms ovs_local_ba_optimize (hipFree synchronises the device). Per device, first fit among the few
// kept blocks that are not more than twice the request; the pool holds at most four blocks and frees the oldest beyond that.
struct BaArenaPool {
    struct Block {
        int device;
        size_t cap;
        unsigned char* p;
    };
    static constexpr size_t kKeepPerDevice = 4;
    std::mutex mu;
    std::vector<Block> free_;
    unsigned char* take(int device, size_t bytes, size_t* cap) {
        {
            std::lock_guard<std::mutex> lock(mu);
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].device == device && free_[i].cap >= bytes && free_[i].cap <= 2 * bytes + (1u << 20)) {
                    unsigned char* p = free_[i].p;
                    *cap = free_[i].cap;
                    free_.erase(free_.begin() + (long)i);
                    return p;
                }
        }
        const size_t want = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
        unsigned char* p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) return nullptr;
        *cap = want;
        return p;
    }
    // hipFree of a block under ITS device (the caller's current device is restored)
    static void release(const Block& b) {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != b.device) (void)hipSetDevice(b.device);
        (void)hipFree(b.p);
        if (cur >= 0 && cur != b.device) (void)hipSetDevice(cur);
    }
    void give(int device, unsigned char* p, size_t cap) {   // the caller guarantees that no work on the block is in flight
        if (!p) return;
        Block drop{-1, 0, nullptr};
        {
            std::lock_guard<std::mutex> lock(mu);
            // the cap is per device: one device's blocks never evict another's
            size_t mine = 0, oldest = free_.size();
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].device == device) {
                    if (oldest == free_.size()) oldest = i;
                    ++mine;
                }
            if (mine >= kKeepPerDevice) {
                drop = free_[oldest];
                free_.erase(free_.begin() + (long)oldest);
            }
            free_.push_back(Block{device, cap, p});
        }
        if (drop.p) release(drop);
    }
    void trim() {   // ovs_ba_pool_trim: a long-lived process gives the kept blocks back
        std::vector<Block> all;
        {
            std::lock_guard<std::mutex> lock(mu);
            all.swap(free_);
        }
        for (const Block& b : all) release(b);
    }
};
static BaArenaPool g_ba_pool;

extern "C" {

ovs_status ovs_ba_pool_trim(void) {
    g_ba_pool.trim();
    return OVS_OK;
}

ovs_status ovs_ba_graph_destroy(ovs_ba_graph* g) {
    if (!g) return OVS_OK;
    (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();   // the arenas go back to the pool: nothing of this graph may still be running (hipFree synchronised implicitly)
    g_ba_pool.give(g->device, g->d_arena, g->arena_cap);          // every array of the graph lives in one of the two arenas
    g_ba_pool.give(g->device, g->d_solver_arena, g->solver_cap);
    delete g;
    return OVS_OK;
}

static ovs_status graph_create(int model, int32_t device, int32_t n_pose, const uint8_t* pose_fixed, int32_t n_pt, const ovs_ba_edge* mono,
                               int32_t n_mono, const ovs_ba_edge_stereo* stereo, int32_t n_stereo, const ovs_ba_cam* cam, double focal_x_baseline,
                               ovs_ba_graph** out) {
    if (!out || !cam || n_pose < 1 || n_pt < 1 || n_mono < 0 || n_stereo < 0 || (n_mono > 0 && !mono) || (n_stereo > 0 && !stereo))
        return OVS_ERR_INVALID;
    *out = nullptr;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    OVS_HIP_TRY(hipSetDevice(device));
    const bool trace = ovs::tuning().ba_trace;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    ovs_ba_graph* g = new (std::nothrow) ovs_ba_graph();
    if (!g) return OVS_ERR_INVALID;
    g->device = device;
    g->n_pose = n_pose;
    g->n_pt = n_pt;
    g->n_mono = n_mono;
    g->n_stereo = n_stereo;
    g->cam = *cam;
    g->bf = focal_x_baseline;
    g->model = model;
    g->fixed.assign((size_t)n_pose, 0);
    if (pose_fixed) g->fixed.assign(pose_fixed, pose_fixed + n_pose);
    g->slot.assign((size_t)n_pose, -1);
    for (int k = 0; k < n_pose; ++k)
        if (!g->fixed[k]) {
            g->slot[k] = g->n_free++;
            g->slot_pose.push_back(k);
        }
    const int ne = n_mono + n_stereo;
    // The build's temporaries and the host image of the arena belong to the calling thread and keep their pages between calls (mapping_module
    // builds a graph per keyframe: 13 MB of fresh vectors per call were ~3000 page faults, most of the build's 0.6 ms on the host -- round 5).
    // The image is laid out first and filled in place: no per-array vector, no second copy.
    // Round 6: the image is page-locked, so the edge records (4.8 of the 6.0 MB at config 5) go up while the host is still sorting.
    struct Scratch {
        std::vector<int32_t> edge_pose, edge_pt, fl, fp, seen;
        unsigned char* image = nullptr;   // page-locked
        size_t image_cap = 0;
        ~Scratch() {
            if (image) (void)hipHostFree(image);
        }
    };
    static thread_local Scratch sc;
    size_t top = 0;
    auto place = [&top](size_t bytes) {   // 256-byte aligned offsets
        const size_t off = (top + 255) & ~(size_t)255;
        top = off + std::max<size_t>(bytes, 1);
        return off;
    };
    const int nf = g->n_free, n_pairs = nf * (nf + 1) / 2;
    const size_t o_edges = place(sizeof(GEdge) * (size_t)ne), o_lm_start = place(sizeof(int32_t) * ((size_t)n_pt + 1)),
                 o_lm_edges = place(sizeof(int32_t) * (size_t)ne), o_lm_nmono = place(sizeof(int32_t) * (size_t)n_pt),
                 o_pose_start = place(sizeof(int32_t) * ((size_t)n_pose + 1)), o_pose_edges = place(sizeof(int32_t) * (size_t)ne),
                 o_fixed = place((size_t)n_pose), o_active = place((size_t)ne), o_slot_of_pose = place(sizeof(int32_t) * (size_t)n_pose),
                 o_pose_pt = place(sizeof(int32_t) * (size_t)ne), o_pair_ab = place(sizeof(int32_t) * 2 * (size_t)n_pairs),
                 o_slot_pose = place(sizeof(int32_t) * (size_t)nf), o_wg_pair = place(sizeof(int32_t) * 8 * ((size_t)n_pairs / 8 + (size_t)nf + 1));
    // k_linearize's work partition (sizes are upper bounds: the tables are built below, from the counting sorts)
    const size_t max_chunks = (size_t)ne / kPoseChunk + (size_t)n_pose;
    const size_t o_lm_wg_first = place(sizeof(int32_t) * ((size_t)n_pt + 1)),
                 o_chunk_kf = place(sizeof(int32_t) * max_chunks), o_chunk_start = place(sizeof(int32_t) * ((size_t)n_pose + 1));
    const size_t upload_bytes = (top + 255) & ~(size_t)255;   // what the device reads before writing it ends here; scratch follows
    const size_t o_dup_host = place(sizeof(unsigned long long));   // (host image only: where the duplicate check's word comes down to)
    const size_t image_bytes = top;
    top = upload_bytes;
    const size_t o_lm_of_slot = place(sizeof(int32_t) * (size_t)ne), o_dup = place(sizeof(unsigned long long));
    const size_t o_lm_tmp = place(sizeof(double) * 4 * (size_t)n_pt);
    const size_t o_pose_part = place(sizeof(double) * 27 * 2 * max_chunks);   // (k_linearize2: one sum per HALF chunk)
    const size_t o_ledges = place(sizeof(GEdge) * (size_t)ne);
    const size_t arena_bytes = top;
#define G_TRY(expr)                            \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) {                \
            ovs::set_last_error(#expr, _e);    \
            ovs_ba_graph_destroy(g);           \
            return OVS_ERR_HIP;                \
        }                                      \
    } while (0)
    if (sc.image_cap < image_bytes) {
        if (sc.image) (void)hipHostFree(sc.image);
        sc.image = nullptr;
        sc.image_cap = 0;
        const size_t cap = image_bytes + image_bytes / 4;   // (head room: the next local map is a little larger more often than not)
        G_TRY(hipHostMalloc(reinterpret_cast<void**>(&sc.image), cap, hipHostMallocDefault));
        sc.image_cap = cap;
    }
    unsigned char* const img = sc.image;
    g->d_arena = g_ba_pool.take(device, arena_bytes, &g->arena_cap);
    G_TRY(g->d_arena ? hipSuccess : hipErrorOutOfMemory);
    GEdge* const edges = reinterpret_cast<GEdge*>(img + o_edges);
    int32_t* const lm_start = reinterpret_cast<int32_t*>(img + o_lm_start);
    int32_t* const lm_edges = reinterpret_cast<int32_t*>(img + o_lm_edges);
    int32_t* const lm_nmono = reinterpret_cast<int32_t*>(img + o_lm_nmono);
    int32_t* const pose_start = reinterpret_cast<int32_t*>(img + o_pose_start);
    int32_t* const pose_edges = reinterpret_cast<int32_t*>(img + o_pose_edges);
    int32_t* const pose_pt = reinterpret_cast<int32_t*>(img + o_pose_pt);
    if (sc.edge_pose.size() < (size_t)ne) {
        sc.edge_pose.resize((size_t)ne);
        sc.edge_pt.resize((size_t)ne);
    }
    int32_t* const edge_pose = sc.edge_pose.data();
    int32_t* const edge_pt = sc.edge_pt.data();
    // counting sorts: ascending edge index inside every landmark / keyframe (mono edges have the lower indices, so "mono first" is free).
    // Four passes over the edges in all (round 6: the histograms ride on the record fill, pose_pt on the scatter, the landmark of a slot on the
    // duplicate check -- eight passes were 0.69 ms of a 6 ms ovs_local_ba_optimize at config 5).
    std::memset(lm_start, 0, sizeof(int32_t) * ((size_t)n_pt + 1));
    std::memset(lm_nmono, 0, sizeof(int32_t) * (size_t)n_pt);
    std::memset(pose_start, 0, sizeof(int32_t) * ((size_t)n_pose + 1));
    bool bad_index = false;   // (round 6: the range check rides on the first pass instead of being a pass of its own)
    for (int i = 0; i < n_mono; ++i) {
        const int32_t kp = mono[i].pose_idx, pt = mono[i].point_idx;
        if ((uint32_t)kp >= (uint32_t)n_pose || (uint32_t)pt >= (uint32_t)n_pt) {
            bad_index = true;
            break;
        }
        edges[i] = GEdge{kp, pt, mono[i].obs_x, mono[i].obs_y, 0.0, mono[i].inv_sigma_sq};
        edge_pose[i] = kp;
        edge_pt[i] = pt;
        ++lm_start[(size_t)pt + 1];
        ++lm_nmono[pt];
        ++pose_start[(size_t)kp + 1];
    }
    for (int i = 0; i < n_stereo && !bad_index; ++i) {
        const int32_t kp = stereo[i].pose_idx, pt = stereo[i].point_idx;
        if ((uint32_t)kp >= (uint32_t)n_pose || (uint32_t)pt >= (uint32_t)n_pt) {
            bad_index = true;
            break;
        }
        edges[(size_t)n_mono + i] = GEdge{kp, pt, stereo[i].obs_x, stereo[i].obs_y, stereo[i].obs_x_right, stereo[i].inv_sigma_sq};
        edge_pose[(size_t)n_mono + i] = kp;
        edge_pt[(size_t)n_mono + i] = pt;
        ++lm_start[(size_t)pt + 1];
        ++pose_start[(size_t)kp + 1];
    }
    if (bad_index) {
        ovs::set_last_error_text("ovs_ba_graph_create: an edge's keyframe or landmark index is out of range");
        ovs_ba_graph_destroy(g);
        return OVS_ERR_INVALID;
    }
    const double t_p1 = now();
    // the records are final: they travel (null stream, page-locked source: the call returns at once) under the remaining passes
    const size_t early_bytes = std::min(upload_bytes, (o_edges + sizeof(GEdge) * (size_t)ne + 255) & ~(size_t)255);
    if (ne > 0) G_TRY(hipMemcpyAsync(g->d_arena, img, early_bytes, hipMemcpyHostToDevice, nullptr));
    for (int j = 0; j < n_pt; ++j) lm_start[(size_t)j + 1] += lm_start[j];
    for (int k = 0; k < n_pose; ++k) pose_start[(size_t)k + 1] += pose_start[k];
    sc.fl.assign(lm_start, lm_start + n_pt);
    sc.fp.assign(pose_start, pose_start + n_pose);
    for (int e = 0; e < ne; ++e) {
        const int32_t pt = edge_pt[e];
        lm_edges[(size_t)sc.fl[pt]++] = e;
        const int32_t at = sc.fp[edge_pose[e]]++;
        pose_edges[(size_t)at] = e;
        pose_pt[(size_t)at] = pt;   // the landmark of every entry of pose_edges: k_schur's pair blocks and k_lin_pose walk a keyframe's observations without the 48-byte records
    }
    const double t_p2 = now();
    // k_lin_landmark's work partition: runs of whole landmarks with at most kLmSlots edges (and 256 landmarks). (The landmark of every slot and
    // the check that a keyframe observes a landmark at most once -- upstream: landmark::add_observation ignores a second observation by the same
    // keyframe; the reduced system relies on it: k_edge_table keeps ONE edge per (keyframe, landmark), two edges of one free keyframe to one
    // landmark would need cross terms while Hpp / Hll / rhs would still count both, so such an edge list is refused -- moved to the device in
    // round 6: k_edges_by_slot, k_dup_check below.)
    {
        int32_t* const wg_first = reinterpret_cast<int32_t*>(img + o_lm_wg_first);
        int n_wg = 0, first = 0;
        size_t pl_bound = 0;
        for (int j = 0; j < n_pt; ++j) {
            const size_t nj = (size_t)(lm_start[(size_t)j + 1] - lm_start[j]);
            pl_bound += nj * (nj + 1) / 2;
            if (j > first && (lm_start[(size_t)j + 1] - lm_start[first] > kLmSlots || j - first >= 256)) {   // j does not fit: it opens the next run
                wg_first[n_wg++] = first;
                first = j;
            }
        }
        wg_first[n_wg++] = first;
        wg_first[n_wg] = n_pt;
        g->n_lm_wg = n_wg;
        g->pl_bound = pl_bound;
        // k_lin_pose: chunks of kPoseChunk entries of a keyframe's edge list
        int32_t* const chunk_kf = reinterpret_cast<int32_t*>(img + o_chunk_kf);
        int32_t* const chunk_start = reinterpret_cast<int32_t*>(img + o_chunk_start);
        int n_ch = 0;
        for (int k = 0; k < n_pose; ++k) {
            chunk_start[k] = n_ch;
            for (int i = pose_start[k]; i < pose_start[(size_t)k + 1]; i += kPoseChunk) chunk_kf[n_ch++] = k;
        }
        chunk_start[n_pose] = n_ch;
        g->n_chunks = n_ch;
    }
    const double t1 = now();
    std::memcpy(img + o_fixed, g->fixed.data(), (size_t)n_pose);
    std::memset(img + o_active, 1, (size_t)std::max(ne, 1));
    std::memcpy(img + o_slot_of_pose, g->slot.data(), sizeof(int32_t) * (size_t)n_pose);   // keyframe -> block of the reduced system or -1
    // reduced system: the blocks (a, b), a <= b in slot order, one workgroup each (which landmarks two keyframes share is found on the device)
    if (nf > 0) {
        int32_t* const pab = reinterpret_cast<int32_t*>(img + o_pair_ab);
        // a-major: a row (a, a .. nf - 1) stays together (k_schur_l hands contiguous runs of pairs to one XCD) and starts with its longest list,
        // the diagonal pair's (all of a's landmarks). "All diagonal pairs first" was measured: no change.
        int p = 0;
        for (int a = 0; a < nf; ++a)
            for (int b = a; b < nf; ++b) {
                pab[(size_t)2 * p] = a;
                pab[(size_t)2 * p + 1] = b;
                ++p;
            }
        g->n_pairs = n_pairs;
        // k_schur_l's work order. Rows are dealt to the eight XCDs in serpentine order (rows 0 .. 7 to XCDs 0 .. 7, rows 8 .. 15 to XCDs 7 .. 0,
        // ...: row a has nf - a pairs, so the eight sums come out within a row's length of each other); XCD x's i-th pair is workgroup 8 i + x.
        {
            int32_t* const wp = reinterpret_cast<int32_t*>(img + o_wg_pair);
            int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int cap = n_pairs / 8 + nf + 1;
            for (int i = 0; i < 8 * cap; ++i) wp[i] = -1;
            int first = 0;   // index of pair (a, a)
            for (int a = 0; a < nf; ++a) {
                const int x = ((a >> 3) & 1) ? 7 - (a & 7) : (a & 7);
                for (int b = a; b < nf; ++b) wp[(size_t)8 * fill[x]++ + x] = first + (b - a);
                first += nf - a;
            }
            g->n_pair_wg = 8 * *std::max_element(fill, fill + 8);
        }
        std::memcpy(img + o_slot_pose, g->slot_pose.data(), sizeof(int32_t) * (size_t)nf);
    }
    const double t2 = now();
    {
        const size_t from = ne > 0 ? early_bytes : 0;
        if (upload_bytes > from) G_TRY(hipMemcpyAsync(g->d_arena + from, img + from, upload_bytes - from, hipMemcpyHostToDevice, nullptr));
    }
    unsigned char* A = g->d_arena;
    g->d_edges = reinterpret_cast<GEdge*>(A + o_edges);
    g->d_lm_start = reinterpret_cast<int32_t*>(A + o_lm_start);
    g->d_lm_edges = reinterpret_cast<int32_t*>(A + o_lm_edges);
    g->d_lm_nmono = reinterpret_cast<int32_t*>(A + o_lm_nmono);
    g->d_pose_start = reinterpret_cast<int32_t*>(A + o_pose_start);
    g->d_pose_edges = reinterpret_cast<int32_t*>(A + o_pose_edges);
    g->d_fixed = A + o_fixed;
    g->d_active = A + o_active;
    g->d_lm_tmp = reinterpret_cast<double*>(A + o_lm_tmp);
    g->d_lm_of_slot = reinterpret_cast<int32_t*>(A + o_lm_of_slot);
    g->d_lm_wg_first = reinterpret_cast<int32_t*>(A + o_lm_wg_first);
    g->d_chunk_kf = reinterpret_cast<int32_t*>(A + o_chunk_kf);
    g->d_chunk_start = reinterpret_cast<int32_t*>(A + o_chunk_start);
    g->d_pose_part = reinterpret_cast<double*>(A + o_pose_part);
    g->d_ledges = reinterpret_cast<GEdge*>(A + o_ledges);
    if (ne > 0) {   // null stream: ordered behind the upload above and before whatever stream the caller linearises on (the wait costs ~10 us)
        unsigned long long* const d_dup = reinterpret_cast<unsigned long long*>(A + o_dup);
        hipLaunchKernelGGL(k_edges_by_slot, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, nullptr, g->d_edges, g->d_lm_edges, ne, g->d_ledges,
                           g->d_lm_of_slot, d_dup);
        G_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_dup_check, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, nullptr, g->d_ledges, g->d_lm_start, ne, d_dup);
        G_TRY(hipGetLastError());
        G_TRY(hipMemcpyAsync(img + o_dup_host, d_dup, sizeof(unsigned long long), hipMemcpyDeviceToHost, nullptr));
    }
    G_TRY(hipStreamSynchronize(nullptr));   // the image is this thread's next graph's as well: nothing of it may still be on its way
    if (ne > 0) {
        unsigned long long dup;
        std::memcpy(&dup, img + o_dup_host, sizeof(dup));
        if (dup != ~0ull) {
            ovs::set_last_error_text("ovs_ba_graph_create: keyframe " + std::to_string((uint32_t)dup) + " has two edges to landmark " +
                                     std::to_string((uint32_t)(dup >> 32)));
            ovs_ba_graph_destroy(g);
            return OVS_ERR_INVALID;
        }
    }
    g->d_slot_of_pose = reinterpret_cast<int32_t*>(A + o_slot_of_pose);
    g->d_pose_pt = reinterpret_cast<int32_t*>(A + o_pose_pt);
    if (g->n_free > 0) {
        g->d_pair_ab = reinterpret_cast<int32_t*>(A + o_pair_ab);
        g->d_wg_pair = reinterpret_cast<int32_t*>(A + o_wg_pair);
        g->d_slot_pose = reinterpret_cast<int32_t*>(A + o_slot_pose);
    }
#undef G_TRY
    if (trace)
        std::fprintf(stderr, "[ovs_ba_graph_create] %.2f ms: edge records + histograms %.2f, scatter %.2f, landmark-order pass %.2f, other arrays %.2f, rest of "
                             "the %.1f MB upload + k_edges_by_slot %.2f\n",
                     now() - t0, t_p1 - t0, t_p2 - t_p1, t1 - t_p2, t2 - t1, upload_bytes / 1e6, now() - t2);
    *out = g;
    return OVS_OK;
}

ovs_status ovs_ba_graph_create(int32_t device, int32_t n_pose, const uint8_t* pose_fixed, int32_t n_pt, const ovs_ba_edge* mono, int32_t n_mono,
                               const ovs_ba_edge_stereo* stereo, int32_t n_stereo, const ovs_ba_cam* cam, double focal_x_baseline,
                               ovs_ba_graph** out) {
    return graph_create(0, device, n_pose, pose_fixed, n_pt, mono, n_mono, stereo, n_stereo, cam, focal_x_baseline, out);
}

ovs_status ovs_ba_graph_create_equirect(int32_t device, int32_t n_pose, const uint8_t* pose_fixed, int32_t n_pt, const ovs_ba_edge* mono,
                                        int32_t n_mono, int32_t cols, int32_t rows, ovs_ba_graph** out) {
    if (cols < 1 || rows < 1) return OVS_ERR_INVALID;
    const ovs_ba_cam cam = {(double)cols, (double)rows, 0.0, 0.0};
    return graph_create(1, device, n_pose, pose_fixed, n_pt, mono, n_mono, nullptr, 0, &cam, 0.0, out);
}

ovs_status ovs_ba_graph_linearize_dev(ovs_ba_graph* g, const double* d_poses, const double* d_points, double huber_mono, double huber_stereo,
                                      double* d_Hpp, double* d_bp, double* d_Hll, double* d_bl, double* d_Hpl, double* d_chi2, void* stream) {
    if (!g || !d_poses || !d_points || !d_Hpp || !d_bp || !d_Hll || !d_bl || !d_Hpl || !d_chi2) return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(g->device));
    return graph_linearize(g, d_poses, d_points, huber_mono, huber_stereo, d_Hpp, d_bp, d_Hll, d_bl, d_Hpl, d_chi2, (hipStream_t)stream);
}

}   // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// Levenberg-Marquardt on top of the graph (used by ovs_local_ba_optimize in ba_optimize.hip)
// ---------------------------------------------------------------------------------------------------------------------------
namespace ovs {

// The reduced camera system is stored the way the device solver wants it (ba_solve.hip): pitch n_pad = 6 n_free rounded up to 16, an identity
// block on the padding, the right-hand side as row n_pad, zero rows behind it. The padding survives a solve, so it is written once here.
static ovs_status solver_workspace_create(ovs_ba_graph* g, hipStream_t s);
ovs_status ba_graph_reset_system(ovs_ba_graph* g, hipStream_t s);

ovs_status ba_graph_ensure_solver(ovs_ba_graph* g, hipStream_t s) {
    if (g->d_Hinv) return OVS_OK;
    const ovs_status st = solver_workspace_create(g, s);
    if (st != OVS_OK) {   // d_Hinv doubles as the "work space is ready" mark: a half-built one must not pass for ready on the next call
        (void)hipStreamSynchronize(s);
        g_ba_pool.give(g->device, g->d_solver_arena, g->solver_cap);
        g->d_solver_arena = nullptr;
        g->d_Hinv = nullptr;
    }
    return st;
}

static ovs_status solver_workspace_create(ovs_ba_graph* g, hipStream_t s) {
    const size_t ne = std::max<size_t>((size_t)g->n_edge(), 1);
    const int n = 6 * std::max(g->n_free, 1), n_pad = dense_solve_pad(n);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t sys = dense_solve_doubles(n);
    const size_t b_hinv = al(sizeof(double) * 9 * (size_t)g->n_pt), b_y = al(sizeof(double) * 18 * ne),
                 b_s = al(sizeof(double) * (sys + 6 * (size_t)g->n_pose)), b_dxp = al(sizeof(double) * 6 * (size_t)g->n_pose),
                 b_tab = al(sizeof(int32_t) * (size_t)std::max(g->n_free, 1) * (size_t)g->n_pt);
    const size_t n_lists = (size_t)4 * (size_t)std::max(g->n_pairs, 0);
    // the lists' offsets are 32-bit and their storage is sized by the bound: a map whose bound is beyond 2^28 entries (2 GB; e.g. hundreds of
    // keyframes that all observe the same landmarks) keeps the per-trial scan (k_schur) instead
    const bool pl_ok = g->pl_bound <= ((size_t)1 << 28);
    const size_t b_plc = al(sizeof(int32_t) * (n_lists + 1)), b_ple = al(sizeof(int2) * (pl_ok ? std::max<size_t>(g->pl_bound, 1) : 1));
    g->d_solver_arena = g_ba_pool.take(g->device, b_hinv + b_y + b_s + b_dxp + 512 + b_tab + 2 * b_plc + b_ple, &g->solver_cap);
    OVS_HIP_TRY(g->d_solver_arena ? hipSuccess : hipErrorOutOfMemory);
    unsigned char* A = g->d_solver_arena;
    g->d_Hinv = reinterpret_cast<double*>(A);
    g->d_Y = reinterpret_cast<double*>(A + b_hinv);
    g->d_S = reinterpret_cast<double*>(A + b_hinv + b_y);   // padded system
    g->d_dxp = reinterpret_cast<double*>(A + b_hinv + b_y + b_s);
    g->d_scal = reinterpret_cast<double*>(A + b_hinv + b_y + b_s + b_dxp);
    g->d_fail = reinterpret_cast<int32_t*>(A + b_hinv + b_y + b_s + b_dxp + 256);
    g->d_edge_of = reinterpret_cast<int32_t*>(A + b_hinv + b_y + b_s + b_dxp + 512);
    OVS_HIP_TRY(hipMemsetAsync(g->d_fail, 0, 2 * sizeof(int32_t), s));   // both failure words (ba_graph_schur)
    OVS_HIP_TRY(hipMemsetAsync(g->d_edge_of, 0xff, b_tab, s));   // -1
    if (g->n_edge() > 0 && g->n_free > 0) {
        hipLaunchKernelGGL(k_edge_table, dim3((g->n_edge() + 255) / 256), dim3(256), 0, s, g->d_edges, g->n_edge(), g->d_slot_of_pose, g->n_pt, g->d_edge_of);
        OVS_LAUNCH_TRY("k_edge_table");
    }
    g->d_pl_cnt = reinterpret_cast<int32_t*>(A + b_hinv + b_y + b_s + b_dxp + 512 + b_tab);
    g->d_pl_off = reinterpret_cast<int32_t*>(A + b_hinv + b_y + b_s + b_dxp + 512 + b_tab + b_plc);
    g->d_pl_ent = reinterpret_cast<int2*>(A + b_hinv + b_y + b_s + b_dxp + 512 + b_tab + 2 * b_plc);
    g->pl_ready = false;
    if (pl_ok && g->n_edge() > 0 && g->n_free > 0 && g->n_pairs > 0) {   // behind k_edge_table on the same stream
        g->pl_ready = true;
        hipLaunchKernelGGL(k_pair_lists<false>, dim3(g->n_pairs), dim3(256), 0, s, g->d_pose_start, g->d_pose_edges, g->d_pose_pt, g->d_pair_ab, g->d_slot_pose,
                           g->d_edge_of, g->n_pt, g->d_pl_cnt, (const int32_t*)nullptr, (int2*)nullptr);
        OVS_LAUNCH_TRY("k_pair_lists<count>");
        hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, s, g->d_pl_cnt, (int)n_lists, g->d_pl_off);
        OVS_LAUNCH_TRY("k_scan_i32");
        hipLaunchKernelGGL(k_pair_lists<true>, dim3(g->n_pairs), dim3(256), 0, s, g->d_pose_start, g->d_pose_edges, g->d_pose_pt, g->d_pair_ab, g->d_slot_pose,
                           g->d_edge_of, g->n_pt, (int32_t*)nullptr, g->d_pl_off, g->d_pl_ent);
        OVS_LAUNCH_TRY("k_pair_lists<fill>");
    }
    g->s_pitch = n_pad;
    g->d_rhs = g->d_S + (size_t)n_pad * n_pad;
    return ba_graph_reset_system(g, s);
}

// The padded system's constant part: everything zero, an identity block on rows / columns n .. n_pad - 1 (ba_solve.hip: "the padding reproduces
// itself"). Written when the workspace is created -- and again after a FAILED solve: a non-finite entry of S reaches the padding through the
// trailing update (0 * NaN) before any pivot is tested, and the schur kernels rewrite the n x n part and the right-hand side only, so without the
// repair every later trial on this graph would fail as well, whatever lambda grows to.
ovs_status ba_graph_reset_system(ovs_ba_graph* g, hipStream_t s) {
    if (!g->d_S) return OVS_OK;
    const int n = 6 * std::max(g->n_free, 1), n_pad = g->s_pitch;
    OVS_HIP_TRY(hipMemsetAsync(g->d_S, 0, sizeof(double) * dense_solve_doubles(n), s));
    static const double ones[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};   // (n_pad - n < 16; static: the copy needs no wait here)
    if (n_pad > n)
        OVS_HIP_TRY(hipMemcpy2DAsync(g->d_S + (size_t)n * n_pad + n, sizeof(double) * ((size_t)n_pad + 1), ones, sizeof(double), sizeof(double),
                                     (size_t)(n_pad - n), hipMemcpyHostToDevice, s));
    return OVS_OK;
}

// (H + lambda I) dx = b, landmarks eliminated on the device: S (pitch g->s_pitch) and rhs at g->d_S.
// `fail_word`: 0 / 1 -- which of the two failure words (d_fail[0..1]) this trial reports in; `clear_first`: zero it by a memset here (the host-solver
// path; the device-solver path has k_trial_update clear the other word for the next trial, and both words at the start of a round)
ovs_status ba_graph_schur(ovs_ba_graph* g, const double* d_Hpp, const double* d_bp, const double* d_Hll, const double* d_bl, const double* d_Hpl,
                          double lambda, hipStream_t s, int fail_word, bool clear_first) {
    const GraphDev v = g->view();
    if (clear_first) OVS_HIP_TRY(hipMemsetAsync(g->d_fail + fail_word, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_lm_prepare, dim3((std::max(g->n_pt, g->n_edge()) + 255) / 256), dim3(256), 0, s, v, d_Hll, d_Hpl, lambda, g->d_Hinv, g->d_Y,
                       g->d_fail + fail_word);
    OVS_LAUNCH_TRY("k_lm_prepare");
    if (g->n_free > 0) {
        static const bool lists = [] {   // OVS_BA_SCHUR_LISTS=0: every trial's launch scans for the pairs' common landmarks itself (rounds 4-6)
            const char* e = std::getenv("OVS_BA_SCHUR_LISTS");
            return !(e && e[0] == '0');
        }();
        static const bool coop = [] {   // OVS_BA_SCHUR_COOP=0: every lane gathers its own two records
            const char* e = std::getenv("OVS_BA_SCHUR_COOP");
            return !(e && e[0] == '0');
        }();
        static const bool xcd = [] {   // OVS_BA_SCHUR_XCD=0: pairs in launch order (a-major), whatever XCD a workgroup lands on
            const char* e = std::getenv("OVS_BA_SCHUR_XCD");
            return !(e && e[0] == '0');
        }();
        const int32_t* const wgp = xcd ? g->d_wg_pair : nullptr;
        const int n_pwg = xcd ? g->n_pair_wg : g->n_pairs;
        if (lists && g->pl_ready && coop)
            hipLaunchKernelGGL(k_schur_l<true>, dim3(g->n_free + n_pwg), dim3(256), 0, s, v, g->n_free, g->d_pair_ab, g->d_slot_pose, g->d_pl_off, g->d_pl_ent,
                               d_Hpp, d_bp, d_bl, d_Hpl, g->d_Y, lambda, g->s_pitch, g->d_S, g->d_rhs, wgp, n_pwg);
        else if (lists && g->pl_ready)
            hipLaunchKernelGGL(k_schur_l<false>, dim3(g->n_free + n_pwg), dim3(256), 0, s, v, g->n_free, g->d_pair_ab, g->d_slot_pose, g->d_pl_off, g->d_pl_ent,
                               d_Hpp, d_bp, d_bl, d_Hpl, g->d_Y, lambda, g->s_pitch, g->d_S, g->d_rhs, wgp, n_pwg);
        else
            hipLaunchKernelGGL(k_schur, dim3(g->n_free + g->n_pairs), dim3(256), 0, s, v, g->n_free, g->d_pose_pt, g->d_pair_ab, g->d_slot_pose,
                               g->d_edge_of, d_Hpp, d_bp, d_bl, d_Hpl, g->d_Y, lambda, g->s_pitch, g->d_S, g->d_rhs);
        OVS_LAUNCH_TRY("k_schur");
    }
    return OVS_OK;
}

// host-solver path: the keyframes' increments were uploaded to d_dxp (six per keyframe). The landmarks' gain-ratio terms stay in d_lm_tmp[3 n_pt ..):
// the linearisation of the trial state (ba_graph_linearize with trial_scale) sums them into d_scal[0].
ovs_status ba_graph_backsub(ovs_ba_graph* g, const double* d_Hpl, const double* d_bl, double lambda, const double* d_X, double* d_Xn, hipStream_t s) {
    const GraphDev v = g->view();
    hipLaunchKernelGGL(k_backsub, dim3((g->n_pt + 127) / 128), dim3(128), 0, s, v, g->d_Hinv, d_Hpl, d_bl, g->d_dxp, lambda, d_X, d_Xn,
                       g->d_lm_tmp + 3 * (size_t)g->n_pt);
    OVS_LAUNCH_TRY("k_backsub");
    return OVS_OK;
}

// device-solver path: keyframes (T -> Tn, the 7-double records, dxp, the keyframes' gain-ratio terms in d_scal[1]) and landmarks (X -> Xn, their
// terms in d_lm_tmp[3 n_pt ..)) of the trial state from the solution the dense solver left in d_rhs
ovs_status ba_graph_trial_update(ovs_ba_graph* g, const double* d_T, const double* d_bp, const double* d_Hpl, const double* d_bl, double lambda,
                                 double* d_Tn, double* d_p7n, const double* d_X, double* d_Xn, hipStream_t s, int next_fail_word) {
    const GraphDev v = g->view();
    const int lm_per_wg = landmarks_per_workgroup(g->n_pose, g->n_pt);
    if (tuning().ba_backsub_edges)
        hipLaunchKernelGGL(k_trial_update<true>, dim3(1 + g->n_lm_wg), dim3(256), 0, s, v, d_T, g->d_slot_of_pose, g->d_rhs, d_bp, lambda, d_Tn, d_p7n, g->d_dxp,
                           g->d_scal + 1, g->d_Hinv, d_Hpl, d_bl, d_X, d_Xn, g->d_lm_tmp + 3 * (size_t)g->n_pt, g->d_fail + next_fail_word, lm_per_wg);
    else
        hipLaunchKernelGGL(k_trial_update<false>, dim3(1 + (g->n_pt + lm_per_wg - 1) / lm_per_wg), dim3(256), 0, s, v, d_T, g->d_slot_of_pose, g->d_rhs, d_bp, lambda, d_Tn,
                           d_p7n, g->d_dxp, g->d_scal + 1, g->d_Hinv, d_Hpl, d_bl, d_X, d_Xn, g->d_lm_tmp + 3 * (size_t)g->n_pt, g->d_fail + next_fail_word,
                           lm_per_wg);
    OVS_LAUNCH_TRY("k_trial_update");
    return OVS_OK;
}

// level-1 edges (round-1 outliers) are masked instead of rebuilding the graph: an inactive edge contributes exact zeros
ovs_status ba_graph_set_active(ovs_ba_graph* g, const uint8_t* host_mask, hipStream_t s) {
    if (g->n_edge() == 0) return OVS_OK;
    OVS_HIP_TRY(hipMemcpyAsync(g->d_active, host_mask, (size_t)g->n_edge(), hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    return OVS_OK;
}

ovs_status ba_graph_edge_chi2(ovs_ba_graph* g, const double* d_poses, const double* d_points, double* d_chi, uint8_t* d_depth, hipStream_t s) {
    if (g->n_edge() == 0) return OVS_OK;
    hipLaunchKernelGGL(k_edge_chi2, dim3((g->n_edge() + 255) / 256), dim3(256), 0, s, g->view(), d_poses, d_points, d_chi, d_depth);
    OVS_LAUNCH_TRY("k_edge_chi2");
    return OVS_OK;
}

// k_edge_gate; `write_active`: the verdicts also become the graph's active mask (what ba_graph_set_active uploads on the host-gate path)
ovs_status ba_graph_edge_gate(ovs_ba_graph* g, double thr_mono, double thr_stereo, const double* d_chi, const uint8_t* d_depth, const double* d_chi_r1,
                              const uint8_t* d_out1, bool use_final, uint8_t* d_out, bool write_active, int32_t* d_n_active, hipStream_t s) {
    if (g->n_edge() == 0) return OVS_OK;
    hipLaunchKernelGGL(k_edge_gate, dim3((g->n_edge() + 255) / 256), dim3(256), 0, s, g->n_edge(), g->n_mono, thr_mono, thr_stereo, d_chi, d_depth, d_chi_r1,
                       d_out1, use_final ? 1 : 0, d_out, write_active ? g->d_active : nullptr, d_n_active);
    OVS_LAUNCH_TRY("k_edge_gate");
    return OVS_OK;
}

ovs_status ba_graph_linearize(ovs_ba_graph* g, const double* d_poses, const double* d_points, double huber_mono, double huber_stereo, double* d_Hpp,
                              double* d_bp, double* d_Hll, double* d_bl, double* d_Hpl, double* d_chi3, hipStream_t s, double* d_chi_mirror,
                              bool trial_scale, unsigned long long* host_ll, unsigned int seq) {
    return graph_linearize(g, d_poses, d_points, huber_mono, huber_stereo, d_Hpp, d_bp, d_Hll, d_bl, d_Hpl, d_chi3, s, d_chi_mirror, trial_scale, host_ll,
                           seq);
}

}   // namespace ovs

namespace ovs {
struct BaGraphInfo {
    int n_free;
    const int32_t* slot;
    double *d_S, *d_dxp, *d_scal;
    int32_t* d_fail;
    const int32_t* d_slot_of_pose;
    int s_pitch;
    double* d_rhs;
};
BaGraphInfo ba_graph_info(ovs_ba_graph* g) {
    return BaGraphInfo{g->n_free, g->slot.data(), g->d_S, g->d_dxp, g->d_scal, g->d_fail, g->d_slot_of_pose, g->s_pitch, g->d_rhs};
}
}   // namespace ovs
