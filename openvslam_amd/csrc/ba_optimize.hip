// ba_optimize.hip -- B4: optimize::local_bundle_adjuster::optimize, the part behind the graph build (expected:
// src/openvslam/optimize/local_bundle_adjuster.cc; g2o OptimizationAlgorithmLevenberg + BlockSolver_6_3 with Schur complement).
//
// The caller (the class shim) flattens the local map into poses / landmarks / observation edges; this file runs what
// optimizer.optimize(num_first_iter) -> outlier levels -> optimizer.optimize(num_second_iter) does, on top of ba_graph.hip:
//   * the state (poses as SE3Quat records, points) and both block sets (current system / trial system) live in HBM for the whole call;
//   * one Levenberg-Marquardt trial = six launches on the device (round 6: k_lm_prepare, k_schur_l, the dense solver of ba_solve.hip,
//     k_trial_update, k_linearize2, k_reduce_scalars: ~230 us at config 5) whose outcome the host polls from twelve flag-carrying words in
//     page-locked memory (OVS_BA_LL_NOTIFY=0: ONE 260-byte download); with ovs_local_ba_set_solver(1) (BASELINE's north star keeps the
//     Cholesky of the reduced camera system on the HOST; at most 6 n_pose square) a 0.7 MB download (S | rhs | bp), 2.4 KB of pose increments
//     up, k_backsub, the SE3 update of <= 50 poses on the host, and the linearisation of the trial state;
//   * g2o's damping schedule (ORACLE_SPEC rule 25), the chi-square outlier gates between the two rounds and the final outlier flags.
// Round 1: 16 MB per trial crossed PCIe and the landmark elimination ran on 8 host threads (117 ms at config 5).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ba_host_math.h"
#include "ovs_common.h"

struct ovs_ba_graph;

namespace ovs {
// ba_graph.hip
ovs_status ba_graph_ensure_solver(ovs_ba_graph* g, hipStream_t s);
ovs_status ba_graph_schur(ovs_ba_graph* g, const double* d_Hpp, const double* d_bp, const double* d_Hll, const double* d_bl, const double* d_Hpl,
                          double lambda, hipStream_t s, int fail_word, bool clear_first);
ovs_status ba_graph_backsub(ovs_ba_graph* g, const double* d_Hpl, const double* d_bl, double lambda, const double* d_X, double* d_Xn, hipStream_t s);
ovs_status ba_graph_set_active(ovs_ba_graph* g, const uint8_t* host_mask, hipStream_t s);
ovs_status ba_graph_edge_chi2(ovs_ba_graph* g, const double* d_poses, const double* d_points, double* d_chi, uint8_t* d_depth, hipStream_t s);
ovs_status ba_graph_edge_gate(ovs_ba_graph* g, double thr_mono, double thr_stereo, const double* d_chi, const uint8_t* d_depth, const double* d_chi_r1,
                              const uint8_t* d_out1, bool use_final, uint8_t* d_out, bool write_active, int32_t* d_n_active, hipStream_t s);
ovs_status ba_graph_linearize(ovs_ba_graph* g, const double* d_poses, const double* d_points, double huber_mono, double huber_stereo, double* d_Hpp,
                              double* d_bp, double* d_Hll, double* d_bl, double* d_Hpl, double* d_chi3, hipStream_t s, double* d_chi_mirror = nullptr,
                              bool trial_scale = false, unsigned long long* host_ll = nullptr, unsigned int seq = 0);
ovs_status ba_graph_trial_update(ovs_ba_graph* g, const double* d_T, const double* d_bp, const double* d_Hpl, const double* d_bl, double lambda,
                                 double* d_Tn, double* d_p7n, const double* d_X, double* d_Xn, hipStream_t s, int next_fail_word);
struct BaGraphInfo {
    int n_free;
    const int32_t* slot;        // pose -> reduced block or -1
    double *d_S, *d_dxp, *d_scal;   // S | rhs | bp copy;  6 per keyframe;  [0] landmarks' / [1] keyframes' part of the gain ratio's denominator
    int32_t* d_fail;
    const int32_t* d_slot_of_pose;
    int s_pitch;                 // doubles per row of d_S (6 n_free rounded up to 16; ba_solve.hip's padded layout)
    double* d_rhs;               // row s_pitch of the system
};
BaGraphInfo ba_graph_info(ovs_ba_graph* g);
// ba_solve.hip
int dense_solve_max_n();
int dense_solve_pad(int n);
size_t dense_solve_doubles(int n);
ovs_status launch_dense_solve(double* d_S, int n, int32_t* d_fail, hipStream_t s, unsigned long long* d_tstats = nullptr);
ovs_status ba_graph_reset_system(ovs_ba_graph* g, hipStream_t s);
// where the reduced camera system is solved: 0 = on the device (k_chol_solve), 1 = on the host (ba_host_math.h cholesky_solve)
std::atomic<int> g_lba_solver{0};
}   // namespace ovs

namespace {

using namespace ovs_ba_host;
using ovs::set_last_error;

// upstream: constexpr float chi_sq_2D = 5.99146, chi_sq_3D = 7.81473 and their float square roots, widened to double where g2o consumes them
constexpr double kChi2D = 0x1.7f7414p+2, kChi3D = 0x1.f4248ap+2, kSqrtChi2D = 0x1.394fbcp+1, kSqrtChi3D = 0x1.65d26ap+1;

struct DevBlocks {   // one linearisation in HBM: Hpp | bp | Hll | bl | Hpl | chi2[2], max|diag|
    double *Hpp = nullptr, *bp = nullptr, *Hll = nullptr, *bl = nullptr, *Hpl = nullptr, *chi = nullptr;
    static size_t doubles(int n_pose, int n_pt, size_t n_edge) { return (size_t)42 * n_pose + (size_t)12 * n_pt + 18 * std::max<size_t>(n_edge, 1) + 4; }
    void carve(double* base, int n_pose, int n_pt, size_t n_edge) {
        Hpp = base;
        bp = Hpp + (size_t)36 * n_pose;
        Hll = bp + (size_t)6 * n_pose;
        bl = Hll + (size_t)9 * n_pt;
        Hpl = bl + (size_t)3 * n_pt;
        chi = Hpl + 18 * std::max<size_t>(n_edge, 1);
    }
};

// Per-thread work space that only grows: local BA runs once per keyframe on the mapping thread, and a dozen hipMalloc / hipHostMalloc /
// stream-create calls per call (3-5 ms) would cost as much as the optimisation itself.
struct LmScratch {
    int device = -1;
    hipStream_t stream = nullptr;
    unsigned char* d = nullptr;
    size_t d_cap = 0;
    double* h_pin = nullptr;
    size_t pin_cap = 0;
    unsigned char* h_edge = nullptr;   // pinned: two slots of [chi2 per edge (f64) | depth flag per edge (u8)], the results of edge_chi2 (round 1, final)
    size_t edge_cap = 0;               // edges per slot
    double* h_pts = nullptr;           // pinned: the landmarks on their way up (start of the call) and down (its end)
    size_t pts_cap = 0;                // doubles
    ~LmScratch() { release(); }
    void release() {
        if (d) (void)hipFree(d);
        if (h_pin) (void)hipHostFree(h_pin);
        if (h_edge) (void)hipHostFree(h_edge);
        if (h_pts) (void)hipHostFree(h_pts);
        h_pts = nullptr;
        pts_cap = 0;
        h_edge = nullptr;
        edge_cap = 0;
        if (stream) (void)hipStreamDestroy(stream);
        d = nullptr;
        h_pin = nullptr;
        stream = nullptr;
        d_cap = pin_cap = 0;
    }
};

struct Lm {
    int n_pose = 0, n_pt = 0, setup_type = 0;
    // OVS_BA_TRACE=1: wall-clock breakdown on stderr (where a call's milliseconds go; tools/time_lba.py)
    double t_schur = 0, t_chol = 0, t_trial = 0;
    int n_trials = 0;
    bool err_at_trial = false;   // the active edges' errors were last computed at the trial state (d_poses_n, d_Xn), which was then rejected
    static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    hipStream_t stream = nullptr;
    DevBlocks cur, trial, work;   // accepted state | last evaluated trial | where the next trial is evaluated (device solver only)
    double *d_poses = nullptr, *d_poses_n = nullptr, *d_poses_w = nullptr, *d_X = nullptr, *d_Xn = nullptr, *d_Xw = nullptr, *d_echi = nullptr;
    double *d_T = nullptr, *d_Tn = nullptr, *d_Tw = nullptr;   // R | t per keyframe (12 doubles), the state k_pose_update advances
    uint8_t* d_edepth = nullptr;
    // the outlier gates on the device (round 6): round 1's chi2 per edge (kept for the final verdict of level-1 edges), a second chi2 array for the
    // launch whose chi2 nobody reads, round 1's and the final flags, the inlier count
    double *d_echi_r1 = nullptr, *d_echi_s = nullptr;
    uint8_t *d_out1 = nullptr, *d_outf = nullptr;
    int32_t* d_nact = nullptr;
    double* h_pts = nullptr;   // LmScratch::h_pts
    double* h_pin = nullptr;   // pinned: S | rhs | bp | staging | chi3 | scal | fail
    size_t pin_doubles = 0, stage_off = 0;
    unsigned char* h_edge = nullptr;   // LmScratch::h_edge
    size_t edge_cap = 0;

    ovs_status init(int device, int np, int npt, size_t ne_max, const double* points) {
        static thread_local LmScratch sc;
        n_pose = np;
        n_pt = npt;
        if (sc.device != device) {
            sc.release();
            sc.device = device;
        }
        if (!sc.stream) OVS_HIP_TRY(hipStreamCreateWithFlags(&sc.stream, hipStreamNonBlocking));
        stream = sc.stream;
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t nb = al(sizeof(double) * DevBlocks::doubles(np, npt, ne_max)), b_p = al(sizeof(double) * 7 * np), b_x = al(sizeof(double) * 3 * npt),
                     b_e = al(sizeof(double) * std::max<size_t>(ne_max, 1)), b_d = al(std::max<size_t>(ne_max, 1));
        const size_t b_t = al(sizeof(double) * 12 * np);
        const size_t need = 3 * nb + 3 * b_p + 3 * b_x + 3 * b_t + 3 * b_e + 3 * b_d + 256;
        if (sc.d_cap < need) {
            if (sc.d) (void)hipFree(sc.d);
            sc.d = nullptr;
            sc.d_cap = 0;
            OVS_HIP_TRY(hipMalloc(&sc.d, need));
            sc.d_cap = need;
        }
        unsigned char* A = sc.d;
        cur.carve(reinterpret_cast<double*>(A), np, npt, ne_max);
        trial.carve(reinterpret_cast<double*>(A + nb), np, npt, ne_max);
        work.carve(reinterpret_cast<double*>(A + 2 * nb), np, npt, ne_max);
        A += 3 * nb;
        d_poses = reinterpret_cast<double*>(A);
        d_poses_n = reinterpret_cast<double*>(A + b_p);
        d_poses_w = reinterpret_cast<double*>(A + 2 * b_p);
        A += 3 * b_p;
        d_X = reinterpret_cast<double*>(A);
        d_Xn = reinterpret_cast<double*>(A + b_x);
        d_Xw = reinterpret_cast<double*>(A + 2 * b_x);
        A += 3 * b_x;
        d_T = reinterpret_cast<double*>(A);
        d_Tn = reinterpret_cast<double*>(A + b_t);
        d_Tw = reinterpret_cast<double*>(A + 2 * b_t);
        A += 3 * b_t;
        d_echi = reinterpret_cast<double*>(A);
        d_edepth = A + b_e;
        A += b_e + b_d;
        d_echi_r1 = reinterpret_cast<double*>(A);
        d_echi_s = reinterpret_cast<double*>(A + b_e);
        A += 2 * b_e;
        d_out1 = A;
        d_outf = A + b_d;
        d_nact = reinterpret_cast<int32_t*>(A + 2 * b_d);
        // padded system | bp | staging of the keyframes' records (7 + 12 per keyframe) | chi3, scal, fail | a trial's result block (264 bytes)
        stage_off = ovs::dense_solve_doubles(6 * np) + 6 * (size_t)np;
        pin_doubles = stage_off + 19 * (size_t)np + 16 + 40 + 16;   // (+ 16: the trial's outcome as flag-carrying words, see run_round)
        if (sc.pin_cap < pin_doubles) {
            if (sc.h_pin) (void)hipHostFree(sc.h_pin);
            sc.h_pin = nullptr;
            sc.pin_cap = 0;
            OVS_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sc.h_pin), sizeof(double) * pin_doubles, hipHostMallocDefault));
            sc.pin_cap = pin_doubles;
        }
        h_pin = sc.h_pin;
        const size_t ne = std::max<size_t>(ne_max, 1);
        if (sc.edge_cap < ne) {   // (pageable std::vectors here cost a staged 0.9 MB copy and fresh pages twice per call: ~0.3 ms of a 7.7 ms call)
            if (sc.h_edge) (void)hipHostFree(sc.h_edge);
            sc.h_edge = nullptr;
            sc.edge_cap = 0;
            const size_t cap = (ne + 1023) & ~(size_t)1023;
            OVS_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sc.h_edge), 2 * cap * 9, hipHostMallocDefault));
            sc.edge_cap = cap;
        }
        h_edge = sc.h_edge;
        edge_cap = sc.edge_cap;
        if (sc.pts_cap < (size_t)3 * npt) {
            if (sc.h_pts) (void)hipHostFree(sc.h_pts);
            sc.h_pts = nullptr;
            sc.pts_cap = 0;
            const size_t cap = ((size_t)3 * npt + (size_t)3 * npt / 4 + 511) & ~(size_t)511;
            OVS_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sc.h_pts), sizeof(double) * cap, hipHostMallocDefault));
            sc.pts_cap = cap;
        }
        h_pts = sc.h_pts;
        // through the page-locked block, no wait: the landmarks travel while the caller (ovs_local_ba_optimize) indexes the edges; the block is
        // written next at the end of the call, behind a dozen stream synchronisations
        std::memcpy(h_pts, points, sizeof(double) * 3 * (size_t)npt);
        OVS_HIP_TRY(hipMemcpyAsync(d_X, h_pts, sizeof(double) * 3 * (size_t)npt, hipMemcpyHostToDevice, stream));
        return OVS_OK;
    }

    static void pack_poses(const std::vector<Pose>& T, std::vector<double>& p7) {
        p7.resize(7 * T.size());
        for (size_t k = 0; k < T.size(); ++k) {
            p7[7 * k] = T[k].t[0];
            p7[7 * k + 1] = T[k].t[1];
            p7[7 * k + 2] = T[k].t[2];
            rot_to_quat(T[k].R, &p7[7 * k + 3]);
        }
    }

    // the 7-double records (and, for the device solver, R | t) of all keyframes through the page-locked staging area: no wait here -- the area is
    // written again only at the start of the next round, long after the round's first stream synchronisation
    ovs_status upload_poses(const std::vector<Pose>& T, double* dst, double* dst_rt) {
        std::vector<double> p7;
        pack_poses(T, p7);
        double* const st7 = h_pin + stage_off;
        std::memcpy(st7, p7.data(), sizeof(double) * p7.size());
        OVS_HIP_TRY(hipMemcpyAsync(dst, st7, sizeof(double) * p7.size(), hipMemcpyHostToDevice, stream));
        if (dst_rt) {
            double* const rt = st7 + 7 * (size_t)n_pose;
            for (int k = 0; k < n_pose; ++k) {
                std::memcpy(rt + (size_t)12 * k, T[k].R, sizeof(double) * 9);
                std::memcpy(rt + (size_t)12 * k + 9, T[k].t, sizeof(double) * 3);
            }
            OVS_HIP_TRY(hipMemcpyAsync(dst_rt, rt, sizeof(double) * 12 * (size_t)n_pose, hipMemcpyHostToDevice, stream));
        }
        return OVS_OK;
    }

    double huber_mono(bool robust) const { return robust ? (setup_type == 0 ? kSqrtChi2D : kSqrtChi3D) : 0.0; }
    double huber_stereo(bool robust) const { return robust ? kSqrtChi3D : 0.0; }

    // one optimizer.optimize(iters) call on graph g. Returns the number of iterations entered.
    ovs_status run_round(ovs_ba_graph* g, std::vector<Pose>& T, int iters, bool robust, const volatile uint8_t* stop, double* chi_start,
                         double* chi_end, int* n_iter) {
        ovs_status st = ovs::ba_graph_ensure_solver(g, stream);   // (before ba_graph_info: the work space is allocated on first use)
        if (st != OVS_OK) return st;
        const ovs::BaGraphInfo gi = ovs::ba_graph_info(g);
        const int nf = gi.n_free, n = 6 * nf;
        // the reduced camera system is solved where ovs_local_ba_set_solver says; systems beyond the one-workgroup solver's LDS go to the host
        const bool dev_solve = ovs::g_lba_solver.load(std::memory_order_relaxed) == 0 && n <= ovs::dense_solve_max_n();
        st = upload_poses(T, d_poses, dev_solve ? d_T : nullptr);
        if (st != OVS_OK) return st;
        // both failure words start a round clear: the device solver's trials rely on the PREVIOUS trial's k_trial_update to clear the word they
        // are about to use, which a round that ran on the host solver (ovs_local_ba_set_solver between rounds) has not done
        if (dev_solve) OVS_HIP_TRY(hipMemsetAsync(gi.d_fail, 0, 2 * sizeof(int32_t), stream));
        st = ovs::ba_graph_linearize(g, d_poses, d_X, huber_mono(robust), huber_stereo(robust), cur.Hpp, cur.bp, cur.Hll, cur.bl, cur.Hpl, cur.chi,
                                     stream);
        if (st != OVS_OK) return st;
        double* h_chi = h_pin + pin_doubles - 16 - 40 - 16;
        double* h_blk = h_pin + pin_doubles - 40 - 16;
        volatile unsigned long long* const h_ll = reinterpret_cast<volatile unsigned long long*>(h_pin + pin_doubles - 16);
        static thread_local unsigned int ll_seq = 0;   // sequence number of a trial's words: per thread, like the page-locked block they land in
        for (int i = 0; i < 16; ++i) h_ll[i] = 0ull;    // (the stream is idle here; whatever the block's last user left cannot pass for a word)
        static const bool ll_notify = [] {   // OVS_BA_LL_NOTIFY=0: a trial's outcome through a D2H copy and a stream synchronisation (rounds 4-5)
            const char* e = std::getenv("OVS_BA_LL_NOTIFY");
            return !(e && e[0] == '0');
        }();   // one download per trial: [0] landmarks' / [1] keyframes' gain-ratio parts, [2..4] chi2 triple, byte 256: fail flag
        OVS_HIP_TRY(hipMemcpyAsync(h_chi, cur.chi, sizeof(double) * 3, hipMemcpyDeviceToHost, stream));
        OVS_HIP_TRY(hipStreamSynchronize(stream));
        double current_chi = h_chi[1];
        *chi_start = current_chi;
        *chi_end = current_chi;
        *n_iter = 0;
        err_at_trial = false;
        if (iters <= 0) return OVS_OK;
        // computeLambdaInit: tau * the largest diagonal entry of the active vertices' Hessian blocks
        double lambda = 1e-5 * h_chi[2], ni = 2;
        std::vector<double> S, rhs, dxp((size_t)6 * n_pose, 0.0);
        std::vector<Pose> Tn;
        for (int it = 0; it < iters; ++it) {
            if (stop && *stop) break;
            ++*n_iter;
            double rho = 0;
            int qmax = 0;
            err_at_trial = false;   // solve() starts with computeActiveErrors() at the current estimate
            do {
                const double t0 = now();
                ++n_trials;
                const int fw = dev_solve ? (n_trials & 1) : 0;   // the device solver's trials alternate between two failure words
                st = ovs::ba_graph_schur(g, cur.Hpp, cur.bp, cur.Hll, cur.bl, cur.Hpl, lambda, stream, fw, !dev_solve);
                if (st != OVS_OK) return st;
                int32_t* h_fail = reinterpret_cast<int32_t*>(h_chi + 8);
                if (dev_solve) {
                    // ---- the whole trial on the device: solve, keyframe / landmark updates, linearisation at the trial state (into the
                    //      `work` set: a failed solve must leave the last evaluated trial state, which edge_chi2 may still need, untouched)
                    if (n > 0) {
                        st = ovs::launch_dense_solve(gi.d_S, n, gi.d_fail + fw, stream);
                        if (st != OVS_OK) return st;
                    }
                    st = ovs::ba_graph_trial_update(g, d_T, cur.bp, cur.Hpl, cur.bl, lambda, d_Tw, d_poses_w, d_X, d_Xw, stream, fw ^ 1);
                    if (st != OVS_OK) return st;
                    // the trial state's chi2 triple is mirrored next to the solver's scalars (gi.d_scal[2..4]; gi.d_fail sits 256 bytes behind
                    // gi.d_scal in the same arena): ONE 260-byte download per trial instead of three copies (round 5: two copy launches and their
                    // gaps less per trial). (One kernel writing the values straight into the page-locked block was measured in round 4 -- the
                    // system-scope flush at its end costs ~50 us per trial.)
                    const unsigned int seq = ++ll_seq == 0u ? ++ll_seq : ll_seq;   // (never 0: the block starts zeroed)
                    st = ovs::ba_graph_linearize(g, d_poses_w, d_Xw, huber_mono(robust), huber_stereo(robust), work.Hpp, work.bp, work.Hll, work.bl,
                                                 work.Hpl, work.chi, stream, gi.d_scal + 2, true,
                                                 ll_notify ? const_cast<unsigned long long*>(h_ll) : nullptr, seq);
                    if (st != OVS_OK) return st;
                    bool polled = false;
                    if (ll_notify) {
                        // Round 6: the trial's last kernel wrote the outcome into the page-locked block as twelve words {seq | half a double};
                        // poll them (no copy command, no stream wait). Every 4096 polls the stream is asked whether it still runs: a launch
                        // that died would otherwise never write the words.
                        unsigned long long w[12];
                        unsigned int spins = 0;
                        bool dead = false;
                        for (int i = 0; i < 12 && !dead;) {
                            w[i] = h_ll[i];
                            if ((unsigned int)(w[i] >> 32) == seq) {
                                ++i;
                                continue;
                            }
                            if ((++spins & 4095u) == 0u) {
                                const hipError_t q = hipStreamQuery(stream);
                                if (q != hipErrorNotReady) {   // idle (or failed): one last look, then the copy path decides
                                    w[i] = h_ll[i];
                                    if ((unsigned int)(w[i] >> 32) == seq) continue;
                                    dead = true;
                                }
                            }
                        }
                        if (!dead) {
                            auto val = [&](int i) {
                                const unsigned long long bits = (w[2 * i] & 0xffffffffull) | (w[2 * i + 1] << 32);
                                double d;
                                std::memcpy(&d, &bits, sizeof(d));
                                return d;
                            };
                            h_chi[4] = val(0);
                            h_chi[5] = val(1);
                            h_chi[0] = val(2);
                            h_chi[1] = val(3);
                            h_chi[2] = val(4);
                            const uint32_t f2[2] = {(uint32_t)(w[10] & 0xffffffffull), (uint32_t)(w[11] & 0xffffffffull)};
                            *h_fail = (int32_t)f2[fw];
                            polled = true;
                        }
                    }
                    if (!polled) {
                        OVS_HIP_TRY(hipMemcpyAsync(h_blk, gi.d_scal, 256 + 2 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                        OVS_HIP_TRY(hipStreamSynchronize(stream));   // (polling hipStreamQuery instead: the same 6.5-6.6 ms per call, round 5)
                        h_chi[0] = h_blk[2];
                        h_chi[1] = h_blk[3];
                        h_chi[2] = h_blk[4];
                        h_chi[4] = h_blk[0];
                        h_chi[5] = h_blk[1];
                        *h_fail = reinterpret_cast<const int32_t*>(reinterpret_cast<const unsigned char*>(h_blk) + 256)[fw];
                    }
                    const bool ok = *h_fail == 0;
                    double temp_chi = 1.7976931348623157e308, scale = 1e-3;
                    if (!ok) {   // a failed factorisation may have left non-finite values in the padding, which no later trial rewrites
                        st = ovs::ba_graph_reset_system(g, stream);
                        if (st != OVS_OK) return st;
                    }
                    if (ok) {
                        temp_chi = h_chi[1];
                        scale = (h_chi[5] + h_chi[4]) + 1e-3;   // keyframes' part, then the landmarks' (g2o's computeScale order)
                        std::swap(trial, work);
                        std::swap(d_poses_n, d_poses_w);
                        std::swap(d_Xn, d_Xw);
                        std::swap(d_Tn, d_Tw);
                        err_at_trial = true;
                    }
                    t_trial += now() - t0;
                    rho = (current_chi - temp_chi) / scale;
                    if (ok && rho > 0 && std::isfinite(temp_chi)) {
                        double alpha = 1.0 - std::pow(2 * rho - 1, 3.0);
                        alpha = std::min(alpha, 2.0 / 3.0);
                        lambda *= std::max(1.0 / 3.0, alpha);
                        ni = 2;
                        current_chi = temp_chi;
                        std::swap(d_X, d_Xn);
                        std::swap(d_poses, d_poses_n);
                        std::swap(d_T, d_Tn);
                        std::swap(cur, trial);
                        err_at_trial = false;
                    } else {
                        lambda *= ni;
                        ni *= 2;
                        if (!std::isfinite(lambda)) break;
                    }
                    ++qmax;
                    continue;
                }
                const size_t np_ = (size_t)gi.s_pitch, sys_rows = np_ + 1;   // S rows and the rhs row
                double* const h_bp_w = h_pin + sys_rows * np_;
                if (n > 0) {
                    OVS_HIP_TRY(hipMemcpyAsync(h_pin, gi.d_S, sizeof(double) * sys_rows * np_, hipMemcpyDeviceToHost, stream));
                    OVS_HIP_TRY(hipMemcpyAsync(h_bp_w, cur.bp, sizeof(double) * 6 * (size_t)n_pose, hipMemcpyDeviceToHost, stream));
                }
                OVS_HIP_TRY(hipMemcpyAsync(h_fail, gi.d_fail, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                OVS_HIP_TRY(hipStreamSynchronize(stream));
                bool ok = *h_fail == 0;
                const double t1 = now();
                t_schur += t1 - t0;
                const double* h_bp = h_bp_w;
                if (ok && n > 0) {
                    S.resize((size_t)n * n);
                    for (int i = 0; i < n; ++i) std::memcpy(&S[(size_t)i * n], h_pin + (size_t)i * np_, sizeof(double) * n);   // drop the padding
                    rhs.assign(h_pin + np_ * np_, h_pin + np_ * np_ + n);
                    ok = cholesky_solve(S, n, rhs);
                }
                const double t2 = now();
                t_chol += t2 - t1;
                double temp_chi = 1.7976931348623157e308;
                double scale = 1e-3;
                if (ok) {
                    std::fill(dxp.begin(), dxp.end(), 0.0);
                    Tn = T;
                    for (int k = 0; k < n_pose; ++k)
                        if (gi.slot[k] >= 0) {
                            for (int a = 0; a < 6; ++a) dxp[(size_t)6 * k + a] = rhs[(size_t)6 * gi.slot[k] + a];
                            se3_oplus(Tn[k], &dxp[(size_t)6 * k]);
                        }
                    std::vector<double> p7;
                    pack_poses(Tn, p7);
                    OVS_HIP_TRY(hipMemcpyAsync(gi.d_dxp, dxp.data(), sizeof(double) * dxp.size(), hipMemcpyHostToDevice, stream));
                    OVS_HIP_TRY(hipMemcpyAsync(d_poses_n, p7.data(), sizeof(double) * p7.size(), hipMemcpyHostToDevice, stream));
                    st = ovs::ba_graph_backsub(g, cur.Hpl, cur.bl, lambda, d_X, d_Xn, stream);
                    if (st != OVS_OK) return st;
                    st = ovs::ba_graph_linearize(g, d_poses_n, d_Xn, huber_mono(robust), huber_stereo(robust), trial.Hpp, trial.bp, trial.Hll,
                                                 trial.bl, trial.Hpl, trial.chi, stream, nullptr, true);
                    if (st != OVS_OK) return st;
                    OVS_HIP_TRY(hipMemcpyAsync(h_chi, trial.chi, sizeof(double) * 3, hipMemcpyDeviceToHost, stream));
                    OVS_HIP_TRY(hipMemcpyAsync(h_chi + 4, gi.d_scal, sizeof(double), hipMemcpyDeviceToHost, stream));
                    OVS_HIP_TRY(hipStreamSynchronize(stream));   // also covers dxp / p7 (locals)
                    temp_chi = h_chi[1];
                    err_at_trial = true;   // computeActiveErrors() ran on the trial state (d_poses_n, d_Xn); cleared below if it is accepted
                    double sc = 0;
                    for (int k = 0; k < n_pose; ++k)
                        if (gi.slot[k] >= 0)
                            for (int a = 0; a < 6; ++a) sc += dxp[(size_t)6 * k + a] * (lambda * dxp[(size_t)6 * k + a] + h_bp[(size_t)6 * k + a]);
                    scale = (sc + h_chi[4]) + 1e-3;
                }
                t_trial += now() - t2;
                rho = (current_chi - temp_chi) / scale;
                if (ok && rho > 0 && std::isfinite(temp_chi)) {
                    double alpha = 1.0 - std::pow(2 * rho - 1, 3.0);
                    alpha = std::min(alpha, 2.0 / 3.0);
                    lambda *= std::max(1.0 / 3.0, alpha);
                    ni = 2;
                    current_chi = temp_chi;
                    T.swap(Tn);
                    std::swap(d_X, d_Xn);
                    std::swap(d_poses, d_poses_n);
                    std::swap(cur, trial);   // the accepted trial's blocks are the next iteration's system
                    err_at_trial = false;    // the trial state is the estimate now
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
        if (dev_solve) {   // the accepted keyframe state comes back once per round
            std::vector<double> rt((size_t)12 * n_pose);
            OVS_HIP_TRY(hipMemcpyAsync(rt.data(), d_T, sizeof(double) * rt.size(), hipMemcpyDeviceToHost, stream));
            OVS_HIP_TRY(hipStreamSynchronize(stream));
            for (int k = 0; k < n_pose; ++k) {
                std::memcpy(T[k].R, &rt[(size_t)12 * k], sizeof(double) * 9);
                std::memcpy(T[k].t, &rt[(size_t)12 * k + 9], sizeof(double) * 3);
            }
        }
        return OVS_OK;
    }

    // What upstream reads after optimizer.optimize(): edge->chi2() -- the error STORED by the last computeActiveErrors(), i.e. at the last
    // LM trial state when the round ended on a rejected step (g2o pops the estimate back but leaves the errors) -- and
    // edge->depth_is_positive(), which is evaluated from the vertices' current, accepted estimates (T, d_X).
    ovs_status edge_chi2(ovs_ba_graph* g, const std::vector<Pose>& T, size_t ne, int slot, const double*& chi, const uint8_t*& depth) {
        ovs_status st = upload_poses(T, d_poses, nullptr);
        if (st != OVS_OK) return st;
        double* const h_chi = reinterpret_cast<double*>(h_edge + (size_t)slot * edge_cap * 9);
        uint8_t* const h_depth = h_edge + (size_t)slot * edge_cap * 9 + edge_cap * 8;
        chi = h_chi;
        depth = h_depth;
        if (err_at_trial) {
            st = ovs::ba_graph_edge_chi2(g, d_poses_n, d_Xn, d_echi, d_edepth, stream);
            if (st != OVS_OK) return st;
            if (ne) OVS_HIP_TRY(hipMemcpyAsync(h_chi, d_echi, sizeof(double) * ne, hipMemcpyDeviceToHost, stream));
        }
        st = ovs::ba_graph_edge_chi2(g, d_poses, d_X, d_echi, d_edepth, stream);
        if (st != OVS_OK) return st;
        if (ne) {
            if (!err_at_trial) OVS_HIP_TRY(hipMemcpyAsync(h_chi, d_echi, sizeof(double) * ne, hipMemcpyDeviceToHost, stream));
            OVS_HIP_TRY(hipMemcpyAsync(h_depth, d_edepth, ne, hipMemcpyDeviceToHost, stream));
        }
        OVS_HIP_TRY(hipStreamSynchronize(stream));
        return OVS_OK;
    }

    // edge_chi2's two evaluations without the downloads (round 6): the chi2 upstream would read goes to `d_chi_judged`, the depth flags of the
    // accepted state to d_edepth. Nothing is waited for.
    ovs_status edge_chi2_dev(ovs_ba_graph* g, const std::vector<Pose>& T, double* d_chi_judged) {
        ovs_status st = upload_poses(T, d_poses, nullptr);
        if (st != OVS_OK) return st;
        if (err_at_trial) {
            st = ovs::ba_graph_edge_chi2(g, d_poses_n, d_Xn, d_chi_judged, d_edepth, stream);
            if (st != OVS_OK) return st;
        }
        return ovs::ba_graph_edge_chi2(g, d_poses, d_X, err_at_trial ? d_echi_s : d_chi_judged, d_edepth, stream);
    }
    // after round 1: flags into d_out1, the graph's active mask, the number of inliers (one 4-byte download and the wait for it)
    ovs_status gate_round1(ovs_ba_graph* g, const std::vector<Pose>& T, size_t ne, size_t* n_act) {
        *n_act = 0;
        if (ne == 0) return OVS_OK;
        OVS_HIP_TRY(hipMemsetAsync(d_nact, 0, sizeof(int32_t), stream));
        ovs_status st = edge_chi2_dev(g, T, d_echi_r1);
        if (st != OVS_OK) return st;
        st = ovs::ba_graph_edge_gate(g, kChi2D, kChi3D, d_echi_r1, d_edepth, nullptr, nullptr, false, d_out1, true, d_nact, stream);
        if (st != OVS_OK) return st;
        int32_t* const h_n = reinterpret_cast<int32_t*>(h_edge);
        OVS_HIP_TRY(hipMemcpyAsync(h_n, d_nact, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        OVS_HIP_TRY(hipStreamSynchronize(stream));
        *n_act = (size_t)*h_n;
        return OVS_OK;
    }
    // the final flags (ne bytes, into the page-locked block) and the landmarks (into h_pts) in one wait
    ovs_status gate_final(ovs_ba_graph* g, const std::vector<Pose>& T, size_t ne, bool round2_ran, const uint8_t*& flags) {
        flags = h_edge;
        if (ne) {
            ovs_status st = edge_chi2_dev(g, T, d_echi);
            if (st != OVS_OK) return st;
            st = ovs::ba_graph_edge_gate(g, kChi2D, kChi3D, d_echi, d_edepth, d_echi_r1, d_out1, round2_ran, d_outf, false, nullptr, stream);
            if (st != OVS_OK) return st;
            OVS_HIP_TRY(hipMemcpyAsync(h_edge, d_outf, ne, hipMemcpyDeviceToHost, stream));
        }
        OVS_HIP_TRY(hipMemcpyAsync(h_pts, d_X, sizeof(double) * 3 * (size_t)n_pt, hipMemcpyDeviceToHost, stream));
        OVS_HIP_TRY(hipStreamSynchronize(stream));
        return OVS_OK;
    }
};

struct GraphGuard {
    ovs_ba_graph* g = nullptr;
    ~GraphGuard() {
        if (g) ovs_ba_graph_destroy(g);
    }
};

}   // namespace

// model 0: perspective (ovs_ba_graph_create), model 1: equirectangular (cam = {cols, rows, -, -}, mono edges only)
static ovs_status local_ba_optimize_impl(int model, int32_t device, double* poses, const uint8_t* pose_fixed, int32_t n_pose, double* points,
                                         int32_t n_pt, const ovs_ba_edge* mono, int32_t n_mono, const ovs_ba_edge_stereo* stereo, int32_t n_stereo,
                                         const ovs_ba_cam* cam, double focal_x_baseline, int32_t setup_type, int32_t num_first_iter,
                                         int32_t num_second_iter, const volatile uint8_t* force_stop_flag, uint8_t* mono_outlier,
                                         uint8_t* stereo_outlier, double* info) {
    if (!poses || !points || !cam || n_pose < 1 || n_pt < 1 || n_mono < 0 || n_stereo < 0 || (n_mono > 0 && (!mono || !mono_outlier)) ||
        (n_stereo > 0 && (!stereo || !stereo_outlier)) || num_first_iter < 0 || num_second_iter < 0)
        return OVS_ERR_INVALID;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    OVS_HIP_TRY(hipSetDevice(device));
    // ---- the work space first (round 6): the landmarks' upload is on its way while the host indexes the edges
    const bool trace = ovs::tuning().ba_trace;
    const bool dev_gate = ovs::tuning().ba_dev_outliers;
    const double t_begin = Lm::now();
    const size_t ne = (size_t)n_mono + n_stereo;
    Lm L;
    L.setup_type = setup_type;
    ovs_status st = L.init(device, n_pose, n_pt, ne, points);
    if (st != OVS_OK) return st;
    // ---- round 1 graph: all edges (validates the indices)
    GraphGuard g1;
    st = model == 1 ? ovs_ba_graph_create_equirect(device, n_pose, pose_fixed, n_pt, mono, n_mono, (int32_t)cam->fx, (int32_t)cam->fy, &g1.g)
                    : ovs_ba_graph_create(device, n_pose, pose_fixed, n_pt, mono, n_mono, stereo, n_stereo, cam, focal_x_baseline, &g1.g);
    if (st != OVS_OK) {
        (void)hipStreamSynchronize(L.stream);   // (the landmarks' upload reads the thread's page-locked block)
        return st;
    }
    const double t_g1 = Lm::now();
    std::vector<Pose> T((size_t)n_pose);
    for (int k = 0; k < n_pose; ++k) {
        quat_to_rot(poses + 7 * (size_t)k + 3, T[k].R);
        for (int a = 0; a < 3; ++a) T[k].t[a] = poses[7 * (size_t)k + a];
    }
    double info_l[6] = {0, 0, 0, 0, 0, 0};
    int it1 = 0, it2 = 0;
    if (ne == 0) num_first_iter = num_second_iter = 0;
    st = L.run_round(g1.g, T, num_first_iter, true, force_stop_flag, &info_l[0], &info_l[1], &it1);
    if (st != OVS_OK) return st;
    const double t_r1 = Lm::now();
    double t_gate1 = t_r1, t_r2 = t_r1;
    if (dev_gate) {
        // ---- round 6: the chi-square gates on the device (k_edge_gate): the per-edge arrays stay in HBM, the active mask is written where
        //      round 2 reads it, 4 bytes come down after round 1 and the flags (one byte per edge) at the end
        size_t n_act = 0;
        st = L.gate_round1(g1.g, T, ne, &n_act);
        if (st != OVS_OK) return st;
        t_gate1 = Lm::now();
        const bool stopped = force_stop_flag && *force_stop_flag;
        if (!stopped) {
            st = L.run_round(g1.g, T, n_act ? num_second_iter : 0, false, force_stop_flag, &info_l[2], &info_l[3], &it2);
            if (st != OVS_OK) return st;
        }
        t_r2 = Lm::now();
        const uint8_t* flags = nullptr;
        st = L.gate_final(g1.g, T, ne, !stopped, flags);
        if (st != OVS_OK) return st;
        if (n_mono > 0) std::memcpy(mono_outlier, flags, (size_t)n_mono);
        if (n_stereo > 0) std::memcpy(stereo_outlier, flags + n_mono, (size_t)n_stereo);
    } else {
        const double *chi_r1 = nullptr, *chi = nullptr;   // (page-locked slots of the scratch: round 1's stays valid beside the final one)
        const uint8_t* depth = nullptr;
        st = L.edge_chi2(g1.g, T, ne, 0, chi_r1, depth);
        if (st != OVS_OK) return st;
        chi = chi_r1;
        std::vector<uint8_t> out_r1(ne);
        for (int i = 0; i < n_mono; ++i) out_r1[i] = (kChi2D < chi[i]) || !depth[i];
        for (int i = 0; i < n_stereo; ++i) out_r1[(size_t)n_mono + i] = (kChi3D < chi[(size_t)n_mono + i]) || !depth[(size_t)n_mono + i];
        t_gate1 = Lm::now();
        const bool stopped = force_stop_flag && *force_stop_flag;
        if (!stopped) {
            // ---- round 2: inliers only (outliers go to level 1), no robust kernel. The graph is kept: level-1 edges are masked, they then
            // contribute exact zeros and the sums over the remaining edges keep their order -- the result a rebuilt graph would give, without
            // indexing 100 k edges a second time
            std::vector<uint8_t> act(ne);
            size_t n_act = 0;
            for (size_t e = 0; e < ne; ++e) n_act += (act[e] = out_r1[e] ? 0 : 1);
            st = ovs::ba_graph_set_active(g1.g, act.data(), L.stream);
            if (st != OVS_OK) return st;
            st = L.run_round(g1.g, T, n_act ? num_second_iter : 0, false, force_stop_flag, &info_l[2], &info_l[3], &it2);
            if (st != OVS_OK) return st;
        }
        t_r2 = Lm::now();
        // ---- final outlier flags: an edge optimised in round 2 is judged at the final state; a level-1 edge keeps its round-1 chi2
        //      (g2o does not recompute the error of inactive edges) but its depth test sees the final state
        st = L.edge_chi2(g1.g, T, ne, 1, chi, depth);
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
        OVS_HIP_TRY(hipMemcpyAsync(L.h_pts, L.d_X, sizeof(double) * 3 * (size_t)n_pt, hipMemcpyDeviceToHost, L.stream));
        OVS_HIP_TRY(hipStreamSynchronize(L.stream));
    }
    std::vector<double> p7;
    Lm::pack_poses(T, p7);
    for (int k = 0; k < n_pose; ++k)
        if (!(pose_fixed && pose_fixed[k])) std::memcpy(poses + 7 * (size_t)k, &p7[(size_t)7 * k], sizeof(double) * 7);
    std::memcpy(points, L.h_pts, sizeof(double) * 3 * (size_t)n_pt);
    if (trace)
        std::fprintf(stderr, "[ovs_local_ba_optimize] total %.2f ms: work space + graph build %.2f, round 1 %.2f, gates %.2f, round 2 %.2f, final gates + "
                             "download %.2f; %d trials: schur+download %.2f, host cholesky %.2f, update+linearise (device solver: the whole trial) %.2f ms\n",
                     Lm::now() - t_begin, t_g1 - t_begin, t_r1 - t_g1, t_gate1 - t_r1, t_r2 - t_gate1, Lm::now() - t_r2, L.n_trials, L.t_schur, L.t_chol,
                     L.t_trial);
    if (info) {
        info_l[4] = it1;
        info_l[5] = it2;
        std::memcpy(info, info_l, sizeof(info_l));
    }
    return OVS_OK;
}

extern "C" {

ovs_status ovs_local_ba_optimize(int32_t device, double* poses, const uint8_t* pose_fixed, int32_t n_pose, double* points, int32_t n_pt,
                                 const ovs_ba_edge* mono, int32_t n_mono, const ovs_ba_edge_stereo* stereo, int32_t n_stereo,
                                 const ovs_ba_cam* cam, double focal_x_baseline, int32_t setup_type, int32_t num_first_iter, int32_t num_second_iter,
                                 const volatile uint8_t* force_stop_flag, uint8_t* mono_outlier, uint8_t* stereo_outlier, double* info) {
    return local_ba_optimize_impl(0, device, poses, pose_fixed, n_pose, points, n_pt, mono, n_mono, stereo, n_stereo, cam, focal_x_baseline, setup_type,
                                  num_first_iter, num_second_iter, force_stop_flag, mono_outlier, stereo_outlier, info);
}

ovs_status ovs_local_ba_set_solver(int32_t where) {
    if (where != 0 && where != 1) return OVS_ERR_INVALID;
    ovs::g_lba_solver.store(where, std::memory_order_relaxed);
    return OVS_OK;
}
int32_t ovs_local_ba_get_solver(void) { return ovs::g_lba_solver.load(std::memory_order_relaxed); }

ovs_status ovs_local_ba_optimize_equirect(int32_t device, double* poses, const uint8_t* pose_fixed, int32_t n_pose, double* points, int32_t n_pt,
                                          const ovs_ba_edge* mono, int32_t n_mono, int32_t cols, int32_t rows, int32_t num_first_iter,
                                          int32_t num_second_iter, const volatile uint8_t* force_stop_flag, uint8_t* mono_outlier, double* info) {
    if (cols < 1 || rows < 1) return OVS_ERR_INVALID;
    const ovs_ba_cam cam = {(double)cols, (double)rows, 0.0, 0.0};
    return local_ba_optimize_impl(1, device, poses, pose_fixed, n_pose, points, n_pt, mono, n_mono, nullptr, 0, &cam, 0.0, 0, num_first_iter,
                                  num_second_iter, force_stop_flag, mono_outlier, nullptr, info);
}

}   // extern "C"
