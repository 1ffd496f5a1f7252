// ba_solve.hip -- the reduced camera system of local BA on the device (round 4):
//   k_chol_solve    dense Cholesky + both substitutions of  S x = rhs  (S = 6 n_free square, symmetric positive definite after the
//                   landmarks were eliminated: ba_graph.hip k_schur) in ONE workgroup, the trailing updates on the
//                   f64 matrix cores; the trailing tiles travel through memory: systems of 289 .. 1024 unknowns;
//   k_chol_resident (round 5) the same factorisation with the trailing tiles resident in registers / LDS: systems up to 288 unknowns
//                   (BASELINE config 5), 168 instead of 252 us per solve; described where it is defined, below;
// (The keyframes' trial state -- T <- exp(dx) T -- moved to ba_graph.hip in round 5, where it shares a launch with the back-substitution.)
// Upstream solves this system on the host (g2o BlockSolver + a dense / CSparse Cholesky inside optimizer.optimize(), expected
// src/openvslam/optimize/local_bundle_adjuster.cc); until round 4 so did this library (ba_host_math.h cholesky_solve, still selectable:
// ovs_local_ba_set_solver(1)). One LM trial then cost a 0.66 MB download, 0.47 ms of host arithmetic and two uploads, 15 times per call
// at BASELINE config 5 (288 x 288) -- more than the linearisations. Here a trial never leaves the device: the host reads back three
// scalars and a flag.
//
// Algorithm (right-looking, blocked by 16 columns, lower triangle, in place in HBM / L2; the matrix is 0.66 MB at config 5), 8 waves:
//   for every panel j0:  the panel (rows j0.. of columns [j0, j0 + 16)) sits in LDS; wave 0 factors the 16 x 16 diagonal block in registers
//   (one row per lane, broadcasts by v_readlane, reciprocal square roots by v_rsq_f64 + Newton); one thread per remaining row solves its 16
//   entries against the block (the block's entries broadcast from registers too); the trailing matrix gets C -= P_i P_c^T per 16 x 16 tile
//   with four v_mfma_f64_16x16x4_f64 (A = -P_i, B = P_c from LDS, C from / to memory) -- a wave's first tiles are requested BEFORE the panel is
//   factored, later batches one batch ahead, and barriers inside a panel order LDS only, so the requests stay in flight; the tiles of the
//   next panel's columns are written straight into a second LDS panel (systems up to 528 unknowns).
//   The right-hand side rides along as one more ROW of the matrix: its "row solve" is the forward substitution. The backward substitution
//   L^T x = y runs block-wise from the last block, right-looking: wave 0 solves a 16 x 16 block (from the block's L^T, which the panel
//   write-back leaves above the diagonal), then every thread takes the block's columns out of its own unknown, with the rows of L it needs
//   for the NEXT block already requested.
//   The system is stored padded to a multiple of 16 with an identity block (layout below).
// Where the time goes at 288 unknowns (OVS_BA_TRACE=1 with ovs_ba_dense_solve; ~240 us per solve): the trailing tiles travel through one
// compute unit's 64 bytes per clock (8 MB read + written: ~125 us with the requests), the 288 pivot steps of the diagonal blocks are one
// dependent chain (~50 us), row solves ~25 us, backward substitution ~35 us. Tried and dropped in round 4: (i) the whole triangle resident in
// registers as matrix-core accumulators (189 tiles = 378 KB of the 512 KB register file: needs one wave per SIMD with 512 registers, and the
// compiler spills the other phases' arrays); (ii) the last tile columns resident in LDS (27 tiles, 40 % of the tile updates): no change --
// the batches wait on s_waitcnt vmcnt(0), i.e. also on the requests just issued for the next batch, because loads under a branch cannot be
// counted; (iii) therefore unconditional, software-pipelined requests 12 tiles deep (the compiler then emits vmcnt(44..47)): 48 loads per
// lane in flight from 8 waves saturate the compute unit's miss queue instead, requests 39 -> 65 us, no net gain. One compute unit moves
// ~30-50 GB/s of 128-byte tile rows; spreading the tiles over more compute units costs a panel broadcast and two grid barriers per panel,
// about what it saves at this size.
// Numerics: fused multiply-adds and the matrix cores' internal order instead of the host's mul / sub pairs in column order -- results
// agree with LAPACK's to ~cond x 1e-16 relative (tests/test_gpu_ba.py::test_dense_solve_matches_numpy), far inside the 1e-7 the optimiser's
// result is stated to (ORACLE_SPEC rule 28); identical bits from run to run (no atomics, fixed tile order).
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

namespace {

constexpr int kNb = 16;       // panel width = tile edge
constexpr int kPitch = 17;    // doubles per staged panel row (odd: rows of a tile fall into different banks)
constexpr int kSolveThreads = 512;   // 8 waves, 256 registers each: two batches of kBatch C tiles (4 doubles each) stay in flight per wave
constexpr int kBatch = 6;
constexpr int kMaxN = 1024;          // unknowns; the backward substitution gives a thread kMaxN / kSolveThreads columns

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double lane_bcast(double v, int src_lane) {   // src_lane is wave-uniform
    union {
        double d;
        int i[2];
    } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], src_lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], src_lane);
    return u.d;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Storage of the system (ba_graph.hip lays the reduced camera system out like this): (n_pad + 16) rows of n_pad doubles, n_pad = n rounded up
// to the panel width. Rows / columns n .. n_pad - 1 carry an identity block, row n_pad is the right-hand side, the rows behind it are zero
// (they only complete the last 16-row tile). Every access below is then plain row * n_pad + column; the padding reproduces itself (the
// Cholesky factor of an identity block is the identity, products with zero rows vanish), so it is written once per graph.
}   // namespace

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains every outstanding global load (s_waitcnt vmcnt(0)), i.e. the
// C tiles / L rows requested ahead of their use below. Memory written by this kernel and read back from HBM / L2 by OTHER waves (the
// trailing tiles) is ordered by the one full barrier at the end of a panel.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// column-major walk over the tiles (ti, tc) behind a panel: tc = 0 .. mt - 2 (tile columns), ti = tc .. mt - 1 (the last tile row carries the
// rhs). The first tiles are those of the NEXT panel's columns.
// the t-th tile of that walk: column tc holds mt - tc tiles, so tc = the largest integer with tc mt - tc (tc - 1) / 2 <= t
__device__ __forceinline__ void tile_seek(int& ti, int& tc, int mt, int t) {
    const float b = 2.0f * (float)mt + 1.0f;
    int c = (int)((b - __builtin_sqrtf(fmaxf(b * b - 8.0f * (float)t, 0.0f))) * 0.5f);
    c = min(max(c, 0), mt);
    while (c > 0 && c * mt - c * (c - 1) / 2 > t) --c;
    while ((c + 1) * mt - (c + 1) * c / 2 <= t) ++c;
    tc = __builtin_amdgcn_readfirstlane(c);
    ti = __builtin_amdgcn_readfirstlane(c + (t - (c * mt - c * (c - 1) / 2)));
}

// ---- the phases of one panel, shared by the two kernels below ------------------------------------------------------------------------
// diagonal block: wave 0, lane r holds row r (entries c <= r are the lower triangle); leaves L_dd in P rows 0..15 and 1 / L[c][c] in invd
__device__ __forceinline__ void factor_diagonal_block(double* __restrict__ P, double* __restrict__ invd, int j0, int lane, int* s_bad) {
    const int r = lane & 15;
    double a[kNb];
#pragma unroll
    for (int c = 0; c < kNb; ++c) a[c] = P[r * kPitch + c];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < kNb; ++k) {
        const double piv = lane_bcast(a[k], k);
        bad |= !(piv > 0.0 && piv < __builtin_inf());   // (+Inf would pass `> 0` and turn into NaN in the Newton steps)
        const double y = rsqrt_newton(piv);
        a[k] = (r == k) ? piv * y : a[k] * y;
        if (lane == k) invd[j0 + k] = y;
#pragma unroll
        for (int c = k + 1; c < kNb; ++c) a[c] = __builtin_fma(-a[k], lane_bcast(a[k], c), a[c]);   // L[c][k] sits in lane c
        __builtin_amdgcn_sched_barrier(0);   // keeps the 120 broadcasts of the unrolled nest from being hoisted (scalar register spills)
    }
    if (lane < kNb) {
#pragma unroll
        for (int c = 0; c < kNb; ++c)
            if (c <= r) P[r * kPitch + c] = a[c];
        if (bad && lane == 0) *s_bad = 1;
    }
}

// the rows below the block: x L_dd^T = p, one thread per row; the rhs row's solution is this block of y. A wave keeps the block's 136
// entries in three registers spread over its lanes (entry t = c (c + 1) / 2 + k in lane t % 64) and broadcasts them with v_readlane: read
// from LDS by every lane, the 136 uniform loads per row made this phase LDS-bound.
__device__ __forceinline__ void solve_rows(double* __restrict__ P, const double* __restrict__ invd, double* __restrict__ vec, int j0, int m,
                                           int n_pad, int tid, int lane) {
    double lreg[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = lane + 64 * u;   // -> (c, k), k <= c: c = the largest integer with c (c + 1) / 2 <= t
        int c = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        c += ((c + 1) * (c + 2) / 2 <= t) ? 1 : 0;
        c -= (c * (c + 1) / 2 > t) ? 1 : 0;
        const int k = t - c * (c + 1) / 2;
        lreg[u] = t < kNb * (kNb + 1) / 2 ? P[c * kPitch + k] : 0.0;
    }
    double ivreg = invd[j0 + (lane & 15)];
    for (int r = kNb + tid; r < m; r += kSolveThreads) {   // (v_readlane reads its lane whatever the execution mask)
        asm volatile("" : "+v"(lreg[0]), "+v"(lreg[1]), "+v"(lreg[2]), "+v"(ivreg));   // broadcasts stay inside the row loop (scalar registers)
        double x[kNb];
#pragma unroll
        for (int c = 0; c < kNb; ++c) x[c] = P[r * kPitch + c];
#pragma unroll
        for (int c = 0; c < kNb; ++c) {
            double v = x[c];
#pragma unroll
            for (int k = 0; k < c; ++k) {
                const int t = c * (c + 1) / 2 + k;
                v = __builtin_fma(-x[k], lane_bcast(lreg[t >> 6], t & 63), v);
            }
            x[c] = v * lane_bcast(ivreg, c);
        }
#pragma unroll
        for (int c = 0; c < kNb; ++c) P[r * kPitch + c] = x[c];
        if (j0 + r == n_pad) {
#pragma unroll
            for (int c = 0; c < kNb; ++c) vec[j0 + c] = x[c];   // y
        }
    }
}

// the finished panel goes back to memory (the backward substitution reads L row-wise, and the diagonal blocks' L^T)
__device__ __forceinline__ void write_back_panel(double* __restrict__ S, const double* __restrict__ P, int j0, int m, int n_pad, int tid) {
    for (int idx = tid; idx < m * kNb; idx += kSolveThreads) {
        const int r = idx >> 4, c = idx & 15;
        S[(size_t)(j0 + r) * n_pad + j0 + c] = (r < kNb && c > r) ? P[c * kPitch + r] : P[r * kPitch + c];   // (L_dd^T above the diagonal)
    }
}

// backward substitution L^T x = y, right-looking, from the last block: wave 0 solves the 16 x 16 block (lane c holds row c of L_dd^T), then
// thread c' < jb takes the block's 16 columns out of y[c']. The block's data for the NEXT step is requested before this step's arithmetic, so
// no step waits for memory. Ends with x in the rhs row.
__device__ __forceinline__ void backward_substitution(double* __restrict__ S, double* __restrict__ vec, const double* __restrict__ invd,
                                                      int n_pad, int tid, int lane, int wave) {
    const int rl = lane & 15;
    constexpr int kCols = kMaxN / kSolveThreads;   // columns c' = tid + u * kSolveThreads of a thread
    double lr[kCols][kNb], lr_next[kCols][kNb], tt[kNb];
    auto request_rows = [&](int jb, double (&L16)[kCols][kNb]) {
#pragma unroll
        for (int u = 0; u < kCols; ++u)
#pragma unroll
            for (int k = 0; k < kNb; ++k) L16[u][k] = tid + u * kSolveThreads < jb ? S[(size_t)(jb + k) * n_pad + tid + u * kSolveThreads] : 0.0;
    };
    auto request_block = [&](int jb) {   // wave 0: row rl of the diagonal block's L^T
        const int C = jb + rl;
#pragma unroll
        for (int k = 0; k < kNb; ++k) tt[k] = k > rl ? S[(size_t)C * n_pad + jb + k] : 0.0;
    };
    request_rows(n_pad - kNb, lr_next);
    if (wave == 0) request_block(n_pad - kNb);
    for (int jb = n_pad - kNb; jb >= 0; jb -= kNb) {
#pragma unroll
        for (int u = 0; u < kCols; ++u)
#pragma unroll
            for (int k = 0; k < kNb; ++k) lr[u][k] = lr_next[u][k];
        if (jb >= kNb) request_rows(jb - kNb, lr_next);
        if (wave == 0) {
            double r = vec[jb + rl];
            const double iv = invd[jb + rl];
#pragma unroll
            for (int k = kNb - 1; k >= 0; --k) {
                const double xk = lane_bcast(r * iv, k);   // lane k holds its finished residual
                if (rl == k) r = xk;
                else if (rl < k) r = __builtin_fma(-tt[k], xk, r);
            }
            if (lane < kNb) vec[jb + rl] = r;
            if (jb >= kNb) request_block(jb - kNb);   // in flight under the barrier and the column update
        }
        lds_barrier();
#pragma unroll
        for (int u = 0; u < kCols; ++u)
            if (tid + u * kSolveThreads < jb) {
                double v = vec[tid + u * kSolveThreads];
#pragma unroll
                for (int k = 0; k < kNb; ++k) v = __builtin_fma(-lr[u][k], vec[jb + k], v);
                vec[tid + u * kSolveThreads] = v;
            }
        lds_barrier();
    }
    for (int i = tid; i < n_pad; i += kSolveThreads) S[(size_t)n_pad * n_pad + i] = vec[i];
}


// S: the padded system above (lower triangle read). On return the rhs row holds x, S is overwritten (L on and below the diagonal, the
// diagonal blocks' L^T above it). *fail |= 2 when a pivot is not positive (the system is not positive definite; x is then
// unspecified). dbuf: the LDS holds two panels -- the trailing update then writes the next panel's columns straight into the other one.
// tstats: NULL, or 8 counters thread 0 adds its phase times to (wall_clock64 ticks of 10 ns; OVS_BA_TRACE with ovs_ba_dense_solve)
__global__ __launch_bounds__(kSolveThreads) void k_chol_solve(double* __restrict__ S, int n_pad, int dbuf, int32_t* __restrict__ fail,
                                                             unsigned long long* __restrict__ tstats) {
    extern __shared__ double lds[];
    const int panel_doubles = (n_pad + kNb) * kPitch;
    double* P = lds;                                                  // the panel: row r <-> logical row j0 + r
    double* Pn = lds + (dbuf ? panel_doubles : 0);                    // next panel (dbuf) or the same storage
    double* const vec = lds + (dbuf ? 2 : 1) * (size_t)panel_doubles; // n_pad: y, then x
    double* const invd = vec + n_pad;                                 // n_pad: 1 / L[c][c]
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kWaves = kSolveThreads / 64;
    const int n_rows = n_pad + kNb;
    const int rl = lane & 15, kq = lane >> 4;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < n_pad; i += kSolveThreads) vec[i] = 0.0;
    unsigned long long t_prev = tstats ? wall_clock64() : 0;
#define SOLVE_MARK(i)                                        \
    if (tstats && tid == 0) {                                \
        const unsigned long long t_now = wall_clock64();     \
        atomicAdd(&tstats[i], t_now - t_prev);               \
        t_prev = t_now;                                      \
    }

    for (int j0 = 0; j0 < n_pad; j0 += kNb) {
        const int m = n_rows - j0;        // staged rows
        const int mt = (m - kNb) / kNb;   // tile rows behind the panel; tile columns: mt - 1
        // ---- this wave's first kBatch tiles: their C values are requested now and arrive under the panel's factorisation
        const int n_tiles = (mt - 1) * (mt + 2) / 2;
        int tlin = wave;   // this wave's tiles: tlin, tlin + kWaves, ...
        // a batch = kBatch tile descriptors (ti | tc << 16, -1 = none; wave-uniform) and their C values (4 doubles per lane and tile).
        // Addresses are 32-bit byte offsets from S, rebuilt where they are used: kept as 64-bit pointers from the loads to the stores they
        // would cost as many registers as the values.
        int da[kBatch], db[kBatch];
        v4d accA[kBatch], accB[kBatch];
        const uint32_t row_step = 4u * (uint32_t)n_pad * 8u;   // bytes between the 4 rows of a lane's C values
        const char* const Sb = reinterpret_cast<const char*>(S);
        char* const Sw = reinterpret_cast<char*>(S);
        auto tile_off = [&](int d) -> uint32_t {   // byte offset of this lane's first C value of tile d
            const int i = d & 0xffff, c = d >> 16;
            return ((uint32_t)(j0 + kNb + kNb * i + kq) * (uint32_t)n_pad + (uint32_t)(j0 + kNb + kNb * c + rl)) * 8u;
        };
        auto gather = [&](int (&gd)[kBatch], v4d (&acc)[kBatch]) {
#pragma unroll
            for (int q = 0; q < kBatch; ++q) {
                gd[q] = -1;
                if (tlin < n_tiles) {
                    int ti, tc;
                    tile_seek(ti, tc, mt, tlin);
                    gd[q] = ti | (tc << 16);
                    const uint32_t o = tile_off(gd[q]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q][e] = *reinterpret_cast<const double*>(Sb + (o + (uint32_t)e * row_step));
                    tlin += kWaves;
                }
            }
        };
        gather(da, accA);
        if (j0 == 0 || !dbuf)
            for (int idx = tid; idx < m * kNb; idx += kSolveThreads) {
                const int r = idx >> 4, c = idx & 15;
                P[r * kPitch + c] = S[(size_t)(j0 + r) * n_pad + j0 + c];
            }
        lds_barrier();
        SOLVE_MARK(0)   // tile requests + panel load + barrier
        if (wave == 0) factor_diagonal_block(P, invd, j0, lane, &s_bad);
        lds_barrier();
        SOLVE_MARK(1)   // diagonal block + barrier
        if (s_bad) {
            if (tid == 0) atomicOr(fail, 2);
            return;
        }
        solve_rows(P, invd, vec, j0, m, n_pad, tid, lane);
        lds_barrier();
        SOLVE_MARK(2)   // row solves + barrier
        write_back_panel(S, P, j0, m, n_pad, tid);
        SOLVE_MARK(3)   // write-back (thread 0's share)
        // ---- trailing update C -= P_i P_c^T: kBatch tiles of a wave are computed while its next kBatch are on their way
        auto compute = [&](const int (&gd)[kBatch], v4d (&acc)[kBatch]) {
#pragma unroll
            for (int q = 0; q < kBatch; ++q)
                if (gd[q] >= 0) {
                    const double* pa = P + (kNb + kNb * (gd[q] & 0xffff) + rl) * kPitch + kq;
                    const double* pb = P + (kNb + kNb * (gd[q] >> 16) + rl) * kPitch + kq;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * s4], pb[4 * s4], acc[q], 0, 0, 0);
                }
#pragma unroll
            for (int q = 0; q < kBatch; ++q)
                if (gd[q] >= 0) {
                    const uint32_t o = tile_off(gd[q]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) *reinterpret_cast<double*>(Sw + (o + (uint32_t)e * row_step)) = acc[q][e];
                    if (dbuf && (gd[q] >> 16) == 0) {   // the next panel's columns
#pragma unroll
                        for (int e = 0; e < 4; ++e) Pn[(kNb * (gd[q] & 0xffff) + kq + 4 * e) * kPitch + rl] = acc[q][e];
                    }
                }
        };
        for (;;) {
            const bool more_b = tlin < n_tiles;
            if (more_b) gather(db, accB);
            compute(da, accA);
            if (!more_b) break;
            const bool more_a = tlin < n_tiles;
            if (more_a) gather(da, accA);
            compute(db, accB);
            if (!more_a) break;
        }
        SOLVE_MARK(4)   // trailing update, wave 0's tiles
        __syncthreads();
        SOLVE_MARK(5)   // ... waiting for the other waves
        if (dbuf) {
            double* const t = P;
            P = Pn;
            Pn = t;
        }
    }
    backward_substitution(S, vec, invd, n_pad, tid, lane, wave);
    SOLVE_MARK(6)   // backward substitution
#undef SOLVE_MARK
}
// ================================================================================================================================
// k_chol_resident (round 5): the same factorisation for systems up to 288 unknowns (BASELINE config 5: 48 free keyframes) with the TRAILING
// MATRIX RESIDENT IN REGISTERS. k_chol_solve above moves every trailing tile through one compute unit's memory path once per panel (8 MB read +
// written at n = 288: half of its 250 us, see the header); here a wave keeps its share of the 16 x 16 tiles as matrix-core accumulators for the
// whole kernel: tile (ti, tc), 1 <= tc <= ti, number t in column-major order, lives in wave t % 8, accumulator slot t / 8 (153 tiles at 18 tile
// rows: 20 slots = 160 registers). A tile leaves the registers once: after the update by panel tc - 1 it is written into the LDS panel buffer
// of panel tc, where it is factored / solved and from where the trailing updates of panel tc read their operands. Per panel: wave 0 factors
// the diagonal block (broadcasts by DPP row_newbcast inside the 16-lane row instead of v_readlane pairs through scalar registers: ~190
// instead of ~420 cycles per pivot) and forward-substitutes the right-hand side's block, one thread per row solves the rows below it (and
// takes the block out of its right-hand side entry: the rhs is a vector in LDS here, not a tile row), the finished panel goes to memory for
// the backward substitution, then every wave runs four v_mfma_f64_16x16x4_f64 per resident tile with A / B from the LDS panel.
// Same arithmetic per entry as k_chol_solve (fused multiply-adds, the matrix cores' order inside a 16 x 16 x 16 product), same storage, same
// result to rounding; identical bits from run to run.
// Measured at 288 unknowns (profiles/r05e_chol_phases.txt, r05e_lba_kernel_stats.txt): 168 us per solve against k_chol_solve's 252 -- loads 13,
// diagonal blocks 52, row solves 23, write-back 11, trailing update 47 (the f64 matrix pipe's own time is 29: 4400 instructions x 64 cycles over
// four SIMDs), backward substitution 33. Tried and dropped (profiles/r05k_chol_phases_lookahead_variant.txt): wave 0 factoring the NEXT diagonal
// block beside the other seven waves' trailing update, with 1 / d instead of 1 / sqrt(d) on the pivot chain -- 174 us: the pivot chain's f64
// operations slow down by half next to the matrix instructions of the wave that shares its SIMD (72 instead of 52 us), seven owners stretch the
// trailing update by 8 / 7, and the next panel's column as a phase of its own costs 1.7 us per panel.
constexpr int kResSlots = 18;      // accumulator slots per wave (144 registers: 19 already spill); tiles beyond 8 x kResSlots live in LDS (lane-major, 2 KB each)
constexpr int kResMaxPad = 288;
constexpr int kResMaxTiles = (kResMaxPad / 16) * (kResMaxPad / 16 - 1) / 2;   // 153

template <int N>
__device__ __forceinline__ double row_bcast_c(double v) {   // lane N of every 16-lane row -> all lanes of that row
    union {
        double d;
        int i[2];
    } u, w;
    u.d = v;
    w.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x150 + N, 0xf, 0xf, false);   // row_newbcast:N
    w.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x150 + N, 0xf, 0xf, false);
    return w.d;
}
__device__ __forceinline__ double row_bcast(double v, int n) {   // n is a constant after unrolling: the switch folds away
    switch (n) {
        case 0: return row_bcast_c<0>(v);
        case 1: return row_bcast_c<1>(v);
        case 2: return row_bcast_c<2>(v);
        case 3: return row_bcast_c<3>(v);
        case 4: return row_bcast_c<4>(v);
        case 5: return row_bcast_c<5>(v);
        case 6: return row_bcast_c<6>(v);
        case 7: return row_bcast_c<7>(v);
        case 8: return row_bcast_c<8>(v);
        case 9: return row_bcast_c<9>(v);
        case 10: return row_bcast_c<10>(v);
        case 11: return row_bcast_c<11>(v);
        case 12: return row_bcast_c<12>(v);
        case 13: return row_bcast_c<13>(v);
        case 14: return row_bcast_c<14>(v);
        default: return row_bcast_c<15>(v);
    }
}

// ---- round 6: the diagonal block's pivot chain, scheduled by hand ---------------------------------------------------------------------
// factor_block_dpp below compiles to a strictly serial stream: per pivot, ten dependent operations of the reciprocal square root (nothing
// between them), THEN the 15 - k column updates at five instructions per broadcast (two v_mov 0 for update_dpp's `old`, s_nop, two 32-bit DPP
// moves) -- ~430 cycles per pivot, 52 us of a 169 us solve. Here every instruction of the block is its own `asm volatile` (the compiler keeps
// their order, allocates the registers and sees none of the hazards, which are therefore padded inside the strings): a broadcast is ONE
// v_mov_b64_dpp row_newbcast (the only DPP control 64-bit operations have), column k + 1 is updated first so that the NEXT pivot's chain starts
// at once, and the remaining column updates and the right-hand side's forward-substitution step are issued BETWEEN the chain's dependent
// operations. Same operations on the same values as factor_block_dpp (the same bits); lanes r < k carry junk in a[k] and s, as there, and
// nothing reads it. 1 / L[k][k] and y_k are uniform after their broadcasts: every lane stores them (same address, same value).
#define OVS_BC64(N, NOP)                                                                                                                    \
    asm volatile(NOP "v_mov_b64_dpp %0, %1 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));                              \
    break;
template <bool kWait>   // kWait: the source was written by one of the two preceding vector instructions (DPP reads need two wait states)
__device__ __forceinline__ double bc64(double v, int n) {   // n is a constant after unrolling: the switch folds away
    double o;
    if (kWait) {
        switch (n) {
            case 0: OVS_BC64(0, "s_nop 1\n\t") case 1: OVS_BC64(1, "s_nop 1\n\t") case 2: OVS_BC64(2, "s_nop 1\n\t") case 3: OVS_BC64(3, "s_nop 1\n\t")
            case 4: OVS_BC64(4, "s_nop 1\n\t") case 5: OVS_BC64(5, "s_nop 1\n\t") case 6: OVS_BC64(6, "s_nop 1\n\t") case 7: OVS_BC64(7, "s_nop 1\n\t")
            case 8: OVS_BC64(8, "s_nop 1\n\t") case 9: OVS_BC64(9, "s_nop 1\n\t") case 10: OVS_BC64(10, "s_nop 1\n\t") case 11: OVS_BC64(11, "s_nop 1\n\t")
            case 12: OVS_BC64(12, "s_nop 1\n\t") case 13: OVS_BC64(13, "s_nop 1\n\t") case 14: OVS_BC64(14, "s_nop 1\n\t") default: OVS_BC64(15, "s_nop 1\n\t")
        }
    } else {
        switch (n) {
            case 0: OVS_BC64(0, "") case 1: OVS_BC64(1, "") case 2: OVS_BC64(2, "") case 3: OVS_BC64(3, "")
            case 4: OVS_BC64(4, "") case 5: OVS_BC64(5, "") case 6: OVS_BC64(6, "") case 7: OVS_BC64(7, "")
            case 8: OVS_BC64(8, "") case 9: OVS_BC64(9, "") case 10: OVS_BC64(10, "") case 11: OVS_BC64(11, "")
            case 12: OVS_BC64(12, "") case 13: OVS_BC64(13, "") case 14: OVS_BC64(14, "") default: OVS_BC64(15, "")
        }
    }
    return o;
}
#undef OVS_BC64
__device__ __forceinline__ double vmul64(double a, double b) {
    double d;
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ double vfma64(double a, double b, double c) {
    double d;
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ double vfnma64(double a, double b, double c) {   // c - a b, one rounding
    double d;
    asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ double vrsq64(double p) {   // (a transcendental result needs a wait state before a vector instruction reads it)
    double d;
    asm volatile("v_rsq_f64 %0, %1\n\ts_nop 0" : "=v"(d) : "v"(p));
    return d;
}
// one pivot step, K a template argument (a `#pragma unroll` over sixteen of these bodies exceeds the unroller's budget: it peeled five steps and
// left a loop with switch trees over the broadcast lane)
template <int K>
__device__ __forceinline__ void factor_steps(double (&a)[kNb], double& s, double y, bool& bad, double* __restrict__ invd_j, double* __restrict__ vec_j) {
    const double c15 = 1.5, cmh = -0.5;
    a[K] = vmul64(a[K], y);          // column K of L (lane K: piv * y = L[K][K])
    const double sy = vmul64(s, y);  // lane K: y_K
    invd_j[K] = y;
    double yn = y, h = 0.0;
    if constexpr (K + 1 < kNb) {   // column K + 1 first: the next pivot is final, its chain starts
        a[K + 1] = vfnma64(a[K], bc64<true>(a[K], K + 1), a[K + 1]);
        const double piv = bc64<true>(a[K + 1], K + 1);
        bad |= !(piv > 0.0 && piv < __builtin_inf());
        yn = vrsq64(piv);
        h = vmul64(piv, cmh);
    }
    const double yk = bc64<false>(sy, K);
    vec_j[K] = yk;
    s = vfnma64(a[K], yk, s);
    // the three Newton steps of the next pivot, a column update behind each of their dependent operations
#define OVS_COL(C)                                                                              \
    if constexpr ((C) < kNb) a[(C) < kNb ? (C) : 0] = vfnma64(a[K], bc64<false>(a[K], (C)), a[(C) < kNb ? (C) : 0]);
#define OVS_NEWTON(I)                                         \
    {                                                         \
        double t = 0.0, e = 0.0;                              \
        if constexpr (K + 1 < kNb) t = vmul64(h, yn);         \
        OVS_COL(K + 2 + 3 * (I))                              \
        if constexpr (K + 1 < kNb) e = vfma64(t, yn, c15);    \
        OVS_COL(K + 3 + 3 * (I))                              \
        if constexpr (K + 1 < kNb) yn = vmul64(yn, e);        \
        OVS_COL(K + 4 + 3 * (I))                              \
    }
    OVS_NEWTON(0)
    OVS_NEWTON(1)
    OVS_NEWTON(2)
    OVS_COL(K + 11) OVS_COL(K + 12) OVS_COL(K + 13) OVS_COL(K + 14) OVS_COL(K + 15)   // (the columns the nine slots did not take)
#undef OVS_NEWTON
#undef OVS_COL
    if constexpr (K + 1 < kNb) factor_steps<K + 1>(a, s, yn, bad, invd_j, vec_j);
}
__device__ __forceinline__ void factor_block_sched(double* __restrict__ P, double* __restrict__ invd, double* __restrict__ vec, int j0, int lane,
                                                   int* s_bad) {
    const int r = lane & 15;
    double a[kNb];
#pragma unroll
    for (int c = 0; c < kNb; ++c) a[c] = P[r * kPitch + c];
    double s = vec[j0 + r];
    const double piv = bc64<true>(a[0], 0);
    bool bad = !(piv > 0.0 && piv < __builtin_inf());
    double y = vrsq64(piv);
    {
        const double h = vmul64(piv, -0.5);
#pragma unroll
        for (int i = 0; i < 3; ++i) y = vmul64(y, vfma64(vmul64(h, y), y, 1.5));
    }
    factor_steps<0>(a, s, y, bad, invd + j0, vec + j0);
    if (lane < kNb) {
#pragma unroll
        for (int cc = 0; cc < kNb; ++cc)
            if (cc <= r) P[r * kPitch + cc] = a[cc];
        if (bad) *s_bad = 1;   // (lanes of one wave: the same value, any order)
    }
}

// wave 0: the diagonal block (P rows 0..15; lane r of every 16-lane row holds row r) and y = L_dd^-1 rhs_block (vec[j0 ..]); leaves L_dd in P,
// 1 / L[c][c] in invd
__device__ __forceinline__ void factor_block_dpp(double* __restrict__ P, double* __restrict__ invd, double* __restrict__ vec, int j0, int lane,
                                                 int* s_bad) {
    const int r = lane & 15;
    double a[kNb];
#pragma unroll
    for (int c = 0; c < kNb; ++c) a[c] = P[r * kPitch + c];
    double s = vec[j0 + r], my_inv = 0.0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < kNb; ++k) {
        const double piv = row_bcast(a[k], k);
        bad |= !(piv > 0.0 && piv < __builtin_inf());
        const double y = rsqrt_newton(piv);
        a[k] = (r == k) ? piv * y : a[k] * y;
        my_inv = (r == k) ? y : my_inv;
#pragma unroll
        for (int c = k + 1; c < kNb; ++c) a[c] = __builtin_fma(-a[k], row_bcast(a[k], c), a[c]);   // L[c][k] sits in lane c
    }
    // forward substitution of the block's right-hand side: lane r carries rhs_r - sum_{c < r} L[r][c] y_c
#pragma unroll
    for (int k = 0; k < kNb; ++k) {
        const double yk = row_bcast(s * my_inv, k);
        s = (r == k) ? yk : (r > k ? __builtin_fma(-a[k], yk, s) : s);
    }
    if (lane < kNb) {
#pragma unroll
        for (int c = 0; c < kNb; ++c)
            if (c <= r) P[r * kPitch + c] = a[c];
        invd[j0 + r] = my_inv;
        vec[j0 + r] = s;   // y
        if (bad) *s_bad = 1;   // (lanes of one wave: the same value, any order)
    }
}

// rows below the block, one thread per row: x L_dd^T = p, then the row's right-hand side entry loses x . y_block
__device__ __forceinline__ void solve_rows_res(double* __restrict__ P, const double* __restrict__ invd, double* __restrict__ vec, int j0, int m,
                                               int tid, int lane) {
    double lreg[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = lane + 64 * u;   // -> (c, k), k <= c: c = the largest integer with c (c + 1) / 2 <= t
        int c = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        c += ((c + 1) * (c + 2) / 2 <= t) ? 1 : 0;
        c -= (c * (c + 1) / 2 > t) ? 1 : 0;
        const int k = t - c * (c + 1) / 2;
        lreg[u] = t < kNb * (kNb + 1) / 2 ? P[c * kPitch + k] : 0.0;
    }
    double ivreg = invd[j0 + (lane & 15)], yreg = vec[j0 + (lane & 15)];
    for (int r = kNb + tid; r < m; r += kSolveThreads) {
        asm volatile("" : "+v"(lreg[0]), "+v"(lreg[1]), "+v"(lreg[2]), "+v"(ivreg), "+v"(yreg));   // broadcasts stay inside the row loop (scalar registers)
        double x[kNb];
#pragma unroll
        for (int c = 0; c < kNb; ++c) x[c] = P[r * kPitch + c];
        double dot = 0.0;
#pragma unroll
        for (int c = 0; c < kNb; ++c) {
            double v = x[c];
#pragma unroll
            for (int k = 0; k < c; ++k) {
                const int t = c * (c + 1) / 2 + k;
                v = __builtin_fma(-x[k], lane_bcast(lreg[t >> 6], t & 63), v);
            }
            x[c] = v * lane_bcast(ivreg, c);
            dot = __builtin_fma(x[c], lane_bcast(yreg, c), dot);
        }
#pragma unroll
        for (int c = 0; c < kNb; ++c) P[r * kPitch + c] = x[c];
        vec[j0 + r] -= dot;
    }
}

// Tiles are numbered from the BOTTOM-RIGHT corner: u = NT - 1 - ti, v = NT - 1 - tc (0 <= u <= v <= NT - 2), t = v (v + 1) / 2 + u. Then the
// tiles panel j still updates (tc > j) are the first jj (jj + 1) / 2 numbers, jj = NT - 1 - j, and the tiles beyond the register slots (kept
// in LDS) are those of the FIRST tile columns, which leave after one or two updates.
//
// One tile's update C -= A B^T as inline assembly with the accumulator TIED ("+v"): through the builtin, a matrix instruction under the
// (wave-uniform) "is this slot's tile still active" branch made hipcc keep the old and the new accumulator apart -- the untied three-address form,
// 8 more registers per slot, or whole second copies of the accumulator file across a switch -- and spill a third of the tiles. Wait states
// inside the string (nothing in it is padded by the compiler): the A operands were just written by the negating v_xor (s_nop before the first
// matrix instruction); the chain of four accumulates needs none; the result is read by compiler code afterwards (ds_write of a finished
// tile): a 16-pass f64 matrix instruction needs 18 states before a memory / LDS / VALU reader -- 24 are spent (s_nop 15 + s_nop 7 = 10 ns per tile).
__device__ __forceinline__ int res_active_tiles(int jj) { return jj * (jj + 1) / 2; }
__device__ __forceinline__ void tile_update_tied(v4d& c, double a0, double a1, double a2, double a3, double b0, double b1, double b2, double b3) {
    asm volatile(
        "s_nop 3\n"
        "v_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n"
        "v_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n"
        "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n"
        "v_mfma_f64_16x16x4_f64 %0, %4, %8, %0\n"
        "s_nop 15\n"
        "s_nop 7\n"
        : "+v"(c)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}

// Backward substitution of k_chol_resident (round 6). backward_substitution above asks for the L rows of block j - 1 while it works on block j:
// one step of ~0.3 us of arithmetic then waits out what is left of a ~1.8 us round trip (16 rows x n_pad columns through one CU) -- 18 steps,
// 33 us of a 169 us solve. Here the rows of TWO blocks ahead are in flight (three register buffers of 16 doubles, rotated; every load is
// unconditional -- clamped addresses -- so that the compiler can count them and wait for exactly the oldest buffer), and wave 0 takes the
// diagonal blocks' L from LDS, where the panel loop left a copy (DD: strictly lower triangle, zeros elsewhere), instead of their transposes
// from memory. The block solve keeps residuals in the lanes (lane k's residual is final at step k; zeros above the diagonal make the update a
// no-op for the finished lanes: no selects): multiply, one v_mov_b64_dpp, one fused multiply-add per step. Same operations on the same values
// as backward_substitution (the same bits).
__device__ __forceinline__ void backward_substitution_res(double* __restrict__ S, double* __restrict__ vec, const double* __restrict__ invd,
                                                          const double* __restrict__ DD, int n_pad, int tid, int lane, int wave) {
    const int rl = lane & 15;
    if (wave * 64 >= n_pad) {   // (wave-uniform) waves without a column keep the other waves' barriers company and load nothing
        for (int jb = n_pad - kNb; jb >= 0; jb -= kNb) {
            lds_barrier();
            lds_barrier();
        }
        return;
    }
    const int col = min(tid, n_pad - 1);
    auto request = [&](int jb, double (&X)[kNb]) {
        const double* base = S + (size_t)max(jb, 0) * n_pad + col;
#pragma unroll
        for (int k = 0; k < kNb; ++k) X[k] = base[(size_t)k * n_pad];
    };
    auto step = [&](int jb, const double (&X)[kNb]) {
        if (jb < 0) return;   // (uniform)
        if (wave == 0) {
            const double* dd = DD + (size_t)(jb >> 4) * (kNb * kPitch);
            double tt[kNb];
#pragma unroll
            for (int k = 0; k < kNb; ++k) tt[k] = dd[k * kPitch + rl];   // L[k][rl]: column rl of L = row rl of L^T (zero for k <= rl)
            double res = vec[jb + rl];
            const double iv = invd[jb + rl];
#pragma unroll
            for (int k = kNb - 1; k >= 0; --k) {
                const double xk = bc64<true>(vmul64(res, iv), k);
                res = vfnma64(tt[k], xk, res);
            }
            if (lane < kNb) vec[jb + rl] = vmul64(res, iv);
        }
        lds_barrier();
        if (tid < jb) {
            double v = vec[tid];
#pragma unroll
            for (int k = 0; k < kNb; ++k) v = __builtin_fma(-X[k], vec[jb + k], v);
            vec[tid] = v;
        }
        lds_barrier();
    };
    double A[kNb], B[kNb], C[kNb];
    int jb = n_pad - kNb;
    request(jb, A);
    request(jb - kNb, B);
    for (; jb >= 0; jb -= 3 * kNb) {
        request(jb - 2 * kNb, C);
        step(jb, A);
        request(jb - 3 * kNb, A);
        step(jb - kNb, B);
        request(jb - 4 * kNb, B);
        step(jb - 2 * kNb, C);
    }
    if (tid < n_pad) S[(size_t)n_pad * n_pad + tid] = vec[tid];   // (n_pad <= 288 < kSolveThreads)
}

// kTimed: the OVS_BA_TRACE build with phase marks (thread 0 adds its wall_clock64 intervals to tstats[0 .. 7]); the product instantiation carries
// none -- the marks' branches and 64-bit atomics inside the panel loop cost the register allocator 100 registers' worth of spills.
// A failed pivot does NOT leave the kernel early (a second exit from the panel loop had the same effect): the factorisation runs on with
// NaNs and the flag is raised at the end.
// kSched: the diagonal block by factor_block_sched (round 6, default) or by factor_block_dpp (OVS_CHOL_SCHED=0; same bits).
template <bool kTimed, bool kSched>
__global__ __launch_bounds__(kSolveThreads) void k_chol_resident(double* __restrict__ S, int n_pad, int32_t* __restrict__ fail,
                                                                unsigned long long* __restrict__ tstats) {
    extern __shared__ double lds[];
    const int panel_doubles = n_pad * kPitch;
    double* P = lds;                      // the panel: row r <-> logical row j0 + r
    double* Pn = lds + panel_doubles;     // the next panel's columns
    double* const vec = lds + 2 * (size_t)panel_doubles;   // n_pad: rhs, then y, then x
    double* const invd = vec + n_pad;                      // n_pad: 1 / L[c][c]
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kWaves = kSolveThreads / 64;
    const int rl = lane & 15, kq = lane >> 4;
    const int NT = n_pad / kNb;
    const int n_tiles = NT * (NT - 1) / 2;
    if (tid == 0) s_bad = 0;
    unsigned long long t_prev = kTimed ? wall_clock64() : 0;
#define SOLVE_MARK(i)                                        \
    if (kTimed && tid == 0) {                                \
        const unsigned long long t_now = wall_clock64();     \
        atomicAdd(&tstats[i], t_now - t_prev);               \
        t_prev = t_now;                                      \
    }
    // ---- tile t <-> (ti, tc): a table in LDS (ti | tc << 8), built once
    __shared__ int s_td[kResMaxTiles];
    for (int t = tid; t < n_tiles; t += kSolveThreads) {
        int v = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        v += ((v + 1) * (v + 2) / 2 <= t) ? 1 : 0;
        v -= (v * (v + 1) / 2 > t) ? 1 : 0;
        const int u = t - v * (v + 1) / 2;
        s_td[t] = (NT - 1 - u) | ((NT - 1 - v) << 8);
    }
    if (tid == 0 && n_tiles == 0) s_td[0] = 0;   // (one tile row: the slots' stand-in loads below read tile (0, 0), i.e. valid memory)
    __syncthreads();
    // ---- this wave's tiles: slots 0 .. kResSlots - 1 in registers, the rest (tiles 8 kResSlots + u, u = wave, wave + 8, ...: the first tile
    //      columns, which leave after one or two updates) in LDS behind the panels, lane-major: value e of lane l of tile u at T[(u * 4 + e) * 64 + l]
    double* const T = invd + n_pad;
    const int n_lds = n_tiles > kWaves * kResSlots ? n_tiles - kWaves * kResSlots : 0;
    double* const DD = T + (size_t)n_lds * 256;   // kSched: the diagonal blocks' L (NT x 16 x kPitch doubles) for the backward substitution
    int sd[kResSlots];
    v4d acc[kResSlots];
    const int lane_off = kq * n_pad + rl;
#pragma unroll
    for (int s = 0; s < kResSlots; ++s) {
        const int t = kWaves * s + wave;
        // (a slot without a tile -- small systems -- loads tile 0's values, which nothing reads: its number is never below the active count)
        sd[s] = __builtin_amdgcn_readfirstlane(s_td[t < n_tiles ? t : 0]);
        // uniform tile origin + the lane's own 32-bit offset (the same four offsets for every tile): scalar-base loads
        const double* base = S + (size_t)(kNb * (sd[s] & 0xff)) * n_pad + kNb * (sd[s] >> 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[s][e] = base[lane_off + e * 4 * n_pad];
    }
    for (int u = wave; u < n_lds; u += kWaves) {
        const int d = __builtin_amdgcn_readfirstlane(s_td[kWaves * kResSlots + u]);
        const double* src = S + (size_t)(kNb * (d & 0xff) + kq) * n_pad + kNb * (d >> 8) + rl;
#pragma unroll
        for (int e = 0; e < 4; ++e) T[(u * 4 + e) * 64 + lane] = src[(size_t)(4 * e) * n_pad];
    }
    for (int i = tid; i < n_pad; i += kSolveThreads) vec[i] = S[(size_t)n_pad * n_pad + i];
    for (int idx = tid; idx < n_pad * kNb; idx += kSolveThreads) {
        const int r = idx >> 4, c = idx & 15;
        P[r * kPitch + c] = S[(size_t)r * n_pad + c];
    }
    __syncthreads();
    SOLVE_MARK(0)   // loads
    for (int j = 0; j < NT; ++j) {
        const int j0 = j * kNb, m = n_pad - j0;
        if (wave == 0) {
            if (kSched) factor_block_sched(P, invd, vec, j0, lane, &s_bad);
            else factor_block_dpp(P, invd, vec, j0, lane, &s_bad);
        }
        lds_barrier();
        SOLVE_MARK(1)   // diagonal block + forward substitution of its rhs + barrier
        if (kSched && tid >= kSolveThreads - kNb * kNb) {
            // the block's copy for the backward substitution (strictly lower triangle, zeros elsewhere), made by the upper half of the workgroup, which
            // owns few or no rows below a block (n_pad <= 288: rows 16 .. 287 go to threads 0 .. 271) -- off wave 0's critical path
            const int e = tid - (kSolveThreads - kNb * kNb), r = e >> 4, c = e & 15;
            DD[(size_t)j * (kNb * kPitch) + r * kPitch + c] = c < r ? P[r * kPitch + c] : 0.0;
        }
        solve_rows_res(P, invd, vec, j0, m, tid, lane);
        lds_barrier();
        SOLVE_MARK(2)   // row solves + barrier
        write_back_panel(S, P, j0, m, n_pad, tid);
        SOLVE_MARK(3)   // write-back (thread 0's share)
        // ---- trailing update C -= P_ti P_tc^T of the resident tiles; the tiles of the next panel's column move to its LDS buffer
        const int jj = NT - 1 - j;
#pragma unroll
        for (int s = 0; s < kResSlots; ++s) {
            if (kWaves * s + wave < res_active_tiles(jj)) {   // (wave-uniform)
                const int ti = sd[s] & 0xff, tc = sd[s] >> 8;
                const double* pa = P + (kNb * (ti - j) + rl) * kPitch + kq;
                const double* pb = P + (kNb * (tc - j) + rl) * kPitch + kq;
                tile_update_tied(acc[s], -pa[0], -pa[4], -pa[8], -pa[12], pb[0], pb[4], pb[8], pb[12]);
                if (tc == j + 1) {   // the next panel's column: the tile is final, it moves to that panel's buffer
#pragma unroll
                    for (int e = 0; e < 4; ++e) Pn[(kNb * (ti - j - 1) + kq + 4 * e) * kPitch + rl] = acc[s][e];
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // one tile's operands at a time: the scheduler otherwise hoists the LDS reads of all slots (16 registers each)
        }
        const int n_act = res_active_tiles(jj);
        for (int u = wave; u < n_lds; u += kWaves) {   // the tiles that live in LDS
            if (kWaves * kResSlots + u >= n_act) break;
            const int d = __builtin_amdgcn_readfirstlane(s_td[kWaves * kResSlots + u]);
            const int ti = d & 0xff, tc = d >> 8;
            v4d c4;
#pragma unroll
            for (int e = 0; e < 4; ++e) c4[e] = T[(u * 4 + e) * 64 + lane];
            const double* pa = P + (kNb * (ti - j) + rl) * kPitch + kq;
            const double* pb = P + (kNb * (tc - j) + rl) * kPitch + kq;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * s4], pb[4 * s4], c4, 0, 0, 0);
            if (tc == j + 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Pn[(kNb * (ti - j - 1) + kq + 4 * e) * kPitch + rl] = c4[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) T[(u * 4 + e) * 64 + lane] = c4[e];
            }
        }
        SOLVE_MARK(4)   // trailing update, wave 0's tiles
        lds_barrier();
        SOLVE_MARK(5)   // ... waiting for the other waves
        double* const t = P;
        P = Pn;
        Pn = t;
    }
    __syncthreads();   // the panels written back above are read by other threads below
    if (kSched) backward_substitution_res(S, vec, invd, DD, n_pad, tid, lane, wave);
    else backward_substitution(S, vec, invd, n_pad, tid, lane, wave);
    SOLVE_MARK(6)   // backward substitution
#undef SOLVE_MARK
    if (tid == 0 && s_bad) atomicOr(fail, 2);   // (s_bad was last written before the panel loop's barriers)
}
static_assert(kMaxN % kSolveThreads == 0, "columns per thread in the backward substitution");

// largest system the one-workgroup solver stages: (n_pad + 16) x 17 + 2 n_pad doubles of LDS (and one thread per unknown)
int dense_solve_max_n() { return kMaxN; }

int dense_solve_pad(int n) { return (n + kNb - 1) / kNb * kNb; }
size_t dense_solve_doubles(int n) { return (size_t)(dense_solve_pad(n) + kNb) * dense_solve_pad(n); }   // the padded system's storage

// d_S: the padded system of n unknowns (see above); the solution replaces the right-hand side at d_S + n_pad * n_pad.
// CONTRACT on failure: when the factorisation meets a pivot that is not positive (or not finite) the kernel ORs 2 into *d_fail and -- k_chol_resident
// does not leave early: its waves run on in lock step -- the matrix, its padding and the solution are then GARBAGE (NaNs). A caller that reuses d_S
// must rebuild the whole padded system (ba_graph_reset_system does, for ovs_local_ba_optimize) and must not consume the solution; kernels queued
// behind the solve in the same trial (k_trial_update, the linearisation at the trial state) run on NaN states whose outputs the caller discards
// when it reads the failure word. tests/test_gpu_ba.py::test_dense_solve_failure_then_success covers both solver kernels.
ovs_status launch_dense_solve(double* d_S, int n, int32_t* d_fail, hipStream_t s, unsigned long long* d_tstats) {
    if (n < 1 || n > dense_solve_max_n()) return OVS_ERR_INVALID;
    const int n_pad = dense_solve_pad(n);
    if (n_pad <= kResMaxPad && tuning().chol_resident) {   // the trailing matrix fits the register file: k_chol_resident
        const int nt = n_pad / kNb, n_tiles = nt * (nt - 1) / 2, n_lds = std::max(0, n_tiles - (kSolveThreads / 64) * kResSlots);
        const size_t lds = sizeof(double) * (2 * (size_t)n_pad * kPitch + 2 * (size_t)n_pad + (size_t)n_lds * 256 + (size_t)nt * kNb * kPitch);
        static LdsAttrCache cache[4];
        static const bool sched = [] {
            const char* e = std::getenv("OVS_CHOL_SCHED");
            return !(e && e[0] == '0');
        }();
        const int which = (d_tstats ? 2 : 0) + (sched ? 1 : 0);
        const void* const fns[4] = {reinterpret_cast<const void*>(k_chol_resident<false, false>), reinterpret_cast<const void*>(k_chol_resident<false, true>),
                                    reinterpret_cast<const void*>(k_chol_resident<true, false>), reinterpret_cast<const void*>(k_chol_resident<true, true>)};
        hipError_t e = ensure_dynamic_lds(fns[which], lds, cache[which]);
        if (e != hipSuccess) {
            set_last_error("hipFuncSetAttribute(k_chol_resident)", e);
            return OVS_ERR_HIP;
        }
        switch (which) {
            case 0: hipLaunchKernelGGL((k_chol_resident<false, false>), dim3(1), dim3(kSolveThreads), lds, s, d_S, n_pad, d_fail, d_tstats); break;
            case 1: hipLaunchKernelGGL((k_chol_resident<false, true>), dim3(1), dim3(kSolveThreads), lds, s, d_S, n_pad, d_fail, d_tstats); break;
            case 2: hipLaunchKernelGGL((k_chol_resident<true, false>), dim3(1), dim3(kSolveThreads), lds, s, d_S, n_pad, d_fail, d_tstats); break;
            default: hipLaunchKernelGGL((k_chol_resident<true, true>), dim3(1), dim3(kSolveThreads), lds, s, d_S, n_pad, d_fail, d_tstats); break;
        }
        OVS_LAUNCH_TRY("k_chol_resident");
        return OVS_OK;
    }
    const size_t panel = sizeof(double) * (size_t)(n_pad + kNb) * kPitch, rest = sizeof(double) * 2 * (size_t)n_pad;
    const int dbuf = 2 * panel + rest <= (size_t)150 * 1024 ? 1 : 0;   // two panels in LDS up to 528 unknowns
    const size_t lds = (dbuf ? 2 : 1) * panel + rest;
    static LdsAttrCache cache;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_chol_solve), lds, cache);
    if (e != hipSuccess) {
        set_last_error("hipFuncSetAttribute(k_chol_solve)", e);
        return OVS_ERR_HIP;
    }
    hipLaunchKernelGGL(k_chol_solve, dim3(1), dim3(kSolveThreads), lds, s, d_S, n_pad, dbuf, d_fail, d_tstats);
    OVS_LAUNCH_TRY("k_chol_solve");
    return OVS_OK;
}

}   // namespace ovs

extern "C" {

// the solver alone, on host arrays (a test / debugging entry: tests/test_gpu_ba.py compares it with numpy's solve)
ovs_status ovs_ba_dense_solve(int32_t device, const double* S, const double* rhs, int32_t n, double* x) {
    if (!S || !rhs || !x || n < 1 || n > ovs::dense_solve_max_n()) return OVS_ERR_INVALID;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    OVS_HIP_TRY(hipSetDevice(device));
    const int n_pad = ovs::dense_solve_pad(n);
    std::vector<double> h(ovs::dense_solve_doubles(n), 0.0);
    for (int i = 0; i < n; ++i) std::memcpy(&h[(size_t)i * n_pad], S + (size_t)i * n, sizeof(double) * n);
    for (int i = n; i < n_pad; ++i) h[(size_t)i * n_pad + i] = 1.0;
    std::memcpy(&h[(size_t)n_pad * n_pad], rhs, sizeof(double) * n);
    double* d = nullptr;
    const size_t bytes = sizeof(double) * h.size();
    OVS_HIP_TRY(hipMalloc(&d, bytes + 256));
    int32_t* d_fail = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(d) + bytes);
    unsigned long long* d_t = ovs::tuning().ba_trace ? reinterpret_cast<unsigned long long*>(d_fail + 2) : nullptr;   // 8 counters behind the flag
    ovs_status st = OVS_OK;
    int32_t h_fail = 0;
    hipError_t e = hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_fail, 0, 128);
    if (e == hipSuccess) {
        st = ovs::launch_dense_solve(d, n, d_fail, nullptr, d_t);
        if (st == OVS_OK) e = hipDeviceSynchronize();
    }
    if (d_t && e == hipSuccess && st == OVS_OK) {
        unsigned long long h_t[8] = {};
        (void)hipMemcpy(h_t, d_t, sizeof(h_t), hipMemcpyDeviceToHost);
        static const char* nm[7] = {"requests+panel load", "diagonal block", "row solves", "write-back", "trailing (wave 0)", "trailing (others)", "backward"};
        std::fprintf(stderr, "[dense solve n=%d]", n);
        for (int i = 0; i < 7; ++i) std::fprintf(stderr, " %s %.1f us,", nm[i], h_t[i] * 0.01);
        std::fprintf(stderr, "\n");
    }
    if (e == hipSuccess && st == OVS_OK) e = hipMemcpy(x, d + (size_t)n_pad * n_pad, sizeof(double) * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess && st == OVS_OK) e = hipMemcpy(&h_fail, d_fail, sizeof(int32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) {
        ovs::set_last_error("ovs_ba_dense_solve", e);
        return OVS_ERR_HIP;
    }
    if (st != OVS_OK) return st;
    if (h_fail) {
        ovs::set_last_error_text("ovs_ba_dense_solve: the matrix is not positive definite");
        return OVS_ERR_INVALID;
    }
    return OVS_OK;
}

}   // extern "C"
