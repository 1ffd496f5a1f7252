// match_window.hip -- D1 + M3 + M5 + M7: the grid candidate generator and the windowed matchers built on it
// (expected: src/openvslam/data/common.{h,cc}; src/openvslam/match/{projection,area,bow_tree}.{h,cc}, angle_checker.h).
//
// These are small, latency-bound problems (a 3840x1920 frame with 4000 keypoints and 10 000 landmarks is ~3 Hamming distances
// per landmark); what the device buys is that extract -> grid -> match never leaves HBM. Three stages, all exact:
//   1. k_grid_assign   assign_keypoints_to_grid as a CSR (cell id = cx*rows + cy, members ascending = upstream's push_back order):
//                      one workgroup, LDS histogram + scan + fill + per-cell sort.
//   2. k_window_*/k_bow_*  get_keypoints_in_cell + compute_descriptor_distance_32 for every query (one lane per query, count pass,
//                      scan, fill pass): a CSR of keys  d << 20 | octave << 16 | target  in UPSTREAM'S ENUMERATION ORDER
//                      (cells x-major, then y, then members), because strict `<` keeps the first minimum.
//   3. k_list_resolve  upstream's loops are sequential -- a query sees the targets earlier queries claimed (M3, M7) or the
//                      distance they were matched at (M5). A workgroup replays them 64 * NW queries per round: every pending thread
//                      evaluates against the current state and stamps every live candidate of its list; a thread is affected if a
//                      LOWER thread stamped the best / second candidate its decision rests on; every unaffected thread commits (the
//                      rule is spelled out at the kernel). State per target is ONE number, thr[t]: a candidate at distance d is
//                      alive iff d < thr[t] (256 = free, 0 = claimed, M5: the distance it is currently matched at), and thr only
//                      ever decreases, so a decision that rests on (best, second) can only be changed by a lower thread taking
//                      one of those two.
//      The rotation-histogram check (match::angle_checker, 30 bins, keep the 3 fullest) runs in the same kernel.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "ovs_common.h"

namespace ovs {

struct GridP {
    float min_x, min_y;
    float inv_w, inv_h;
    int32_t cols, rows;
};

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kMarkSize = 4096;   // hashed stamp table of the resolver (collisions only cost an extra round)

__device__ __forceinline__ uint32_t hamming256_g(const uint32_t (&a)[8], const uint32_t* __restrict__ b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d = __builtin_popcount(a[i] ^ b[i]) + d;
    return d;
}

// ---- 1. D1 assign_keypoints_to_grid ---------------------------------------------------------------------------------
// Round 6: the scan of the cell histogram by wave shuffles (two barriers; the 1024-wide Hillis-Steele loop had twenty) and, when the keypoints fit
// (items_in_lds: n <= kGridItemsLds), the member lists built and sorted in LDS and written once, coalesced -- the per-cell insertion sort used to walk
// `items` in global memory, a chain of dependent round trips per cell: 17 -> ~8 us per 2000-keypoint frame, once per tracked frame.
constexpr int kGridItemsLds = 8192;
__global__ __launch_bounds__(1024) void k_grid_assign(const ovs_keypoint* __restrict__ kps, int n, GridP gp, int32_t* __restrict__ cell_of,
                                                     int32_t* __restrict__ cell_start, int32_t* __restrict__ items, int items_in_lds) {
    extern __shared__ int32_t s_grid[];
    const int nc = gp.cols * gp.rows;
    int32_t* cnt = s_grid;            // [nc] histogram, then fill cursor
    int32_t* start = s_grid + nc;     // [nc + 1]
    int32_t* const lit = items_in_lds ? s_grid + 2 * nc + 1 : items;   // [n] the member lists while they are built and sorted
    __shared__ int32_t s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int c = tid; c < nc; c += 1024) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        // get_cell_indices: cvRound((pt - min) * inv_cell)
        const int cx = __float2int_rn(__fmul_rn(__fsub_rn(kps[i].x, gp.min_x), gp.inv_w));
        const int cy = __float2int_rn(__fmul_rn(__fsub_rn(kps[i].y, gp.min_y), gp.inv_h));
        int c = -1;
        if (0 <= cx && cx < gp.cols && 0 <= cy && cy < gp.rows) {
            c = cx * gp.rows + cy;
            atomicAdd(&cnt[c], 1);
        }
        cell_of[i] = c;
    }
    __syncthreads();
    // exclusive scan: thread t owns cells [t*per, t*per + per)
    const int per = (nc + 1023) / 1024;
    int local = 0;
    for (int k = 0; k < per; ++k) {
        const int c = tid * per + k;
        if (c < nc) local += cnt[c];
    }
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    int run = incl - local, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int t = s_w[w];
        run += w < wv ? t : 0;
        total += t;
    }
    for (int k = 0; k < per; ++k) {
        const int c = tid * per + k;
        if (c < nc) {
            start[c] = run;
            run += cnt[c];
        }
    }
    if (tid == 0) start[nc] = total;
    __syncthreads();
    for (int c = tid; c <= nc; c += 1024) {
        cell_start[c] = start[c];
        if (c < nc) cnt[c] = start[c];
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int c = cell_of[i];   // (this thread's own store above)
        if (c >= 0) lit[atomicAdd(&cnt[c], 1)] = i;
    }
    __syncthreads();   // also makes this workgroup's global writes visible to itself
    // members in ascending keypoint index (upstream pushes them in keypoint order)
    for (int c = tid; c < nc; c += 1024) {
        const int b = start[c], e = start[c + 1];
        for (int i = b + 1; i < e; ++i) {
            const int v = lit[i];
            int j = i - 1;
            while (j >= b && lit[j] > v) {
                lit[j + 1] = lit[j];
                --j;
            }
            lit[j + 1] = v;
        }
    }
    if (items_in_lds) {
        __syncthreads();
        for (int i = tid; i < total; i += 1024) items[i] = lit[i];
    }
}

// ---- 2. candidate lists ---------------------------------------------------------------------------------------------------
enum { kModeProjection = 0, kModeArea = 1, kModeGeneric = 2 };

struct WinArgs {
    // targets: the frame whose grid is searched
    const ovs_keypoint* t_kps;
    const uint8_t* t_desc;
    const uint8_t* t_occupied;   // projection: keypoint already holds a landmark with observations (NULL = none)
    const float* t_x_right;      // projection: frm.stereo_x_right_ (NULL = monocular)
    const int32_t* cell_start;
    const int32_t* items;
    GridP gp;
    // queries
    int n_q;
    const float* q_xy;           // projection: reproj_in_tracking_; area: prev_matched_pts
    const float* q_x_right;      // projection: x_right_in_tracking_
    const int32_t* q_level;      // projection: scale_level_in_tracking_
    const uint8_t* q_valid;      // projection: is_observable_in_tracking_ && !will_be_erased()
    const ovs_keypoint* q_kps;   // area: frame-1 undistorted keypoints
    const float* q_radius;       // generic: search radius, level window (filled by k_reproject_queries)
    const int32_t* q_minl;
    const int32_t* q_maxl;
    const uint8_t* q_desc;
    float margin;
    float sf[OVS_MAX_LEVELS];
    int mode;
    // Candidates whose Hamming distance can neither win (d > the rule's acceptance threshold) nor make a ratio test fail (fl(d * ratio) >= that
    // threshold) are left out of the lists altogether: every candidate with d >= dead_from (0: no filter; set per rule by dead_from_*() below).
    // The sequential rule decides the same with or without them (a dropped best is a reject either way, a dropped second accepts either way, and
    // so does every later second), but the resolver's rounds are bounded by the chains of queries sharing LIVE candidates and its lists only fit
    // LDS when they are short: with a 100-px margin (BASELINE config 0) a query has ~110 candidates of which ~1 is alive.
    uint32_t dead_from;
};

// One WAVE per query: the lanes take the cells of the query's window (a 100-px margin covers ~340 of the 64 x 48 cells -- a single
// lane walking them was ~0.7 ms of dependent loads per pass), count their members that pass the filters, and an exclusive scan over
// the lanes places every cell's members at their position in UPSTREAM'S ORDER (cells x-major, then y, then members).
template <bool FILL>
__global__ __launch_bounds__(256) void k_window_lists(WinArgs a, uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                     uint32_t* __restrict__ keys, uint32_t key_cap, uint32_t* __restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= a.n_q) return;   // wave-uniform
    bool valid = !a.q_valid || a.q_valid[q];
    float r;
    int minl, maxl;
    if (a.mode == kModeProjection) {
        const int lvl = valid ? a.q_level[q] : 0;
        r = __fmul_rn(a.margin, a.sf[lvl]);
        minl = lvl - 1;
        maxl = lvl;
    } else if (a.mode == kModeGeneric) {
        r = a.q_radius[q];
        minl = a.q_minl[q];
        maxl = a.q_maxl[q];
    } else {
        const int lvl = a.q_kps[q].octave;
        if (0 < lvl) valid = false;   // "only level-0 keypoints"
        r = a.margin;
        minl = maxl = lvl;
    }
    uint32_t total = 0;
    if (valid) {
        const uint32_t base = FILL ? offsets[q] : 0u;
        const float ref_x = a.q_xy[2 * q], ref_y = a.q_xy[2 * q + 1];
        const float q_xr = (a.mode != kModeArea && a.t_x_right) ? a.q_x_right[q] : 0.0f;
        uint32_t qd[8];
        if (FILL || a.dead_from) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(a.q_desc + (size_t)q * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) qd[i] = src[i];
        }
        const GridP& g = a.gp;
        // get_keypoints_in_cell
        const int min_cx = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(ref_x, g.min_x), r), g.inv_w)));
        const int max_cx = min(g.cols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(ref_x, g.min_x), r), g.inv_w)));
        const int min_cy = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(ref_y, g.min_y), r), g.inv_h)));
        const int max_cy = min(g.rows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(ref_y, g.min_y), r), g.inv_h)));
        if (min_cx < g.cols && max_cx >= 0 && min_cy < g.rows && max_cy >= 0) {
            const bool check_level = (0 < minl) || (0 <= maxl);
            const int ncy = max_cy - min_cy + 1, ncell = (max_cx - min_cx + 1) * ncy;
            auto passes = [&](int idx, const ovs_keypoint& kp) -> bool {
                if (check_level && (kp.octave < minl || (0 <= maxl && maxl < kp.octave))) return false;
                if (!(fabsf(__fsub_rn(kp.x, ref_x)) < r && fabsf(__fsub_rn(kp.y, ref_y)) < r)) return false;
                if (a.mode != kModeArea) {
                    if (a.t_occupied && a.t_occupied[idx]) return false;
                    if (a.t_x_right) {
                        const float xr = a.t_x_right[idx];
                        if (0 < xr && r < fabsf(__fsub_rn(q_xr, xr))) return false;
                    }
                }
                return true;
            };
            for (int c0 = 0; c0 < ncell; c0 += 64) {
                const int ci = c0 + lane;
                int b = 0, e = 0;
                if (ci < ncell) {
                    const int cxo = ci / ncy;
                    const int c = (min_cx + cxo) * g.rows + min_cy + (ci - cxo * ncy);
                    b = a.cell_start[c];
                    e = a.cell_start[c + 1];
                }
                uint32_t n_pass = 0;
                for (int k = b; k < e; ++k) {
                    const int idx = a.items[k];
                    bool ok = passes(idx, a.t_kps[idx]);
                    if (ok && a.dead_from) ok = hamming256_g(qd, reinterpret_cast<const uint32_t*>(a.t_desc + (size_t)idx * 32)) < a.dead_from;
                    n_pass += ok ? 1u : 0u;
                }
                uint32_t incl = n_pass;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t t = __shfl_up(incl, off);
                    if (lane >= off) incl += t;
                }
                if (FILL && n_pass) {
                    uint32_t pos = base + total + (incl - n_pass);
                    for (int k = b; k < e; ++k) {
                        const int idx = a.items[k];
                        const ovs_keypoint kp = a.t_kps[idx];
                        if (!passes(idx, kp)) continue;
                        const uint32_t d = hamming256_g(qd, reinterpret_cast<const uint32_t*>(a.t_desc + (size_t)idx * 32));
                        if (a.dead_from && !(d < a.dead_from)) continue;
                        if (pos < key_cap) keys[pos] = (d << 20) | ((uint32_t)(kp.octave & 15) << 16) | (uint32_t)idx;
                        else *overflow = 1u;
                        ++pos;
                    }
                }
                total += __shfl(incl, 63);
            }
        }
    }
    if (!FILL && lane == 0) counts[q] = total;
}

// camera::perspective / camera::equirectangular ::reproject_to_image in double precision, one rounding per operation (the library
// is built with -ffp-contract=off), same operation order as the oracle.
struct CamP {
    int32_t model, setup;
    double fx, fy, cx, cy, fxb;
    int32_t cols, rows;
    float min_x, min_y, max_x, max_y;
    double P[12];   // rot_cw row-major, trans_cw
};

__device__ __forceinline__ bool reproject_to_image(const CamP& c, const double* __restrict__ X, double& u, double& v, float& x_right) {
    const double pcx = (c.P[0] * X[0] + c.P[1] * X[1]) + c.P[2] * X[2] + c.P[9];
    const double pcy = (c.P[3] * X[0] + c.P[4] * X[1]) + c.P[5] * X[2] + c.P[10];
    const double pcz = (c.P[6] * X[0] + c.P[7] * X[1]) + c.P[8] * X[2] + c.P[11];
    if (c.model == 0) {
        if (pcz <= 0.0) return false;
        const double z_inv = 1.0 / pcz;
        u = c.fx * pcx * z_inv + c.cx;
        v = c.fy * pcy * z_inv + c.cy;
        x_right = (float)(u - c.fxb * z_inv);
        if (u < c.min_x || u > c.max_x) return false;
        if (v < c.min_y || v > c.max_y) return false;
        return true;
    }
    const double norm = sqrt((pcx * pcx + pcy * pcy) + pcz * pcz);
    const double bx = pcx / norm, by = pcy / norm, bz = pcz / norm;
    const double latitude = -ovs_det_asin(by);
    const double longitude = ovs_det_atan2(bx, bz);
    u = c.cols * (0.5 + longitude / (2.0 * 3.14159265358979323846));
    v = c.rows * (0.5 - latitude / 3.14159265358979323846);
    x_right = -1.0f;
    return true;
}

// match_current_and_last_frames: one lane per last-frame keypoint -> query (reprojection, radius, level window, validity)
__global__ __launch_bounds__(256) void k_reproject_queries(CamP cam, const ovs_keypoint* __restrict__ last_kps,
                                                          const double* __restrict__ pos_w, const uint8_t* __restrict__ last_valid, int n,
                                                          float margin, const float* __restrict__ sf_dev, int num_levels, int forward,
                                                          int backward, const float* __restrict__ dist_min_max, double ccx, double ccy,
                                                          double ccz, float log_scale_factor, float* __restrict__ q_xy,
                                                          float* __restrict__ q_x_right, float* __restrict__ q_radius,
                                                          int32_t* __restrict__ q_minl, int32_t* __restrict__ q_maxl,
                                                          uint8_t* __restrict__ q_valid, const double* __restrict__ normals = nullptr,
                                                          int level_up = 1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool valid = !last_valid || last_valid[i];
    double u = 0, v = 0;
    float xr = -1.0f;
    const double* X = pos_w + 3 * (size_t)i;
    if (valid) valid = reproject_to_image(cam, X, u, v, xr);
    int lvl, minl, maxl;
    if (dist_min_max) {
        // match_frame_and_keyframe: valid distance range, level from the distance (landmark::predict_scale_level), window [pred-1, pred+1]
        const double dx = X[0] - ccx, dy = X[1] - ccy, dz = X[2] - ccz;
        const double dist = sqrt((dx * dx + dy * dy) + dz * dz);
        // landmark::get_min_valid_distance() = 0.7 * min_valid_dist_, get_max_valid_distance() = 1.3 * max_valid_dist_ (double product
        // returned as float) gate the range; predict_scale_level uses the RAW max_valid_dist_
        const float dmin = dist_min_max[2 * i], dmax = dist_min_max[2 * i + 1];
        if (dist < (double)(float)(0.7 * (double)dmin) || (double)(float)(1.3 * (double)dmax) < dist) valid = false;
        if (normals) {   // match_by_Sim3_transform: viewing-angle gate against the landmark's mean normal
            const double* nrm = normals + 3 * (size_t)i;
            if ((dx * nrm[0] + dy * nrm[1]) + dz * nrm[2] < 0.5 * dist) valid = false;
        }
        lvl = valid ? (int)ceilf(__fdiv_rn(ovs_det_logf(__fdiv_rn(dmax, (float)dist)), log_scale_factor)) : 0;
        if (lvl < 0) lvl = 0;
        else if (num_levels <= lvl) lvl = num_levels - 1;
        minl = lvl - 1;
        maxl = lvl + level_up;
    } else {
        lvl = last_kps[i].octave;
        minl = forward ? lvl : (backward ? 0 : lvl - 1);
        maxl = forward ? num_levels - 1 : (backward ? lvl : lvl + 1);
    }
    q_xy[2 * i] = (float)u;
    q_xy[2 * i + 1] = (float)v;
    q_x_right[i] = xr;
    q_radius[i] = __fmul_rn(margin, sf_dev[lvl]);
    q_minl[i] = minl;
    q_maxl[i] = maxl;
    q_valid[i] = valid ? 1 : 0;
}

// fuse::replace_duplication, candidate search: landmarks are independent (the write-back that follows is host-side graph surgery)
struct FuseArgs {
    const ovs_keypoint* t_kps;
    const uint8_t* t_desc;
    const float* t_x_right;
    const int32_t* cell_start;
    const int32_t* items;
    GridP gp;
    CamP cam;
    double cc[3];                 // camera centre
    const double* lm_pos_w;
    const float* lm_dist;         // (min, max) valid distance
    const double* lm_normal;
    const uint8_t* lm_desc;
    const uint8_t* lm_valid;
    int m, num_levels;
    float log_scale_factor, margin;
    float sf[OVS_MAX_LEVELS], ils[OVS_MAX_LEVELS];
    int variant;                  // kFuseReplace / kFuseDetect / kFuseMutual
    uint32_t max_dist;            // acceptance threshold on the best Hamming distance
    double P1[12];                // kFuseMutual: pose of the landmark's own keyframe (pos_1 = R_1w X + t_1w, then cam.P = [s R_21 | t_21])
};

// kFuseReplace: fuse::replace_duplication (chi-square gate); kFuseDetect: fuse::detect_duplication (no chi-square gate);
// kFuseMutual: one direction of projection::match_keyframes_mutually (no viewing-angle gate, distance = |pos in the other keyframe|)
enum { kFuseReplace = 0, kFuseDetect = 1, kFuseMutual = 2 };

// Round 6: kFuseLanes lanes per landmark. One lane per landmark walked its (up to nine) grid cells one after the other -- cell bounds, item,
// keypoint record, descriptor: four dependent loads per candidate, 2000 landmarks on eight workgroups, 35 us of latency. Now lane `sub` of a
// landmark takes the cells whose ordinal in upstream's enumeration (cx outer, cy inner) is sub, sub + 16, ...; the winner is the minimum of
// {distance : 16 | cell ordinal : 24 | position in the cell : 24} over the sixteen lanes, i.e. the smallest distance and among equal distances
// the first candidate upstream's loop meets (its `d < best` is strict): the same index. The per-landmark prelude is computed by all sixteen.
constexpr int kFuseLanes = 16;
__global__ __launch_bounds__(256) void k_fuse_best(FuseArgs a, int32_t* __restrict__ best_out, int32_t* __restrict__ num_fused) {
    const int gl = blockIdx.x * 256 + threadIdx.x;
    const int l = gl / kFuseLanes, sub = gl % kFuseLanes;
    unsigned long long key = ~0ull;
    int best_idx = -1;
    do {
        if (l >= a.m) break;
        if (a.lm_valid && !a.lm_valid[l]) break;
        const double* Xw = a.lm_pos_w + 3 * (size_t)l;
        // (held by value: a pointer that is either the global position or a local array put that array, 32 bytes, in scratch memory)
        double X[3] = {Xw[0], Xw[1], Xw[2]};
        if (a.variant == kFuseMutual) {
            const double x0 = (a.P1[0] * X[0] + a.P1[1] * X[1]) + a.P1[2] * X[2] + a.P1[9];
            const double x1 = (a.P1[3] * X[0] + a.P1[4] * X[1]) + a.P1[5] * X[2] + a.P1[10];
            const double x2 = (a.P1[6] * X[0] + a.P1[7] * X[1]) + a.P1[8] * X[2] + a.P1[11];
            X[0] = x0;
            X[1] = x1;
            X[2] = x2;
        }
        double u, v;
        float x_right;
        if (!reproject_to_image(a.cam, X, u, v, x_right)) break;
        double dx, dy, dz;
        if (a.variant == kFuseMutual) {   // pos_2 = s R_21 pos_1 + t_21 (the same expression reproject_to_image evaluates)
            dx = (a.cam.P[0] * X[0] + a.cam.P[1] * X[1]) + a.cam.P[2] * X[2] + a.cam.P[9];
            dy = (a.cam.P[3] * X[0] + a.cam.P[4] * X[1]) + a.cam.P[5] * X[2] + a.cam.P[10];
            dz = (a.cam.P[6] * X[0] + a.cam.P[7] * X[1]) + a.cam.P[8] * X[2] + a.cam.P[11];
        } else {
            dx = X[0] - a.cc[0];
            dy = X[1] - a.cc[1];
            dz = X[2] - a.cc[2];
        }
        const double dist = sqrt((dx * dx + dy * dy) + dz * dz);
        const float dmin = a.lm_dist[2 * l], dmax = a.lm_dist[2 * l + 1];   // raw min_valid_dist_ / max_valid_dist_
        if (dist < (double)(float)(0.7 * (double)dmin) || (double)(float)(1.3 * (double)dmax) < dist) break;
        if (a.variant != kFuseMutual) {
            const double* nrm = a.lm_normal + 3 * (size_t)l;
            if ((dx * nrm[0] + dy * nrm[1]) + dz * nrm[2] < 0.5 * dist) break;
        }
        const float ratio = __fdiv_rn(dmax, (float)dist);
        int pred = (int)ceilf(__fdiv_rn(ovs_det_logf(ratio), a.log_scale_factor));
        if (pred < 0) pred = 0;
        else if (a.num_levels <= pred) pred = a.num_levels - 1;
        const float r = __fmul_rn(a.margin, a.sf[pred]);
        const float ref_x = (float)u, ref_y = (float)v;
        const GridP& g = a.gp;
        const int min_cx = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(ref_x, g.min_x), r), g.inv_w)));
        const int max_cx = min(g.cols - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(ref_x, g.min_x), r), g.inv_w)));
        const int min_cy = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(ref_y, g.min_y), r), g.inv_h)));
        const int max_cy = min(g.rows - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(ref_y, g.min_y), r), g.inv_h)));
        if (!(min_cx < g.cols && max_cx >= 0 && min_cy < g.rows && max_cy >= 0)) break;
        uint32_t qd[8];
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.lm_desc + (size_t)l * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) qd[i] = src[i];
        const int ny = max_cy - min_cy + 1, n_cells = (max_cx - min_cx + 1) * ny;
        for (int ord = sub; ord < n_cells; ord += kFuseLanes) {
            {
                const int ox = ord / ny;
                const int cx = min_cx + ox, cy = min_cy + (ord - ox * ny);
                const int c = cx * g.rows + cy;
                const int k0 = a.cell_start[c], e = a.cell_start[c + 1];
                for (int k = k0; k < e; ++k) {
                    const int idx = a.items[k];
                    const ovs_keypoint kp = a.t_kps[idx];
                    if (!(fabsf(__fsub_rn(kp.x, ref_x)) < r && fabsf(__fsub_rn(kp.y, ref_y)) < r)) continue;
                    const int level = kp.octave;
                    if (level < pred - 1 || pred < level) continue;
                    if (a.variant == kFuseReplace) {
                        const double ex = u - (double)kp.x, ey = v - (double)kp.y;
                        const float xr = a.t_x_right ? a.t_x_right[idx] : -1.0f;
                        if (xr >= 0) {
                            const double exr = (double)x_right - (double)xr;
                            const double e2 = (ex * ex + ey * ey) + exr * exr;
                            if ((double)7.81473f < e2 * (double)a.ils[level]) continue;
                        } else {
                            const double e2 = ex * ex + ey * ey;
                            if ((double)5.99146f < e2 * (double)a.ils[level]) continue;
                        }
                    }
                    const uint32_t d = hamming256_g(qd, reinterpret_cast<const uint32_t*>(a.t_desc + (size_t)idx * 32));
                    const unsigned long long cand = ((unsigned long long)d << 48) | ((unsigned long long)(ord & 0xffffff) << 24) | (unsigned long long)((uint32_t)(k - k0) & 0xffffffu);
                    if (cand < key) {   // (within a lane the keys grow with the enumeration: this is upstream's strict `d < best`)
                        key = cand;
                        best_idx = idx;
                    }
                }
            }
        }
    } while (false);
#pragma unroll
    for (int off = kFuseLanes / 2; off > 0; off >>= 1) {   // (xor partners stay inside the landmark's aligned group of sixteen lanes)
        const unsigned long long ok = __shfl_xor(key, off);
        const int oi = __shfl_xor(best_idx, off);
        if (ok < key) {
            key = ok;
            best_idx = oi;
        }
    }
    const int32_t result = (key != ~0ull && (uint32_t)(key >> 48) <= a.max_dist) ? best_idx : -1;
    if (sub == 0 && l < a.m) best_out[l] = result;
    const unsigned long long any = __ballot(sub == 0 && l < a.m && result >= 0);
    if ((threadIdx.x & 63) == 0 && any) atomicAdd(num_fused, (int32_t)__popcll(any));
}

// projection::match_keyframes_mutually, last step: keep idx_1 -> idx_2 only if idx_2 -> idx_1 came back
__global__ __launch_bounds__(256) void k_cross_check(const int32_t* __restrict__ m_2_in_1, const int32_t* __restrict__ m_1_in_2, int n1, int n2,
                                                     int32_t* __restrict__ out, int32_t* __restrict__ num) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    int32_t r = -1;
    if (i < n1) {
        const int32_t j = m_2_in_1[i];
        if (j >= 0 && j < n2 && m_1_in_2[j] == i) r = j;
        out[i] = r;
    }
    const unsigned long long any = __ballot(r >= 0);
    if ((threadIdx.x & 63) == 0 && any) atomicAdd(num, (int32_t)__popcll(any));
}

// bow_tree: queries = the keyframe's feature-vector entries in walk order; the list of a query is its node's frame bucket
struct BowArgs {
    const uint8_t* kf_desc;
    const uint8_t* kf_valid;
    const int32_t* kf_node_ids;
    const int32_t* kf_node_start;
    const int32_t* kf_items;
    int kf_nodes, n_q;           // n_q = kf_node_start[kf_nodes]
    const uint8_t* frm_desc;
    const uint8_t* frm_valid;    // match_keyframes: the target keypoint must hold a live landmark too (NULL = all)
    // robust::match_for_triangulation (tri != 0): pair filters d <= THR_LOW, epipole proximity, epipolar constraint; candidates are
    // listed in REVERSE bucket order because upstream lets a later equal distance replace an earlier one
    int tri;
    uint32_t dead_from;   // as WinArgs::dead_from (bow rule; the triangulation filter d <= THR_LOW is part of its pair test)
    const ovs_keypoint* kf_kps;  // octave of keypoint 1 (threshold scale)
    const float* kf_x_right;
    const float* frm_x_right;
    const double* kf_bearings;
    const double* frm_bearings;
    double E[9], epipole[3];
    float sf[OVS_MAX_LEVELS];
    const int32_t* frm_node_ids;
    const int32_t* frm_node_start;
    const int32_t* frm_items;
    int frm_nodes;
};

template <bool FILL>
__global__ __launch_bounds__(256) void k_bow_lists(BowArgs a, uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                                                  uint32_t* __restrict__ keys, uint32_t key_cap, uint32_t* __restrict__ overflow) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.n_q) return;
    const int kf_idx = a.kf_items[q];
    uint32_t n = 0;
    if (!a.kf_valid || a.kf_valid[kf_idx]) {
        // node of this entry: last node whose start <= q
        int lo = 0, hi = a.kf_nodes - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.kf_node_start[mid] <= q) lo = mid;
            else hi = mid - 1;
        }
        const int node = a.kf_node_ids[lo];
        int l2 = 0, h2 = a.frm_nodes;   // lower_bound
        while (l2 < h2) {
            const int mid = (l2 + h2) >> 1;
            if (a.frm_node_ids[mid] < node) l2 = mid + 1;
            else h2 = mid;
        }
        if (l2 < a.frm_nodes && a.frm_node_ids[l2] == node) {
            const int b = a.frm_node_start[l2], e = a.frm_node_start[l2 + 1];
            if (a.tri) {
                uint32_t qd[8];
                const uint32_t* src = reinterpret_cast<const uint32_t*>(a.kf_desc + (size_t)kf_idx * 32);
#pragma unroll
                for (int i = 0; i < 8; ++i) qd[i] = src[i];
                const bool stereo_1 = a.kf_x_right && 0 <= a.kf_x_right[kf_idx];
                const double* b1 = a.kf_bearings + 3 * (size_t)kf_idx;
                const double thr = (0.2 * 3.14159265358979323846 / 180.0) * (double)a.sf[a.kf_kps[kf_idx].octave];
                uint32_t pos = FILL ? offsets[q] : 0u;
                for (int k = e - 1; k >= b; --k) {
                    const int idx = a.frm_items[k];
                    if (a.frm_valid && !a.frm_valid[idx]) continue;
                    const uint32_t d = hamming256_g(qd, reinterpret_cast<const uint32_t*>(a.frm_desc + (size_t)idx * 32));
                    if ((uint32_t)OVS_HAMMING_DIST_THR_LOW < d) continue;
                    const double* b2 = a.frm_bearings + 3 * (size_t)idx;
                    if (!stereo_1 && !(a.frm_x_right && 0 <= a.frm_x_right[idx])) {
                        const double cos_dist = (a.epipole[0] * b2[0] + a.epipole[1] * b2[1]) + a.epipole[2] * b2[2];
                        if (0.99862953475 < cos_dist) continue;
                    }
                    const double ex = (a.E[0] * b2[0] + a.E[1] * b2[1]) + a.E[2] * b2[2], ey = (a.E[3] * b2[0] + a.E[4] * b2[1]) + a.E[5] * b2[2],
                                 ez = (a.E[6] * b2[0] + a.E[7] * b2[1]) + a.E[8] * b2[2];
                    const double nrm = sqrt((ex * ex + ey * ey) + ez * ez);
                    const double cos_residual = ((ex * b1[0] + ey * b1[1]) + ez * b1[2]) / nrm;
                    const double residual_rad = 3.14159265358979323846 / 2.0 - fabs(ovs_det_acos(cos_residual));
                    if (!(residual_rad < thr)) continue;
                    if (FILL) {
                        if (pos < key_cap) keys[pos] = (d << 20) | (uint32_t)idx;
                        else *overflow = 1u;
                        ++pos;
                    }
                    ++n;
                }
            } else {
            uint32_t qd[8];
            if (FILL || a.dead_from) {
                const uint32_t* src = reinterpret_cast<const uint32_t*>(a.kf_desc + (size_t)kf_idx * 32);
#pragma unroll
                for (int i = 0; i < 8; ++i) qd[i] = src[i];
            }
            if (!a.frm_valid && !a.dead_from) n = (uint32_t)(e - b);
            else
                for (int k = b; k < e; ++k) {
                    const int idx = a.frm_items[k];
                    bool ok = !a.frm_valid || a.frm_valid[idx];
                    if (ok && a.dead_from) ok = hamming256_g(qd, reinterpret_cast<const uint32_t*>(a.frm_desc + (size_t)idx * 32)) < a.dead_from;
                    n += ok ? 1u : 0u;
                }
            if (FILL) {
                uint32_t pos = offsets[q];
                for (int k = b; k < e; ++k) {
                    const int idx = a.frm_items[k];
                    if (a.frm_valid && !a.frm_valid[idx]) continue;
                    const uint32_t d = hamming256_g(qd, reinterpret_cast<const uint32_t*>(a.frm_desc + (size_t)idx * 32));
                    if (a.dead_from && !(d < a.dead_from)) continue;
                    if (pos < key_cap) keys[pos] = (d << 20) | (uint32_t)idx;
                    else *overflow = 1u;
                    ++pos;
                }
            }
            }
        }
    }
    if (!FILL) counts[q] = n;
}

// exclusive scan of counts[n] into offsets[n + 1] (one workgroup; n <= 65535)
__global__ __launch_bounds__(1024) void k_scan_counts(const uint32_t* __restrict__ counts, int n, uint32_t* __restrict__ offsets) {
    __shared__ uint32_t s_part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    uint32_t local = 0;
    for (int k = 0; k < per; ++k) {
        const int i = tid * per + k;
        if (i < n) local += counts[i];
    }
    s_part[tid] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - local;
    for (int k = 0; k < per; ++k) {
        const int i = tid * per + k;
        if (i < n) {
            offsets[i] = run;
            run += counts[i];
        }
    }
    if (tid == 1023) offsets[n] = s_part[1023];
}

// ---- 3. sequential-claim resolver ---------------------------------------------------------------------------------------
enum { kRuleProjection = 0, kRuleArea = 1, kRuleBow = 2, kRuleBestOnly = 3, kRuleTriang = 4 };   // Triang = BestOnly accept, Bow output

struct ResolveArgs {
    const uint32_t* offsets;   // n_q + 1
    const uint32_t* keys;
    int n_q, n_t;
    float lowe_ratio;
    int check_orientation;
    int angle_keep_rule;   // ovs_match_set_variant(OVS_MATCH_VARIANT_ANGLE_KEEP_RULE): filled in by launch_resolve
    int angle_tie_order;   // ovs_match_set_variant(OVS_MATCH_VARIANT_ANGLE_TIE_ORDER): 0 equal sizes -> lower bin first, 1 higher bin first
    const ovs_keypoint* q_kps;    // area: frame-1 keypoints (angle); bow: keyframe keypoints (angle), indexed through q_items
    const int32_t* q_items;       // bow: query -> keyframe keypoint index (NULL: identity)
    const ovs_keypoint* t_kps;    // area / bow: target keypoints (angle, pt)
    float* prev_matched_xy;       // area: updated for the final matches
    int32_t* assigned;            // projection / area: [n_q] target or -1; bow: [n_t] keyframe keypoint index or -1
    uint32_t key_pool;            // LDS words available for a copy of the key CSR
    int offsets_in_lds;           // the list bounds are staged in LDS (room for n_q + 1 words before the key pool)
    uint32_t best_only_thr;       // BestOnly: accept iff best <= this (match_current_and_last_frames: THR_HIGH)
    int bow_by_query;             // bow (match_keyframes): output [n_out_q] indexed by q_items[q] = target or -1
    int n_out_q;
    int32_t* num_matches;
};

template <int RULE>
__device__ __forceinline__ bool rule_accepts(uint32_t best, uint32_t second, float ratio, uint32_t best_only_thr) {
    if (best == kNone) return false;
    const uint32_t bd = best >> 20;
    const uint32_t sd = second == kNone ? (uint32_t)OVS_MAX_HAMMING_DIST : (second >> 20);
    if (RULE == kRuleBestOnly || RULE == kRuleTriang) return bd <= best_only_thr;   // match_current_and_last_frames / match_frame_and_keyframe: no ratio test
    if (RULE == kRuleProjection) {
        if (bd > (uint32_t)OVS_HAMMING_DIST_THR_HIGH) return false;
        const int bl = (int)((best >> 16) & 15u), sl = second == kNone ? -1 : (int)((second >> 16) & 15u);
        return !(bl == sl && (float)bd > __fmul_rn(ratio, (float)sd));
    }
    if (bd > (uint32_t)OVS_HAMMING_DIST_THR_LOW) return false;
    return !(__fmul_rn((float)sd, ratio) < (float)bd);   // area: second * ratio < best; bow: ratio * second < best (same product)
}

// NW waves replay 64 * NW queries per round. A pending query evaluates (best, second) against the current claims and stamps EVERY live
// candidate of its list with its thread index (LDS atomicMin, epoch-tagged). It is AFFECTED if a lower thread of the round stamped its best
// (or, where the rule has a ratio test, its second): some earlier, still undecided query has that target in its list, so it could still
// claim it -- or could still come to look at it, which is why the stamps cover all live candidates and not just the claimable ones: an
// early commit must not change what an earlier query sees when it evaluates again. An unaffected query is final whatever the others do
// (candidate sets only shrink), so EVERY unaffected query of the round commits -- round 2 committed only the ones below the first affected
// thread, and one cluster of neighbours sharing a target held the whole batch back (~6 rounds per 256 tracked-frame queries). The lowest
// pending query is never affected, so each round makes progress; the number of rounds is the longest chain of queries sharing candidates.
// tests/test_resolver_rule.py is the executable statement of this rule against the sequential loops (and of why the stamps must cover
// every live candidate). With NW > 1 the waves meet at two LDS-only workgroup barriers per round.
template <int RULE, int NW>
__global__ __launch_bounds__(64 * NW) void k_list_resolve(ResolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_res[];
    typedef __attribute__((address_space(3))) volatile uint32_t lds_u32;
    typedef __attribute__((address_space(3))) volatile uint16_t lds_u16;
    constexpr int T = 64 * NW;
    __shared__ uint32_t s_first[NW], s_any[NW], s_keep[4], s_keep_cnt[4];
    lds_u32* mark = (lds_u32*)s_res;                                        // [kMarkSize]
    lds_u32* hist = mark + kMarkSize;                                       // [32]
    lds_u16* thr = (lds_u16*)(hist + 32);                                   // [n_t]  alive(d, t) <=> d < thr[t]
    lds_u16* owner = thr + ((a.n_t + 1) & ~1);                              // [n_t]  area: query currently matched to t
    lds_u16* match = owner + ((a.n_t + 1) & ~1);                            // [n_q]  target of query (0xFFFF none)
    lds_u16* accepted = match + ((a.n_q + 1) & ~1);                         // [n_q]  target at acceptance time (orientation entries)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto wg_barrier = [&]() {
        if (NW == 1) {
            __builtin_amdgcn_wave_barrier();
        } else {   // orders LDS traffic only: the key loads of the global path need not drain
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
    };
    const uint32_t max_d = (RULE == kRuleBestOnly || RULE == kRuleTriang) ? a.best_only_thr : (RULE == kRuleProjection ? OVS_HAMMING_DIST_THR_HIGH : OVS_HAMMING_DIST_THR_LOW);
    for (int i = tid; i < a.n_t; i += T) {
        thr[i] = (uint16_t)OVS_MAX_HAMMING_DIST;
        owner[i] = 0xFFFFu;
    }
    for (int i = tid; i < a.n_q; i += T) {
        match[i] = 0xFFFFu;
        accepted[i] = 0xFFFFu;
    }
    for (int i = tid; i < kMarkSize; i += T) mark[i] = ~0u;
    if (tid < 32) hist[tid] = 0;
    // the candidate CSR is one contiguous array: when it fits the rest of the LDS allocation it is copied there once (coalesced
    // 16-byte loads), and the rounds below never touch HBM
    // the queries' list bounds, staged once (coalesced): every batch of the loop below used to start with two dependent global loads
    lds_u32* s_off = (lds_u32*)(accepted + ((a.n_q + 1) & ~1));             // [n_q + 1] (if offsets_in_lds)
    lds_u32* s_keys = s_off + (a.offsets_in_lds ? ((a.n_q + 4) & ~3) : 0);
    if (a.offsets_in_lds)
        for (int i = tid; i <= a.n_q; i += T) s_off[i] = a.offsets[i];
    const uint32_t n_keys = a.offsets[a.n_q];
    const bool keys_in_lds = n_keys <= a.key_pool;
    if (keys_in_lds) {
        const uint32_t n4 = n_keys >> 2;
        const uint4* src4 = reinterpret_cast<const uint4*>(a.keys);
        for (uint32_t i = tid; i < n4; i += T) {
            const uint4 v = src4[i];
            s_keys[4 * i] = v.x;
            s_keys[4 * i + 1] = v.y;
            s_keys[4 * i + 2] = v.z;
            s_keys[4 * i + 3] = v.w;
        }
        for (uint32_t i = 4 * n4 + tid; i < n_keys; i += T) s_keys[i] = a.keys[i];
    }
    wg_barrier();
    uint32_t epoch = 0;

    for (int q0 = 0; q0 < a.n_q; q0 += T) {
        const int q = q0 + tid;
        uint32_t lb = 0, le = 0;
        if (q < a.n_q) {
            lb = a.offsets_in_lds ? (uint32_t)s_off[q] : a.offsets[q];
            le = a.offsets_in_lds ? (uint32_t)s_off[q + 1] : a.offsets[q + 1];
        }
        bool pending = le > lb;
        for (;;) {
            // ---- any query of the batch still undecided? (the barrier also publishes the previous round's commits)
            {
                const unsigned long long pb = __ballot(pending);
                if (NW == 1) {
                    if (!pb) break;
                } else {
                    if (lane == 0) s_any[wv] = pb != 0ull;
                    wg_barrier();
                    uint32_t any = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) any |= s_any[w];
                    if (!any) break;
                }
            }
            uint32_t best = kNone, second = kNone;
            bool acc = false;
            ++epoch;
            const uint32_t tag = (0x3FFFFFu - epoch) << 10;   // newer rounds carry smaller tags: atomicMin overrides stale stamps; low 10 bits: thread
            asm volatile("" ::: "memory");   // thr[] committed in the previous round must be re-read
            if (pending) {
                uint32_t bd = OVS_MAX_HAMMING_DIST, sd = OVS_MAX_HAMMING_DIST;
                // eight keys and their eight thr[] values are fetched as independent batches (one load round trip per batch instead of
                // one per entry: that latency is the whole cost); keys come from LDS when the CSR was staged
                const __attribute__((address_space(3))) uint16_t* thr_nv = (const __attribute__((address_space(3))) uint16_t*)thr;
                // (the staged keys never change: read them through a plain pointer too, or each of the eight reads waits for the one before)
                const __attribute__((address_space(3))) uint32_t* keys_nv = (const __attribute__((address_space(3))) uint32_t*)s_keys;
                // UNCONDITIONAL loads (indices clamped into the list, results masked afterwards): a load under a per-lane condition gets its own
                // branch and its own wait, i.e. eight serialised round trips for the keys and eight for thr[] (seen in the ISA)
                const uint32_t last = le - 1u;   // (pending => le > lb)
                for (uint32_t k0 = lb; k0 < le; k0 += 8) {
                    uint32_t e8[8], t8[8];
                    if (keys_in_lds) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) e8[u] = keys_nv[min(k0 + u, last)];
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u) e8[u] = a.keys[min(k0 + u, last)];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) e8[u] = k0 + u < le ? e8[u] : kNone;
#pragma unroll
                    for (int u = 0; u < 8; ++u) t8[u] = (uint32_t)thr_nv[e8[u] != kNone ? (e8[u] & 0xFFFFu) : 0u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t8[u] = e8[u] != kNone ? t8[u] : 0u;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const uint32_t e = e8[u], d = e >> 20;
                        if (e == kNone || !(d < t8[u])) continue;   // claimed / matched at a distance <= ours
                        // every live candidate, whatever its distance: a later query must neither rest on a target this one may still
                        // claim nor claim a target this one may still come to look at
                        __hip_atomic_fetch_min((__attribute__((address_space(3))) uint32_t*)&mark[(e & 0xFFFFu) & (kMarkSize - 1)],
                                               tag | (uint32_t)tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (d < bd) {
                            second = best;
                            sd = bd;
                            best = e;
                            bd = d;
                        } else if (d < sd) {
                            second = e;
                            sd = d;
                        }
                    }
                }
                acc = rule_accepts<RULE>(best, second, a.lowe_ratio, a.best_only_thr);
            }
            wg_barrier();
            bool affected = false;
            if (pending && best != kNone && (best >> 20) <= max_d) {   // a best beyond the threshold is a final reject (it can only grow)
                const uint32_t m1 = mark[(best & 0xFFFFu) & (kMarkSize - 1)];
                affected = (m1 & ~0x3FFu) == tag && (m1 & 0x3FFu) < (uint32_t)tid;
                if (RULE != kRuleBestOnly && RULE != kRuleTriang && second != kNone) {   // (no ratio test: the decision rests on `best` alone)
                    const uint32_t m2 = mark[(second & 0xFFFFu) & (kMarkSize - 1)];
                    affected |= (m2 & ~0x3FFu) == tag && (m2 & 0x3FFu) < (uint32_t)tid;
                }
            }
            if (pending && !affected && acc) {
                const uint32_t t = best & 0xFFFFu;
                if (RULE == kRuleArea) {
                    const uint32_t prev = owner[t];
                    if (prev != 0xFFFFu) match[prev] = 0xFFFFu;   // the earlier owner loses the target
                    owner[t] = (uint16_t)q;
                    thr[t] = (uint16_t)(best >> 20);
                } else {
                    thr[t] = 0;
                }
                match[q] = (uint16_t)t;
                accepted[q] = (uint16_t)t;
            }
            if (!affected) pending = false;
            if (NW == 1) __builtin_amdgcn_wave_barrier();   // NW > 1: the barrier at the top of the loop publishes the commits
        }
    }
    wg_barrier();
    // ---- match::angle_checker: bin = cvRound(delta / 30) over every ACCEPTED query (a later-stolen match keeps its entry)
    if (RULE != kRuleProjection && a.check_orientation) {
        for (int q = tid; q < a.n_q; q += T) {
            const uint32_t t = accepted[q];
            if (t == 0xFFFFu) continue;
            const int qi = a.q_items ? a.q_items[q] : q;
            float delta = __fsub_rn(a.q_kps[qi].angle, a.t_kps[t].angle);
            if (delta < 0.0f) delta = __fadd_rn(delta, 360.0f);
            if (360.0f <= delta) delta = __fsub_rn(delta, 360.0f);
            int bin = __float2int_rn(__fmul_rn(delta, 1.0f / 30));
            if (bin == 30) bin = 0;
            __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t*)&hist[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            accepted[q] = (uint16_t)(0x8000u | (uint32_t)bin);   // reuse the slot: bin of this entry
        }
        wg_barrier();
        // the three fullest bins, equal sizes -> lower bin first (ORACLE_SPEC rule 17); wave 0 picks them
        if (wv == 0) {
            const uint32_t h = lane < 30 ? (uint32_t)hist[lane] : 0u;
            // larger count first; equal counts: the lower bin (rule 17 as fixed) or, in the variant, the higher bin -- upstream std::sort's on the
            // sizes, which leaves the order of equal bins to the library
            const uint32_t tie_key = a.angle_tie_order ? (uint32_t)(lane + 1) : (uint32_t)(31 - lane);
            uint32_t key = lane < 30 ? ((h << 8) | tie_key) : 0u;
            for (int k = 0; k < 3; ++k) {
                uint32_t m = key;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const uint32_t o = __shfl_xor(m, off);
                    m = o > m ? o : m;
                }
                if (lane == 0) {
                    s_keep[k] = a.angle_tie_order ? (m & 0xFFu) - 1u : (uint32_t)(31 - (int)(m & 0xFFu));
                    s_keep_cnt[k] = m >> 8;
                }
                if (key == m) key = 0u;
            }
            // rule 17's alternative (ORB-SLAM2's ComputeThreeMaxima): a second bin below 0.1 x the fullest drops out and takes the third with it,
            // a third bin below that alone (bin 99 matches nothing)
            if (a.angle_keep_rule && lane == 0) {
                const float max1 = (float)s_keep_cnt[0];
                if ((float)s_keep_cnt[1] < __fmul_rn(0.1f, max1)) s_keep[1] = s_keep[2] = 99u;
                else if ((float)s_keep_cnt[2] < __fmul_rn(0.1f, max1)) s_keep[2] = 99u;
            }
        }
        wg_barrier();
        const int keep0 = (int)s_keep[0], keep1 = (int)s_keep[1], keep2 = (int)s_keep[2];
        for (int q = tid; q < a.n_q; q += T) {
            const uint32_t v = accepted[q];
            if (!(v & 0x8000u) || v == 0xFFFFu) continue;
            const int bin = (int)(v & 0xFFu);
            if (bin != keep0 && bin != keep1 && bin != keep2) match[q] = 0xFFFFu;
        }
        wg_barrier();
    }
    // ---- outputs
    uint32_t total = 0;
    constexpr bool kBowOut = RULE == kRuleBow || RULE == kRuleTriang;
    if (kBowOut) {
        const int n_clear = a.bow_by_query ? a.n_out_q : a.n_t;
        for (int t = tid; t < n_clear; t += T) a.assigned[t] = -1;
        __threadfence_block();
        __syncthreads();   // global writes above must land before the scattered ones below
    }
    for (int q0 = 0; q0 < a.n_q; q0 += T) {
        const int q = q0 + tid;
        uint32_t t = 0xFFFFu;
        if (q < a.n_q) t = match[q];
        if (q < a.n_q) {
            if (kBowOut) {
                if (t != 0xFFFFu) {
                    const int qi = a.q_items ? a.q_items[q] : q;
                    if (a.bow_by_query) a.assigned[qi] = (int32_t)t;
                    else a.assigned[t] = qi;
                }
            } else {
                a.assigned[q] = t == 0xFFFFu ? -1 : (int32_t)t;
                if (RULE == kRuleArea && t != 0xFFFFu) {
                    a.prev_matched_xy[2 * q] = a.t_kps[t].x;
                    a.prev_matched_xy[2 * q + 1] = a.t_kps[t].y;
                }
            }
        }
        total += (uint32_t)__popcll(__ballot(t != 0xFFFFu));
    }
    if (NW == 1) {
        if (lane == 0) *a.num_matches = (int32_t)total;
    } else {
        if (lane == 0) s_first[wv] = total;
        wg_barrier();
        if (tid == 0) {
            uint32_t sum = 0;
            for (int w = 0; w < NW; ++w) sum += s_first[w];
            *a.num_matches = (int32_t)sum;
        }
    }
}

}   // namespace ovs

using namespace ovs;

// ovs_match_set_variant(OVS_MATCH_VARIANT_ANGLE_KEEP_RULE, 0 | 1): process-wide, read at every resolver launch
static std::atomic<int> g_angle_keep_rule{0};
static std::atomic<int> g_angle_tie_order{0};   // OVS_MATCH_VARIANT_ANGLE_TIE_ORDER
static std::atomic<int> g_bf_frame_mask{0};     // OVS_MATCH_VARIANT_BF_FRAME_MASK (read by the class shim of robust::brute_force_match)

struct ovs_wmatcher {
    int device = 0;
    int max_t = 0, max_q = 0;
    uint32_t max_entries = 0;
    hipStream_t stream = nullptr;
    // grid of the target frame
    int32_t* d_cell_of = nullptr;
    int32_t* d_cell_start = nullptr;
    int32_t* d_items = nullptr;
    int grid_cells_cap = 0;
    GridP gp{};
    int grid_n = -1;
    // candidate lists
    uint32_t* d_counts = nullptr;
    uint32_t* d_offsets = nullptr;
    uint32_t* d_keys = nullptr;
    uint32_t* d_overflow = nullptr;
    bool overflow_dirty = true;        // the overflow word may be non-zero (build_lists clears it then)
    int32_t* d_assigned = nullptr;      // max(max_q, max_t)
    int32_t* d_num = nullptr;
    // host-API staging
    ovs_keypoint* d_t_kps = nullptr;
    uint8_t* d_t_desc = nullptr;
    uint8_t* d_t_flag = nullptr;
    float* d_t_f = nullptr;
    ovs_keypoint* d_q_kps = nullptr;
    uint8_t* d_q_desc = nullptr;
    uint8_t* d_q_flag = nullptr;
    float* d_q_xy = nullptr;
    float* d_q_f = nullptr;
    int32_t* d_q_i = nullptr;
    float* d_q_r = nullptr;
    int32_t* d_q_i2 = nullptr;
    double* d_q_pos = nullptr;
    float* d_sf = nullptr;
    double* d_tri_b1 = nullptr;
    double* d_tri_b2 = nullptr;
    int32_t* d_csr = nullptr;           // bow feature vectors: 2 x (ids | start | items)
    size_t csr_cap = 0;
    // round 3: d_overflow | d_num[4] | pad | d_assigned are ONE block (a call's results come down with one copy), and the frame-handle
    // entry points stage their per-call host arrays through one pinned buffer into one device arena (one copy up)
    uint32_t* d_res_block = nullptr;
    int32_t* h_res = nullptr;           // pinned mirror of the result block
    unsigned char* h_stage = nullptr;   // pinned
    unsigned char* d_stage = nullptr;
    size_t stage_cap = 0;
};

// a frame / keyframe resident in HBM (created by ovs_frame_dev_create below)
struct ovs_frame_dev {
    int device = 0;
    int n = 0, cap = 0, n_cells = 0;
    bool has_stereo = false;
    GridP gp{};
    ovs_grid_params gpp{};
    unsigned char* arena = nullptr;
    size_t arena_bytes = 0;
    ovs_keypoint* d_kps = nullptr;
    uint8_t* d_desc = nullptr;
    float* d_x_right = nullptr;
    int32_t *d_cell_of = nullptr, *d_cell_start = nullptr, *d_items = nullptr;
    double* d_bearings = nullptr;   // optional (ovs_frame_dev_attach_bearings): 3 doubles per keypoint, for match_for_triangulation
};

namespace {

constexpr int kMaxGridCells = 16384;

GridP make_gridp(const ovs_grid_params& p) {
    GridP g;
    g.min_x = p.min_x;
    g.min_y = p.min_y;
    g.inv_w = (float)((double)p.cols / (double)(p.max_x - p.min_x));   // camera::base: double division, float member
    g.inv_h = (float)((double)p.rows / (double)(p.max_y - p.min_y));
    g.cols = p.cols;
    g.rows = p.rows;
    return g;
}

size_t resolve_lds_bytes(int n_q, int n_t) {
    return (size_t)kMarkSize * 4 + 32 * 4 + (size_t)2 * 2 * ((n_t + 1) & ~1) + (size_t)2 * 2 * ((n_q + 1) & ~1);
}

template <int RULE>
ovs_status launch_resolve(const ResolveArgs& ra_in, hipStream_t s) {
    ResolveArgs ra = ra_in;
    ra.angle_keep_rule = g_angle_keep_rule.load(std::memory_order_relaxed);
    ra.angle_tie_order = g_angle_tie_order.load(std::memory_order_relaxed);
    size_t fixed = resolve_lds_bytes(ra.n_q, ra.n_t);
    if (fixed > 150 * 1024) return OVS_ERR_CAPACITY;
    // the queries' list bounds are staged in LDS too unless the problem is so large that they would take the room of everything else
    const size_t off_bytes = (size_t)4 * ((ra.n_q + 4) & ~3);
    ra.offsets_in_lds = fixed + off_bytes <= (size_t)120 * 1024 ? 1 : 0;
    if (ra.offsets_in_lds) fixed += off_bytes;
    // the rest of the allocation (96 KiB, or 24 KiB beyond the fixed part for the largest problems) holds a copy of the key CSR when it
    // fits (checked on the device: the size is only known there)
    const size_t lds = std::min((size_t)150 * 1024, std::max((size_t)96 * 1024, fixed + (size_t)24 * 1024));
    ra.key_pool = (uint32_t)((lds - fixed) / 4);
    // 1 / 4 / 8 waves (64 / 256 / 512 queries per round) by problem size: up to 256, up to 1024, beyond
    const int wide_from = tuning().resolve_wide_from;   // 1024 (2048 until the commit rule changed: tracked frame, 2008 queries: 0.158 -> 0.147 ms, area 0.123 -> 0.094)
    const int width = ra.n_q > wide_from ? 2 : (ra.n_q > 256 ? 1 : 0);
    {
        static LdsAttrCache configured[3];   // per (RULE: this instantiation, width), per device
        const void* fn = width == 2 ? reinterpret_cast<const void*>(k_list_resolve<RULE, 8>)
                                    : (width == 1 ? reinterpret_cast<const void*>(k_list_resolve<RULE, 4>) : reinterpret_cast<const void*>(k_list_resolve<RULE, 1>));
        OVS_HIP_TRY(ensure_dynamic_lds(fn, lds, configured[width]));
    }
    if (width == 2)
        hipLaunchKernelGGL((k_list_resolve<RULE, 8>), dim3(1), dim3(512), lds, s, ra);
    else if (width == 1)
        hipLaunchKernelGGL((k_list_resolve<RULE, 4>), dim3(1), dim3(256), lds, s, ra);
    else
        hipLaunchKernelGGL((k_list_resolve<RULE, 1>), dim3(1), dim3(64), lds, s, ra);
    OVS_HIP_TRY(hipGetLastError());
    return OVS_OK;
}

ovs_status grid_assign(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* d_kps, int n, hipStream_t s) {
    if (!gp || gp->cols < 1 || gp->rows < 1 || gp->cols * gp->rows > kMaxGridCells || !(gp->max_x > gp->min_x) || !(gp->max_y > gp->min_y))
        return OVS_ERR_INVALID;
    if (n > w->max_t) return OVS_ERR_CAPACITY;
    w->gp = make_gridp(*gp);
    w->grid_n = n;
    const int nc = gp->cols * gp->rows;
    const int in_lds = (n <= kGridItemsLds && (size_t)(2 * nc + 1 + n) * sizeof(int32_t) <= 144 * 1024) ? 1 : 0;
    const size_t lds = (size_t)(2 * nc + 1 + (in_lds ? n : 0)) * sizeof(int32_t);
    if (lds > 64 * 1024)
        OVS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_assign), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_grid_assign, dim3(1), dim3(1024), lds, s, d_kps, n, w->gp, w->d_cell_of, w->d_cell_start, w->d_items, in_lds);
    OVS_HIP_TRY(hipGetLastError());
    return OVS_OK;
}

// The TARGET side of a windowed call (the keypoints that are searched through the grid): either host arrays -- uploaded into the context's
// buffers and indexed by k_grid_assign, per call -- or a frame / keyframe RESIDENT in HBM (ovs_frame_dev: uploaded and indexed once, when the
// handle was created). Keyframes are the long-lived, immutable objects of the map: mapping_module matches every new keyframe against ~20
// covisible ones (fuse::replace_duplication), loop closing against more; with the handle none of those calls moves keypoints or descriptors.
struct TargetRef {
    const ovs_keypoint* kps;
    const uint8_t* desc;
    const float* x_right;
    const int32_t* cell_start;
    const int32_t* items;
    GridP gp;
};
ovs_status stage_target(ovs_wmatcher* w, const ovs_frame_dev* res, const ovs_grid_params* gp, const ovs_keypoint* kps, const uint8_t* desc,
                        const float* x_right, int n, hipStream_t s, TargetRef* t) {
    if (res) {
        if (res->device != w->device || res->n != n) return OVS_ERR_INVALID;
        *t = TargetRef{res->d_kps, res->d_desc, res->has_stereo ? res->d_x_right : nullptr, res->d_cell_start, res->d_items, res->gp};
        return OVS_OK;
    }
    OVS_HIP_TRY(hipMemcpyAsync(w->d_t_kps, kps, sizeof(ovs_keypoint) * (size_t)n, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_t_desc, desc, (size_t)32 * n, hipMemcpyHostToDevice, s));
    if (x_right) OVS_HIP_TRY(hipMemcpyAsync(w->d_t_f, x_right, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, s));
    const ovs_status st = grid_assign(w, gp, w->d_t_kps, n, s);
    if (st != OVS_OK) return st;
    *t = TargetRef{w->d_t_kps, w->d_t_desc, x_right ? w->d_t_f : nullptr, w->d_cell_start, w->d_items, w->gp};
    return OVS_OK;
}

// WinArgs::dead_from for a rule that accepts iff best <= thr and (ratio rules) !(fl(second * ratio) < best): the smallest distance d > thr with
// fl(d * ratio) >= thr -- from there on a candidate can neither be an accepted best nor make the ratio test fail (the float product is monotone
// in d, and "no second at all" counts as distance 256, which then accepts too). 257: nothing is dead; 0 would switch the filter off.
static uint32_t dead_from_ratio(uint32_t thr, float ratio) {
    if (!(ratio > 0.0f)) return 257u;
    for (uint32_t d = thr + 1u; d <= 256u; ++d)
        if ((float)d * ratio >= (float)thr) return d;
    return 257u;
}
static uint32_t dead_from_thr(uint32_t thr) { return thr + 1u; }   // rules without a ratio test: best <= thr or nothing

template <typename ARGS, typename KCOUNT, typename KFILL>
ovs_status build_lists(ovs_wmatcher* w, const ARGS& args, int n_q, KCOUNT kcount, KFILL kfill, hipStream_t s, int q_per_block = 4) {
    if (n_q > w->max_q) return OVS_ERR_CAPACITY;
    // the overflow word is zero between calls unless a call ended without having read it as zero (a failed call, an overflow): only then is it cleared
    // here (round 6: the clearing was a fill kernel and a launch gap in front of every list build).
    // (Also tried: the counting launch's last workgroup scanning the counts instead of the one-workgroup k_scan_counts launch -- the device-scope
    // release every workgroup needs before it takes its ticket is an L2 write-back on this multi-XCD part: +20 us per call, dropped.)
    if (w->overflow_dirty) OVS_HIP_TRY(hipMemsetAsync(w->d_overflow, 0, sizeof(uint32_t), s));
    w->overflow_dirty = true;
    const dim3 grid((n_q + q_per_block - 1) / q_per_block);
    hipLaunchKernelGGL(kcount, grid, dim3(256), 0, s, args, w->d_counts, (const uint32_t*)w->d_offsets, w->d_keys, w->max_entries, w->d_overflow);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, (const uint32_t*)w->d_counts, n_q, w->d_offsets);
    hipLaunchKernelGGL(kfill, grid, dim3(256), 0, s, args, w->d_counts, (const uint32_t*)w->d_offsets, w->d_keys, w->max_entries, w->d_overflow);
    OVS_HIP_TRY(hipGetLastError());
    return OVS_OK;
}


size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// per-call staging of host arrays: memcpy into the context's pinned buffer, ONE hipMemcpyAsync for all of them
struct Stager {
    ovs_wmatcher* w;
    size_t used = 0;
    bool overflow = false;
    explicit Stager(ovs_wmatcher* w_) : w(w_) {}
    template <typename T>
    const T* put(const T* src, size_t count) {   // nullptr in -> nullptr out
        if (!src) return nullptr;
        const size_t bytes = sizeof(T) * count, off = al256(used);
        if (off + bytes > w->stage_cap) {
            overflow = true;
            return nullptr;
        }
        std::memcpy(w->h_stage + off, src, bytes);
        used = off + bytes;
        return reinterpret_cast<const T*>(w->d_stage + off);
    }
    template <typename T>
    T* reserve(size_t count) {   // device-only scratch in the same arena (filled by a kernel)
        const size_t bytes = sizeof(T) * count, off = al256(used);
        if (off + bytes > w->stage_cap) {
            overflow = true;
            return nullptr;
        }
        used = off + bytes;
        return reinterpret_cast<T*>(w->d_stage + off);
    }
    hipError_t flush(hipStream_t s) {
        flushed_on = s;
        flushed = true;
        return used ? hipMemcpyAsync(w->d_stage, w->h_stage, used, hipMemcpyHostToDevice, s) : hipSuccess;
    }
    // A call returns with its stream idle, on EVERY path: the next call (or the shim's immediate retry) memcpy's into the same pinned buffer, and
    // a frame arena may go back to the pool -- neither may happen while a copy or a kernel of a failed call is still in flight. After a
    // successful call the stream is already idle (fetch_results synchronised it) and this costs a microsecond.
    hipStream_t flushed_on = nullptr;
    bool flushed = false;
    ~Stager() {
        if (flushed) (void)hipStreamSynchronize(flushed_on);
    }
};

// results: overflow flag, counts and `n_out` assignments in one copy
ovs_status fetch_results(ovs_wmatcher* w, int n_out, int32_t* assigned, int32_t* num_matches, hipStream_t s, bool check_overflow = true) {
    OVS_HIP_TRY(hipMemcpyAsync(w->h_res, w->d_res_block, sizeof(int32_t) * (8 + (size_t)n_out), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(assigned, w->h_res + 8, sizeof(int32_t) * (size_t)n_out);
    *num_matches = w->h_res[1];
    if (check_overflow && w->h_res[0] == 0) w->overflow_dirty = false;   // read as zero behind every kernel of the call: still zero for the next one
    return (check_overflow && w->h_res[0]) ? OVS_ERR_CAPACITY : OVS_OK;   // (k_fuse_best writes no overflow flag: the word may be stale)
}


}   // namespace

extern "C" {

ovs_status ovs_wmatcher_create(int32_t max_targets, int32_t max_queries, int32_t max_entries, int32_t device, ovs_wmatcher** out) {
    if (!out || max_targets < 1 || max_queries < 1 || max_entries < 1 || max_targets > 65534 || max_queries > 65534) return OVS_ERR_INVALID;
    *out = nullptr;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    if (resolve_lds_bytes(max_queries, max_targets) > 150 * 1024) return OVS_ERR_CAPACITY;
    ovs_wmatcher* w = new (std::nothrow) ovs_wmatcher();
    if (!w) return OVS_ERR_INVALID;
    w->device = device;
    w->max_t = max_targets;
    w->max_q = max_queries;
    w->max_entries = (uint32_t)max_entries;
#define CREATE_TRY(expr)                       \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) {                \
            ovs::set_last_error(#expr, _e);    \
            ovs_wmatcher_destroy(w);           \
            return OVS_ERR_HIP;                \
        }                                      \
    } while (0)
    CREATE_TRY(hipSetDevice(device));
    CREATE_TRY(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
    const size_t T = (size_t)max_targets, Q = (size_t)max_queries, M = std::max(T, Q);
    CREATE_TRY(hipMalloc(&w->d_cell_of, sizeof(int32_t) * T));
    CREATE_TRY(hipMalloc(&w->d_cell_start, sizeof(int32_t) * (kMaxGridCells + 1)));
    CREATE_TRY(hipMalloc(&w->d_items, sizeof(int32_t) * T));
    CREATE_TRY(hipMalloc(&w->d_counts, sizeof(uint32_t) * (Q + 1)));
    CREATE_TRY(hipMalloc(&w->d_offsets, sizeof(uint32_t) * (Q + 1)));
    CREATE_TRY(hipMalloc(&w->d_keys, sizeof(uint32_t) * (size_t)max_entries));
    // result block: [0] overflow flag, [1..4] counts ([1] result count, [2] per-direction scratch count), [8..] assigned
    CREATE_TRY(hipMalloc(&w->d_res_block, sizeof(int32_t) * (8 + M)));
    w->d_overflow = w->d_res_block;
    w->d_num = reinterpret_cast<int32_t*>(w->d_res_block) + 1;
    w->d_assigned = reinterpret_cast<int32_t*>(w->d_res_block) + 8;
    CREATE_TRY(hipMemset(w->d_res_block, 0, sizeof(int32_t) * 8));
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&w->h_res), sizeof(int32_t) * (8 + M), hipHostMallocDefault));
    // per-call query-side staging: keypoints 28 + descriptors 32 + positions 24 + xy 8 + four 4-byte arrays + flags, per query, and the
    // target-side flags; 256-byte alignment slack per array
    w->stage_cap = Q * (28 + 32 + 24 + 8 + 16 + 1) + T * 2 + 16 * 256 + 4096;
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&w->h_stage), w->stage_cap, hipHostMallocDefault));
    CREATE_TRY(hipMalloc(&w->d_stage, w->stage_cap));
    CREATE_TRY(hipMalloc(&w->d_t_kps, sizeof(ovs_keypoint) * T));
    CREATE_TRY(hipMalloc(&w->d_t_desc, 32 * T));
    CREATE_TRY(hipMalloc(&w->d_t_flag, T));
    CREATE_TRY(hipMalloc(&w->d_t_f, sizeof(float) * T));
    CREATE_TRY(hipMalloc(&w->d_q_kps, sizeof(ovs_keypoint) * Q));
    CREATE_TRY(hipMalloc(&w->d_q_desc, 32 * Q));
    CREATE_TRY(hipMalloc(&w->d_q_flag, Q));
    CREATE_TRY(hipMalloc(&w->d_q_xy, sizeof(float) * 2 * Q));
    CREATE_TRY(hipMalloc(&w->d_q_f, sizeof(float) * Q));
    CREATE_TRY(hipMalloc(&w->d_q_i, sizeof(int32_t) * Q));
    CREATE_TRY(hipMalloc(&w->d_q_r, sizeof(float) * Q));
    CREATE_TRY(hipMalloc(&w->d_q_i2, sizeof(int32_t) * Q));
    CREATE_TRY(hipMalloc(&w->d_q_pos, sizeof(double) * 3 * Q));
    CREATE_TRY(hipMalloc(&w->d_sf, sizeof(float) * OVS_MAX_LEVELS));
    CREATE_TRY(hipMalloc(&w->d_tri_b1, sizeof(double) * 3 * Q));
    CREATE_TRY(hipMalloc(&w->d_tri_b2, sizeof(double) * 3 * T));
    w->csr_cap = 4 * (T + Q) + 16;
    CREATE_TRY(hipMalloc(&w->d_csr, sizeof(int32_t) * w->csr_cap));
#undef CREATE_TRY
    *out = w;
    return OVS_OK;
}

ovs_status ovs_wmatcher_destroy(ovs_wmatcher* w) {
    if (!w) return OVS_OK;
    if (w->stream) hipStreamSynchronize(w->stream);
    if (w->h_res) hipHostFree(w->h_res);
    if (w->h_stage) hipHostFree(w->h_stage);
    void* ptrs[] = {w->d_cell_of, w->d_cell_start, w->d_items, w->d_counts, w->d_offsets, w->d_keys, w->d_res_block, w->d_stage,
                    w->d_t_kps,   w->d_t_desc,     w->d_t_flag, w->d_t_f,    w->d_q_kps,   w->d_q_desc, w->d_q_flag,  w->d_q_xy,    w->d_q_f,
                    w->d_q_i,     w->d_csr,        w->d_q_r,    w->d_q_i2,   w->d_q_pos,   w->d_sf,
                    w->d_tri_b1,  w->d_tri_b2};
    for (void* p : ptrs) hipFree(p);
    if (w->stream) hipStreamDestroy(w->stream);
    delete w;
    return OVS_OK;
}

ovs_status ovs_grid_assign_dev(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* d_kps, int32_t n, void* stream) {
    if (!w || !d_kps || n < 0) return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(w->device));
    return grid_assign(w, gp, d_kps, n, (hipStream_t)stream);
}

ovs_status ovs_assign_keypoints_to_grid(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* kps, int32_t n,
                                        int32_t* cell_start, int32_t* items, int32_t* n_items) {
    if (!w || !gp || (n > 0 && !kps) || n < 0 || !cell_start || !n_items) return OVS_ERR_INVALID;
    if (n > w->max_t) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    if (n) OVS_HIP_TRY(hipMemcpyAsync(w->d_t_kps, kps, sizeof(ovs_keypoint) * n, hipMemcpyHostToDevice, s));
    ovs_status st = grid_assign(w, gp, w->d_t_kps, n, s);
    if (st != OVS_OK) return st;
    const int nc = gp->cols * gp->rows;
    OVS_HIP_TRY(hipMemcpyAsync(cell_start, w->d_cell_start, sizeof(int32_t) * (nc + 1), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    *n_items = cell_start[nc];
    if (items && *n_items > 0) {
        OVS_HIP_TRY(hipMemcpyAsync(items, w->d_items, sizeof(int32_t) * *n_items, hipMemcpyDeviceToHost, s));
        OVS_HIP_TRY(hipStreamSynchronize(s));
    }
    return OVS_OK;
}

ovs_status ovs_projection_match_frame_and_landmarks_dev(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* d_kps,
                                                        const uint8_t* d_desc, const float* d_stereo_x_right, const uint8_t* d_occupied,
                                                        int32_t n, const float* d_lm_xy, const float* d_lm_x_right,
                                                        const int32_t* d_lm_level, const uint8_t* d_lm_desc, const uint8_t* d_lm_valid,
                                                        int32_t m, const float* scale_factors, int32_t num_levels, float margin,
                                                        float lowe_ratio, int32_t* d_assigned, int32_t* d_num_matches, void* stream) {
    if (!w || !d_kps || !d_desc || n < 0 || m < 0 || !d_assigned || !d_num_matches || !scale_factors || num_levels < 1 ||
        num_levels > OVS_MAX_LEVELS || (m > 0 && (!d_lm_xy || !d_lm_level || !d_lm_desc)) || (d_stereo_x_right && !d_lm_x_right))
        return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = (hipStream_t)stream;
    if (m == 0) {
        OVS_HIP_TRY(hipMemsetAsync(d_num_matches, 0, sizeof(int32_t), s));
        return OVS_OK;
    }
    ovs_status st = grid_assign(w, gp, d_kps, n, s);
    if (st != OVS_OK) return st;
    WinArgs a{};
    a.t_kps = d_kps;
    a.t_desc = d_desc;
    a.t_occupied = d_occupied;
    a.t_x_right = d_stereo_x_right;
    a.cell_start = w->d_cell_start;
    a.items = w->d_items;
    a.gp = w->gp;
    a.n_q = m;
    a.q_xy = d_lm_xy;
    a.q_x_right = d_lm_x_right;
    a.q_level = d_lm_level;
    a.q_valid = d_lm_valid;
    a.q_desc = d_lm_desc;
    a.margin = margin;
    for (int l = 0; l < OVS_MAX_LEVELS; ++l) a.sf[l] = l < num_levels ? scale_factors[l] : 1.0f;
    a.mode = kModeProjection;
    a.dead_from = dead_from_ratio(OVS_HAMMING_DIST_THR_HIGH, lowe_ratio);
    st = build_lists(w, a, m, k_window_lists<false>, k_window_lists<true>, s);
    if (st != OVS_OK) return st;
    ResolveArgs ra{};
    ra.offsets = w->d_offsets;
    ra.keys = w->d_keys;
    ra.n_q = m;
    ra.n_t = n;
    ra.lowe_ratio = lowe_ratio;
    ra.assigned = d_assigned;
    ra.num_matches = d_num_matches;
    return launch_resolve<kRuleProjection>(ra, s);
}

ovs_status ovs_projection_match_frame_and_landmarks(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* kps,
                                                    const uint8_t* desc, const float* stereo_x_right, const uint8_t* occupied, int32_t n,
                                                    const float* lm_xy, const float* lm_x_right, const int32_t* lm_level,
                                                    const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m,
                                                    const float* scale_factors, int32_t num_levels, float margin, float lowe_ratio,
                                                    int32_t* assigned, int32_t* num_matches) {
    if (!w || !num_matches || n < 0 || m < 0) return OVS_ERR_INVALID;
    *num_matches = 0;
    if (m == 0) return OVS_OK;
    if (!assigned) return OVS_ERR_INVALID;
    if (n == 0) {
        for (int i = 0; i < m; ++i) assigned[i] = -1;
        return OVS_OK;
    }
    if (!kps || !desc || !lm_xy || !lm_level || !lm_desc) return OVS_ERR_INVALID;
    if (n > w->max_t || m > w->max_q) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    OVS_HIP_TRY(hipMemcpyAsync(w->d_t_kps, kps, sizeof(ovs_keypoint) * n, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_t_desc, desc, (size_t)32 * n, hipMemcpyHostToDevice, s));
    if (stereo_x_right) OVS_HIP_TRY(hipMemcpyAsync(w->d_t_f, stereo_x_right, sizeof(float) * n, hipMemcpyHostToDevice, s));
    if (occupied) OVS_HIP_TRY(hipMemcpyAsync(w->d_t_flag, occupied, (size_t)n, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_xy, lm_xy, sizeof(float) * 2 * m, hipMemcpyHostToDevice, s));
    if (lm_x_right) OVS_HIP_TRY(hipMemcpyAsync(w->d_q_f, lm_x_right, sizeof(float) * m, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_i, lm_level, sizeof(int32_t) * m, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_desc, lm_desc, (size_t)32 * m, hipMemcpyHostToDevice, s));
    if (lm_valid) OVS_HIP_TRY(hipMemcpyAsync(w->d_q_flag, lm_valid, (size_t)m, hipMemcpyHostToDevice, s));
    ovs_status st = ovs_projection_match_frame_and_landmarks_dev(
        w, gp, w->d_t_kps, w->d_t_desc, stereo_x_right ? w->d_t_f : nullptr, occupied ? w->d_t_flag : nullptr, n, w->d_q_xy,
        lm_x_right ? w->d_q_f : nullptr, w->d_q_i, w->d_q_desc, lm_valid ? w->d_q_flag : nullptr, m, scale_factors, num_levels, margin,
        lowe_ratio, w->d_assigned, w->d_num, s);
    if (st != OVS_OK) return st;
    uint32_t overflow = 0;
    OVS_HIP_TRY(hipMemcpyAsync(assigned, w->d_assigned, sizeof(int32_t) * m, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(num_matches, w->d_num, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(&overflow, w->d_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    if (!overflow) w->overflow_dirty = false;
    return overflow ? OVS_ERR_CAPACITY : OVS_OK;
}

ovs_status ovs_area_match_in_consistent_area_dev(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* d_kps_1,
                                                 const uint8_t* d_desc_1, int32_t n1, const ovs_keypoint* d_kps_2,
                                                 const uint8_t* d_desc_2, int32_t n2, float* d_prev_matched_xy,
                                                 int32_t* d_matched_2_in_1, int32_t margin, float lowe_ratio, int32_t check_orientation,
                                                 int32_t* d_num_matches, void* stream) {
    if (!w || n1 < 0 || n2 < 0 || !d_num_matches || (n1 > 0 && (!d_kps_1 || !d_desc_1 || !d_prev_matched_xy || !d_matched_2_in_1)) ||
        (n2 > 0 && (!d_kps_2 || !d_desc_2)))
        return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = (hipStream_t)stream;
    if (n1 == 0) {
        OVS_HIP_TRY(hipMemsetAsync(d_num_matches, 0, sizeof(int32_t), s));
        return OVS_OK;
    }
    ovs_status st = grid_assign(w, gp, d_kps_2, n2, s);
    if (st != OVS_OK) return st;
    WinArgs a{};
    a.t_kps = d_kps_2;
    a.t_desc = d_desc_2;
    a.cell_start = w->d_cell_start;
    a.items = w->d_items;
    a.gp = w->gp;
    a.n_q = n1;
    a.q_xy = d_prev_matched_xy;
    a.q_kps = d_kps_1;
    a.q_desc = d_desc_1;
    a.margin = (float)margin;
    a.mode = kModeArea;
    a.dead_from = dead_from_ratio(OVS_HAMMING_DIST_THR_LOW, lowe_ratio);
    st = build_lists(w, a, n1, k_window_lists<false>, k_window_lists<true>, s);
    if (st != OVS_OK) return st;
    ResolveArgs ra{};
    ra.offsets = w->d_offsets;
    ra.keys = w->d_keys;
    ra.n_q = n1;
    ra.n_t = n2;
    ra.lowe_ratio = lowe_ratio;
    ra.check_orientation = check_orientation;
    ra.q_kps = d_kps_1;
    ra.t_kps = d_kps_2;
    ra.prev_matched_xy = d_prev_matched_xy;
    ra.assigned = d_matched_2_in_1;
    ra.num_matches = d_num_matches;
    return launch_resolve<kRuleArea>(ra, s);
}

ovs_status ovs_area_match_in_consistent_area(ovs_wmatcher* w, const ovs_grid_params* gp, const ovs_keypoint* kps_1, const uint8_t* desc_1,
                                             int32_t n1, const ovs_keypoint* kps_2, const uint8_t* desc_2, int32_t n2,
                                             float* prev_matched_xy, int32_t* matched_2_in_1, int32_t margin, float lowe_ratio,
                                             int32_t check_orientation, int32_t* num_matches) {
    if (!w || !num_matches || n1 < 0 || n2 < 0) return OVS_ERR_INVALID;
    *num_matches = 0;
    if (n1 == 0) return OVS_OK;
    if (!kps_1 || !desc_1 || !prev_matched_xy || !matched_2_in_1) return OVS_ERR_INVALID;
    if (n2 == 0) {
        for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
        return OVS_OK;
    }
    if (!kps_2 || !desc_2) return OVS_ERR_INVALID;
    if (n2 > w->max_t || n1 > w->max_q) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_kps, kps_1, sizeof(ovs_keypoint) * n1, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_desc, desc_1, (size_t)32 * n1, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_xy, prev_matched_xy, sizeof(float) * 2 * n1, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_t_kps, kps_2, sizeof(ovs_keypoint) * n2, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_t_desc, desc_2, (size_t)32 * n2, hipMemcpyHostToDevice, s));
    ovs_status st = ovs_area_match_in_consistent_area_dev(w, gp, w->d_q_kps, w->d_q_desc, n1, w->d_t_kps, w->d_t_desc, n2, w->d_q_xy,
                                                          w->d_assigned, margin, lowe_ratio, check_orientation, w->d_num, s);
    if (st != OVS_OK) return st;
    uint32_t overflow = 0;
    OVS_HIP_TRY(hipMemcpyAsync(matched_2_in_1, w->d_assigned, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(prev_matched_xy, w->d_q_xy, sizeof(float) * 2 * n1, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(num_matches, w->d_num, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(&overflow, w->d_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    if (!overflow) w->overflow_dirty = false;
    return overflow ? OVS_ERR_CAPACITY : OVS_OK;
}

} // extern "C" (helper below has internal linkage)

struct TriParams {   // robust::match_for_triangulation extras (host pointers)
    const float* x_right_1;
    const float* x_right_2;
    const double* bearings_1;
    const double* bearings_2;
    const double* E_12;
    const double* epipole_in_2;
    const float* scale_factors;
    int num_levels;
};

static ovs_status bow_match_impl(ovs_wmatcher* w, const ovs_frame_dev* res_kf, const ovs_frame_dev* res_frm, int by_query, const uint8_t* frm_valid, const TriParams* tri, const ovs_keypoint* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_valid,
                                            int32_t n_kf, const int32_t* kf_node_ids, const int32_t* kf_node_start,
                                            const int32_t* kf_items, int32_t kf_nodes, const ovs_keypoint* frm_kps,
                                            const uint8_t* frm_desc, int32_t n_frm, const int32_t* frm_node_ids,
                                            const int32_t* frm_node_start, const int32_t* frm_items, int32_t frm_nodes, float lowe_ratio,
                                            int32_t check_orientation, int32_t* matched_kf_in_frm, int32_t* num_matches) {
    if (!w || !num_matches || n_kf < 0 || n_frm < 0 || kf_nodes < 0 || frm_nodes < 0) return OVS_ERR_INVALID;
    *num_matches = 0;
    const int n_out = by_query ? n_kf : n_frm;
    if (n_out == 0) return OVS_OK;
    if (!matched_kf_in_frm) return OVS_ERR_INVALID;
    for (int i = 0; i < n_out; ++i) matched_kf_in_frm[i] = -1;
    if (n_frm == 0) return OVS_OK;
    if (n_kf == 0 || kf_nodes == 0 || frm_nodes == 0) return OVS_OK;
    if ((!res_kf && (!kf_kps || !kf_desc)) || !kf_node_ids || !kf_node_start || !kf_items || (!res_frm && (!frm_kps || !frm_desc)) || !frm_node_ids ||
        !frm_node_start || !frm_items)
        return OVS_ERR_INVALID;
    if ((res_kf && (res_kf->device != w->device || res_kf->n != n_kf)) || (res_frm && (res_frm->device != w->device || res_frm->n != n_frm))) return OVS_ERR_INVALID;
    const int nq = kf_node_start[kf_nodes], nfi = frm_node_start[frm_nodes];
    if (nq == 0 || nfi == 0) return OVS_OK;
    if (n_frm > w->max_t || n_kf > w->max_q || nq > w->max_q || (by_query && n_kf > std::max(w->max_t, w->max_q))) return OVS_ERR_CAPACITY;
    const size_t need = (size_t)2 * kf_nodes + 1 + nq + (size_t)2 * frm_nodes + 1 + nfi;
    if (need > w->csr_cap) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    int32_t* p = w->d_csr;
    int32_t* d_kf_ids = p;          p += kf_nodes;
    int32_t* d_kf_start = p;        p += kf_nodes + 1;
    int32_t* d_kf_items = p;        p += nq;
    int32_t* d_f_ids = p;           p += frm_nodes;
    int32_t* d_f_start = p;         p += frm_nodes + 1;
    int32_t* d_f_items = p;
    OVS_HIP_TRY(hipMemcpyAsync(d_kf_ids, kf_node_ids, sizeof(int32_t) * kf_nodes, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(d_kf_start, kf_node_start, sizeof(int32_t) * (kf_nodes + 1), hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(d_kf_items, kf_items, sizeof(int32_t) * nq, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(d_f_ids, frm_node_ids, sizeof(int32_t) * frm_nodes, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(d_f_start, frm_node_start, sizeof(int32_t) * (frm_nodes + 1), hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(d_f_items, frm_items, sizeof(int32_t) * nfi, hipMemcpyHostToDevice, s));
    // keypoints (angle, octave) and descriptors of either side: resident (a frame / keyframe handle) or uploaded per call
    const ovs_keypoint* d_kf_kps = w->d_q_kps;
    const uint8_t* d_kf_desc = w->d_q_desc;
    const ovs_keypoint* d_frm_kps = w->d_t_kps;
    const uint8_t* d_frm_desc = w->d_t_desc;
    if (res_kf) {
        d_kf_kps = res_kf->d_kps;
        d_kf_desc = res_kf->d_desc;
    } else {
        OVS_HIP_TRY(hipMemcpyAsync(w->d_q_kps, kf_kps, sizeof(ovs_keypoint) * n_kf, hipMemcpyHostToDevice, s));
        OVS_HIP_TRY(hipMemcpyAsync(w->d_q_desc, kf_desc, (size_t)32 * n_kf, hipMemcpyHostToDevice, s));
    }
    if (kf_valid) OVS_HIP_TRY(hipMemcpyAsync(w->d_q_flag, kf_valid, (size_t)n_kf, hipMemcpyHostToDevice, s));
    if (res_frm) {
        d_frm_kps = res_frm->d_kps;
        d_frm_desc = res_frm->d_desc;
    } else {
        OVS_HIP_TRY(hipMemcpyAsync(w->d_t_kps, frm_kps, sizeof(ovs_keypoint) * n_frm, hipMemcpyHostToDevice, s));
        OVS_HIP_TRY(hipMemcpyAsync(w->d_t_desc, frm_desc, (size_t)32 * n_frm, hipMemcpyHostToDevice, s));
    }
    if (frm_valid) OVS_HIP_TRY(hipMemcpyAsync(w->d_t_flag, frm_valid, (size_t)n_frm, hipMemcpyHostToDevice, s));
    BowArgs a{};
    a.kf_desc = d_kf_desc;
    a.kf_valid = kf_valid ? w->d_q_flag : nullptr;
    a.kf_node_ids = d_kf_ids;
    a.kf_node_start = d_kf_start;
    a.kf_items = d_kf_items;
    a.kf_nodes = kf_nodes;
    a.n_q = nq;
    a.frm_desc = d_frm_desc;
    a.frm_valid = frm_valid ? w->d_t_flag : nullptr;
    if (tri) {
        a.tri = 1;
        a.kf_kps = d_kf_kps;
        // bearings and stereo_x_right: from the handle when it carries them (ovs_frame_dev_attach_bearings), else uploaded
        if (res_kf && res_kf->d_bearings) {
            a.kf_bearings = res_kf->d_bearings;
        } else {
            if (!tri->bearings_1) return OVS_ERR_INVALID;
            OVS_HIP_TRY(hipMemcpyAsync(w->d_tri_b1, tri->bearings_1, sizeof(double) * 3 * n_kf, hipMemcpyHostToDevice, s));
            a.kf_bearings = w->d_tri_b1;
        }
        if (res_frm && res_frm->d_bearings) {
            a.frm_bearings = res_frm->d_bearings;
        } else {
            if (!tri->bearings_2) return OVS_ERR_INVALID;
            OVS_HIP_TRY(hipMemcpyAsync(w->d_tri_b2, tri->bearings_2, sizeof(double) * 3 * n_frm, hipMemcpyHostToDevice, s));
            a.frm_bearings = w->d_tri_b2;
        }
        if (res_kf) {
            a.kf_x_right = res_kf->has_stereo ? res_kf->d_x_right : nullptr;
        } else if (tri->x_right_1) {
            OVS_HIP_TRY(hipMemcpyAsync(w->d_q_f, tri->x_right_1, sizeof(float) * n_kf, hipMemcpyHostToDevice, s));
            a.kf_x_right = w->d_q_f;
        }
        if (res_frm) {
            a.frm_x_right = res_frm->has_stereo ? res_frm->d_x_right : nullptr;
        } else if (tri->x_right_2) {
            OVS_HIP_TRY(hipMemcpyAsync(w->d_t_f, tri->x_right_2, sizeof(float) * n_frm, hipMemcpyHostToDevice, s));
            a.frm_x_right = w->d_t_f;
        }
        std::memcpy(a.E, tri->E_12, sizeof(double) * 9);
        std::memcpy(a.epipole, tri->epipole_in_2, sizeof(double) * 3);
        for (int l = 0; l < OVS_MAX_LEVELS; ++l) a.sf[l] = l < tri->num_levels ? tri->scale_factors[l] : 1.0f;
    }
    a.frm_node_ids = d_f_ids;
    a.frm_node_start = d_f_start;
    a.frm_items = d_f_items;
    a.frm_nodes = frm_nodes;
    a.dead_from = tri ? 0u : dead_from_ratio(OVS_HAMMING_DIST_THR_LOW, lowe_ratio);
    ovs_status st = build_lists(w, a, nq, k_bow_lists<false>, k_bow_lists<true>, s, 256);
    if (st != OVS_OK) return st;
    ResolveArgs ra{};
    ra.offsets = w->d_offsets;
    ra.keys = w->d_keys;
    ra.n_q = nq;
    ra.n_t = n_frm;
    ra.lowe_ratio = lowe_ratio;
    ra.check_orientation = check_orientation;
    ra.q_kps = d_kf_kps;
    ra.q_items = d_kf_items;
    ra.t_kps = d_frm_kps;
    ra.assigned = w->d_assigned;
    ra.num_matches = w->d_num;
    ra.bow_by_query = by_query;
    ra.n_out_q = n_kf;
    ra.best_only_thr = OVS_HAMMING_DIST_THR_LOW;
    st = tri ? launch_resolve<kRuleTriang>(ra, s) : launch_resolve<kRuleBow>(ra, s);
    if (st != OVS_OK) return st;
    uint32_t overflow = 0;
    OVS_HIP_TRY(hipMemcpyAsync(matched_kf_in_frm, w->d_assigned, sizeof(int32_t) * n_out, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(num_matches, w->d_num, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(&overflow, w->d_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    if (!overflow) w->overflow_dirty = false;
    return overflow ? OVS_ERR_CAPACITY : OVS_OK;
}


extern "C" {

ovs_status ovs_bow_match_frame_and_keyframe(ovs_wmatcher* w, const ovs_keypoint* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_valid,
                                            int32_t n_kf, const int32_t* kf_node_ids, const int32_t* kf_node_start,
                                            const int32_t* kf_items, int32_t kf_nodes, const ovs_keypoint* frm_kps,
                                            const uint8_t* frm_desc, int32_t n_frm, const int32_t* frm_node_ids,
                                            const int32_t* frm_node_start, const int32_t* frm_items, int32_t frm_nodes, float lowe_ratio,
                                            int32_t check_orientation, int32_t* matched_kf_in_frm, int32_t* num_matches) {
    return bow_match_impl(w, nullptr, nullptr, 0, nullptr, nullptr, kf_kps, kf_desc, kf_valid, n_kf, kf_node_ids, kf_node_start, kf_items, kf_nodes, frm_kps, frm_desc, n_frm,
                          frm_node_ids, frm_node_start, frm_items, frm_nodes, lowe_ratio, check_orientation, matched_kf_in_frm, num_matches);
}

ovs_status ovs_bow_match_keyframes(ovs_wmatcher* w, const ovs_keypoint* kps_1, const uint8_t* desc_1, const uint8_t* valid_1, int32_t n1,
                                   const int32_t* node_ids_1, const int32_t* node_start_1, const int32_t* items_1, int32_t nodes_1,
                                   const ovs_keypoint* kps_2, const uint8_t* desc_2, const uint8_t* valid_2, int32_t n2,
                                   const int32_t* node_ids_2, const int32_t* node_start_2, const int32_t* items_2, int32_t nodes_2,
                                   float lowe_ratio, int32_t check_orientation, int32_t* matched_2_in_1, int32_t* num_matches) {
    return bow_match_impl(w, nullptr, nullptr, 1, valid_2, nullptr, kps_1, desc_1, valid_1, n1, node_ids_1, node_start_1, items_1, nodes_1, kps_2, desc_2, n2, node_ids_2,
                          node_start_2, items_2, nodes_2, lowe_ratio, check_orientation, matched_2_in_1, num_matches);
}

ovs_status ovs_robust_match_for_triangulation(ovs_wmatcher* w, const ovs_keypoint* kps_1, const uint8_t* desc_1, const uint8_t* has_lm_1,
                                              const float* x_right_1, const double* bearings_1, int32_t n1, const int32_t* node_ids_1,
                                              const int32_t* node_start_1, const int32_t* items_1, int32_t nodes_1,
                                              const ovs_keypoint* kps_2, const uint8_t* desc_2, const uint8_t* has_lm_2,
                                              const float* x_right_2, const double* bearings_2, int32_t n2, const int32_t* node_ids_2,
                                              const int32_t* node_start_2, const int32_t* items_2, int32_t nodes_2, const double* E_12,
                                              const double* epipole_in_2, const float* scale_factors, int32_t num_levels,
                                              int32_t check_orientation, int32_t* matched_2_in_1, int32_t* num_matches) {
    if (!bearings_1 || !bearings_2 || !E_12 || !epipole_in_2 || !scale_factors || num_levels < 1 || num_levels > OVS_MAX_LEVELS || n1 < 0 || n2 < 0)
        return OVS_ERR_INVALID;
    // "valid" for the bow kernels = the keypoint has NO landmark yet
    std::vector<uint8_t> v1((size_t)std::max(n1, 1), 1), v2((size_t)std::max(n2, 1), 1);
    if (has_lm_1)
        for (int i = 0; i < n1; ++i) v1[i] = has_lm_1[i] ? 0 : 1;
    if (has_lm_2)
        for (int i = 0; i < n2; ++i) v2[i] = has_lm_2[i] ? 0 : 1;
    TriParams tp{x_right_1, x_right_2, bearings_1, bearings_2, E_12, epipole_in_2, scale_factors, num_levels};
    return bow_match_impl(w, nullptr, nullptr, 1, v2.data(), &tp, kps_1, desc_1, v1.data(), n1, node_ids_1, node_start_1, items_1, nodes_1, kps_2, desc_2, n2,
                          node_ids_2, node_start_2, items_2, nodes_2, 0.0f, check_orientation, matched_2_in_1, num_matches);
}

ovs_status ovs_projection_match_current_and_last_frames(ovs_wmatcher* w, const ovs_camera* cam, const ovs_grid_params* gp,
                                                        const ovs_keypoint* curr_kps, const uint8_t* curr_desc,
                                                        const float* curr_stereo_x_right, const uint8_t* curr_occupied, int32_t n_curr,
                                                        const double* pose_cw_curr, const ovs_keypoint* last_kps, const double* last_pos_w,
                                                        const uint8_t* last_lm_desc, const uint8_t* last_valid, int32_t n_last,
                                                        const double* pose_cw_last, const float* scale_factors, int32_t num_levels,
                                                        float margin, int32_t check_orientation, int32_t* assigned, int32_t* num_matches) {
    if (!w || !cam || !gp || !num_matches || n_curr < 0 || n_last < 0 || !pose_cw_curr || !pose_cw_last || !scale_factors || num_levels < 1 ||
        num_levels > OVS_MAX_LEVELS || (cam->model != 0 && cam->model != 1))
        return OVS_ERR_INVALID;
    *num_matches = 0;
    if (n_last == 0) return OVS_OK;
    if (!assigned) return OVS_ERR_INVALID;
    for (int i = 0; i < n_last; ++i) assigned[i] = -1;
    if (n_curr == 0) return OVS_OK;
    if (!curr_kps || !curr_desc || !last_kps || !last_pos_w || !last_lm_desc) return OVS_ERR_INVALID;
    if (n_curr > w->max_t || n_last > w->max_q) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    // motion direction (host, double): trans_wc = -rot_cw^T trans_cw; trans_lc = rot_lw trans_wc + trans_lw
    const double* Rc = pose_cw_curr;
    const double* tc = pose_cw_curr + 9;
    const double twc[3] = {-((Rc[0] * tc[0] + Rc[3] * tc[1]) + Rc[6] * tc[2]), -((Rc[1] * tc[0] + Rc[4] * tc[1]) + Rc[7] * tc[2]),
                           -((Rc[2] * tc[0] + Rc[5] * tc[1]) + Rc[8] * tc[2])};
    const double* Rl = pose_cw_last;
    const double tlc_z = ((Rl[6] * twc[0] + Rl[7] * twc[1]) + Rl[8] * twc[2]) + pose_cw_last[11];
    const int forward = cam->setup == 0 ? 0 : (tlc_z > cam->true_baseline);
    const int backward = cam->setup == 0 ? 0 : (-tlc_z > cam->true_baseline);
    CamP cp{};
    cp.model = cam->model;
    cp.setup = cam->setup;
    cp.fx = cam->fx;
    cp.fy = cam->fy;
    cp.cx = cam->cx;
    cp.cy = cam->cy;
    cp.fxb = cam->focal_x_baseline;
    cp.cols = cam->cols;
    cp.rows = cam->rows;
    cp.min_x = gp->min_x;
    cp.min_y = gp->min_y;
    cp.max_x = gp->max_x;
    cp.max_y = gp->max_y;
    std::memcpy(cp.P, pose_cw_curr, sizeof(double) * 12);
    OVS_HIP_TRY(hipMemcpyAsync(w->d_t_kps, curr_kps, sizeof(ovs_keypoint) * n_curr, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_t_desc, curr_desc, (size_t)32 * n_curr, hipMemcpyHostToDevice, s));
    if (curr_stereo_x_right) OVS_HIP_TRY(hipMemcpyAsync(w->d_t_f, curr_stereo_x_right, sizeof(float) * n_curr, hipMemcpyHostToDevice, s));
    if (curr_occupied) OVS_HIP_TRY(hipMemcpyAsync(w->d_t_flag, curr_occupied, (size_t)n_curr, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_kps, last_kps, sizeof(ovs_keypoint) * n_last, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_pos, last_pos_w, sizeof(double) * 3 * n_last, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_desc, last_lm_desc, (size_t)32 * n_last, hipMemcpyHostToDevice, s));
    uint8_t* d_last_valid = nullptr;
    if (last_valid) {   // staged behind the query flags (the kernel writes q_valid in place: same index, read-then-write by one lane)
        OVS_HIP_TRY(hipMemcpyAsync(w->d_q_flag, last_valid, (size_t)n_last, hipMemcpyHostToDevice, s));
        d_last_valid = w->d_q_flag;
    }
    float sf16[OVS_MAX_LEVELS];
    for (int l = 0; l < OVS_MAX_LEVELS; ++l) sf16[l] = l < num_levels ? scale_factors[l] : 1.0f;
    OVS_HIP_TRY(hipMemcpyAsync(w->d_sf, sf16, sizeof(sf16), hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));   // sf16 is a stack array
    ovs_status st = grid_assign(w, gp, w->d_t_kps, n_curr, s);
    if (st != OVS_OK) return st;
    hipLaunchKernelGGL(k_reproject_queries, dim3((n_last + 255) / 256), dim3(256), 0, s, cp, (const ovs_keypoint*)w->d_q_kps,
                       (const double*)w->d_q_pos, (const uint8_t*)d_last_valid, n_last, margin, (const float*)w->d_sf, num_levels, forward,
                       backward, (const float*)nullptr, 0.0, 0.0, 0.0, 0.0f, w->d_q_xy, w->d_q_f, w->d_q_r, w->d_q_i, w->d_q_i2, w->d_q_flag);
    OVS_HIP_TRY(hipGetLastError());
    WinArgs a{};
    a.t_kps = w->d_t_kps;
    a.t_desc = w->d_t_desc;
    a.t_occupied = curr_occupied ? w->d_t_flag : nullptr;
    a.t_x_right = curr_stereo_x_right ? w->d_t_f : nullptr;
    a.cell_start = w->d_cell_start;
    a.items = w->d_items;
    a.gp = w->gp;
    a.n_q = n_last;
    a.q_xy = w->d_q_xy;
    a.q_x_right = w->d_q_f;
    a.q_valid = w->d_q_flag;
    a.q_radius = w->d_q_r;
    a.q_minl = w->d_q_i;
    a.q_maxl = w->d_q_i2;
    a.q_desc = w->d_q_desc;
    a.margin = margin;
    a.mode = kModeGeneric;
    a.dead_from = dead_from_thr(OVS_HAMMING_DIST_THR_HIGH);
    st = build_lists(w, a, n_last, k_window_lists<false>, k_window_lists<true>, s);
    if (st != OVS_OK) return st;
    ResolveArgs ra{};
    ra.offsets = w->d_offsets;
    ra.keys = w->d_keys;
    ra.n_q = n_last;
    ra.n_t = n_curr;
    ra.check_orientation = check_orientation;
    ra.q_kps = w->d_q_kps;
    ra.t_kps = w->d_t_kps;
    ra.assigned = w->d_assigned;
    ra.num_matches = w->d_num;
    ra.best_only_thr = OVS_HAMMING_DIST_THR_HIGH;
    st = launch_resolve<kRuleBestOnly>(ra, s);
    if (st != OVS_OK) return st;
    uint32_t overflow = 0;
    OVS_HIP_TRY(hipMemcpyAsync(assigned, w->d_assigned, sizeof(int32_t) * n_last, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(num_matches, w->d_num, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(&overflow, w->d_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    if (!overflow) w->overflow_dirty = false;
    return overflow ? OVS_ERR_CAPACITY : OVS_OK;
}

static ovs_status fuse_replace_duplication_impl(ovs_wmatcher* w, const ovs_frame_dev* res, const ovs_camera* cam, const ovs_grid_params* gp, const ovs_keypoint* kps,
                                        const uint8_t* desc, const float* stereo_x_right, int32_t n, const double* pose_cw,
                                        const double* lm_pos_w, const float* lm_dist_min_max, const double* lm_normal,
                                        const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m, const float* scale_factors,
                                        const float* inv_level_sigma_sq, int32_t num_levels, float log_scale_factor, float margin,
                                        int32_t* best_idx, int32_t* num_fused) {
    if (!w || !cam || !gp || !num_fused || n < 0 || m < 0 || !pose_cw || !scale_factors || !inv_level_sigma_sq || num_levels < 1 ||
        num_levels > OVS_MAX_LEVELS || (cam->model != 0 && cam->model != 1))
        return OVS_ERR_INVALID;
    *num_fused = 0;
    if (m == 0) return OVS_OK;
    if (!best_idx) return OVS_ERR_INVALID;
    for (int i = 0; i < m; ++i) best_idx[i] = -1;
    if (n == 0) return OVS_OK;
    if ((!res && (!kps || !desc)) || !lm_pos_w || !lm_dist_min_max || !lm_normal || !lm_desc) return OVS_ERR_INVALID;
    if (n > w->max_t || m > w->max_q) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    FuseArgs a{};
    a.cam.model = cam->model;
    a.cam.setup = cam->setup;
    a.cam.fx = cam->fx;
    a.cam.fy = cam->fy;
    a.cam.cx = cam->cx;
    a.cam.cy = cam->cy;
    a.cam.fxb = cam->focal_x_baseline;
    a.cam.cols = cam->cols;
    a.cam.rows = cam->rows;
    a.cam.min_x = gp->min_x;
    a.cam.min_y = gp->min_y;
    a.cam.max_x = gp->max_x;
    a.cam.max_y = gp->max_y;
    std::memcpy(a.cam.P, pose_cw, sizeof(double) * 12);
    const double* R = pose_cw;
    const double* t = pose_cw + 9;
    a.cc[0] = -((R[0] * t[0] + R[3] * t[1]) + R[6] * t[2]);
    a.cc[1] = -((R[1] * t[0] + R[4] * t[1]) + R[7] * t[2]);
    a.cc[2] = -((R[2] * t[0] + R[5] * t[1]) + R[8] * t[2]);
    for (int l = 0; l < OVS_MAX_LEVELS; ++l) {
        a.sf[l] = l < num_levels ? scale_factors[l] : 1.0f;
        a.ils[l] = l < num_levels ? inv_level_sigma_sq[l] : 1.0f;
    }
    a.num_levels = num_levels;
    a.log_scale_factor = log_scale_factor;
    a.margin = margin;
    a.m = m;
    // the landmark side: five host arrays through the context's pinned buffer, ONE copy up (five pageable hipMemcpyAsync cost more than the kernel)
    Stager stg(w);
    a.lm_pos_w = stg.put(lm_pos_w, (size_t)3 * m);
    a.lm_normal = stg.put(lm_normal, (size_t)3 * m);
    a.lm_dist = stg.put(lm_dist_min_max, (size_t)2 * m);
    a.lm_desc = stg.put(lm_desc, (size_t)32 * m);
    a.lm_valid = stg.put(lm_valid, (size_t)m);
    if (stg.overflow) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(stg.flush(s));
    TargetRef tg;
    ovs_status st = stage_target(w, res, gp, kps, desc, stereo_x_right, n, s, &tg);
    if (st != OVS_OK) return st;
    a.t_kps = tg.kps;
    a.t_desc = tg.desc;
    a.t_x_right = tg.x_right;
    a.cell_start = tg.cell_start;
    a.items = tg.items;
    a.gp = tg.gp;
    a.variant = kFuseReplace;
    a.max_dist = OVS_HAMMING_DIST_THR_LOW;
    OVS_HIP_TRY(hipMemsetAsync(w->d_num, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_fuse_best, dim3((unsigned)(((size_t)m * kFuseLanes + 255) / 256)), dim3(256), 0, s, a, w->d_assigned, w->d_num);
    OVS_HIP_TRY(hipGetLastError());
    return fetch_results(w, m, best_idx, num_fused, s, false);   // one copy down through the pinned mirror
}

static ovs_status projection_match_frame_and_keyframe_impl(ovs_wmatcher* w, const ovs_frame_dev* res, const ovs_camera* cam, const ovs_grid_params* gp,
                                                   const ovs_keypoint* curr_kps, const uint8_t* curr_desc, const uint8_t* curr_occupied,
                                                   int32_t n_curr, const double* pose_cw_curr, const ovs_keypoint* kf_kps,
                                                   const double* kf_pos_w, const float* kf_dist_min_max, const uint8_t* kf_lm_desc,
                                                   const uint8_t* kf_valid, int32_t n_kf, const float* scale_factors, int32_t num_levels,
                                                   float log_scale_factor, float margin, uint32_t hamm_dist_thr, int32_t check_orientation,
                                                   int32_t* assigned, int32_t* num_matches) {
    if (!w || !cam || !gp || !num_matches || n_curr < 0 || n_kf < 0 || !pose_cw_curr || !scale_factors || num_levels < 1 ||
        num_levels > OVS_MAX_LEVELS || (cam->model != 0 && cam->model != 1))
        return OVS_ERR_INVALID;
    *num_matches = 0;
    if (n_kf == 0) return OVS_OK;
    if (!assigned) return OVS_ERR_INVALID;
    for (int i = 0; i < n_kf; ++i) assigned[i] = -1;
    if (n_curr == 0) return OVS_OK;
    if ((!res && (!curr_kps || !curr_desc)) || !kf_kps || !kf_pos_w || !kf_dist_min_max || !kf_lm_desc) return OVS_ERR_INVALID;
    if (n_curr > w->max_t || n_kf > w->max_q) return OVS_ERR_CAPACITY;
    if ((size_t)n_kf * 2 * sizeof(float) > (size_t)w->max_entries * sizeof(uint32_t)) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    CamP cp{};
    cp.model = cam->model;
    cp.setup = cam->setup;
    cp.fx = cam->fx;
    cp.fy = cam->fy;
    cp.cx = cam->cx;
    cp.cy = cam->cy;
    cp.fxb = cam->focal_x_baseline;
    cp.cols = cam->cols;
    cp.rows = cam->rows;
    cp.min_x = gp->min_x;
    cp.min_y = gp->min_y;
    cp.max_x = gp->max_x;
    cp.max_y = gp->max_y;
    std::memcpy(cp.P, pose_cw_curr, sizeof(double) * 12);
    const double* R = pose_cw_curr;
    const double* t = pose_cw_curr + 9;
    const double ccx = -((R[0] * t[0] + R[3] * t[1]) + R[6] * t[2]), ccy = -((R[1] * t[0] + R[4] * t[1]) + R[7] * t[2]),
                 ccz = -((R[2] * t[0] + R[5] * t[1]) + R[8] * t[2]);
    float* d_dist = reinterpret_cast<float*>(w->d_keys);   // consumed by k_reproject_queries before the key buffer is written
    if (curr_occupied) OVS_HIP_TRY(hipMemcpyAsync(w->d_t_flag, curr_occupied, (size_t)n_curr, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_kps, kf_kps, sizeof(ovs_keypoint) * n_kf, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_pos, kf_pos_w, sizeof(double) * 3 * n_kf, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(d_dist, kf_dist_min_max, sizeof(float) * 2 * n_kf, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_desc, kf_lm_desc, (size_t)32 * n_kf, hipMemcpyHostToDevice, s));
    uint8_t* d_valid = nullptr;
    if (kf_valid) {
        OVS_HIP_TRY(hipMemcpyAsync(w->d_q_flag, kf_valid, (size_t)n_kf, hipMemcpyHostToDevice, s));
        d_valid = w->d_q_flag;
    }
    float sf16[OVS_MAX_LEVELS];
    for (int l = 0; l < OVS_MAX_LEVELS; ++l) sf16[l] = l < num_levels ? scale_factors[l] : 1.0f;
    OVS_HIP_TRY(hipMemcpyAsync(w->d_sf, sf16, sizeof(sf16), hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));   // sf16 is a stack array
    TargetRef tg;
    ovs_status st = stage_target(w, res, gp, curr_kps, curr_desc, nullptr, n_curr, s, &tg);
    if (st != OVS_OK) return st;
    hipLaunchKernelGGL(k_reproject_queries, dim3((n_kf + 255) / 256), dim3(256), 0, s, cp, (const ovs_keypoint*)w->d_q_kps,
                       (const double*)w->d_q_pos, (const uint8_t*)d_valid, n_kf, margin, (const float*)w->d_sf, num_levels, 0, 0,
                       (const float*)d_dist, ccx, ccy, ccz, log_scale_factor, w->d_q_xy, w->d_q_f, w->d_q_r, w->d_q_i, w->d_q_i2, w->d_q_flag);
    OVS_HIP_TRY(hipGetLastError());
    WinArgs a{};
    a.t_kps = tg.kps;
    a.t_desc = tg.desc;
    a.t_occupied = curr_occupied ? w->d_t_flag : nullptr;
    a.cell_start = tg.cell_start;
    a.items = tg.items;
    a.gp = tg.gp;
    a.n_q = n_kf;
    a.q_xy = w->d_q_xy;
    a.q_x_right = w->d_q_f;
    a.q_valid = w->d_q_flag;
    a.q_radius = w->d_q_r;
    a.q_minl = w->d_q_i;
    a.q_maxl = w->d_q_i2;
    a.q_desc = w->d_q_desc;
    a.margin = margin;
    a.mode = kModeGeneric;
    a.dead_from = dead_from_thr((uint32_t)hamm_dist_thr);
    st = build_lists(w, a, n_kf, k_window_lists<false>, k_window_lists<true>, s);
    if (st != OVS_OK) return st;
    ResolveArgs ra{};
    ra.offsets = w->d_offsets;
    ra.keys = w->d_keys;
    ra.n_q = n_kf;
    ra.n_t = n_curr;
    ra.check_orientation = check_orientation;
    ra.q_kps = w->d_q_kps;
    ra.t_kps = tg.kps;
    ra.assigned = w->d_assigned;
    ra.num_matches = w->d_num;
    ra.best_only_thr = hamm_dist_thr;
    st = launch_resolve<kRuleBestOnly>(ra, s);
    if (st != OVS_OK) return st;
    uint32_t overflow = 0;
    OVS_HIP_TRY(hipMemcpyAsync(assigned, w->d_assigned, sizeof(int32_t) * n_kf, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(num_matches, w->d_num, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(&overflow, w->d_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    if (!overflow) w->overflow_dirty = false;
    return overflow ? OVS_ERR_CAPACITY : OVS_OK;
}

namespace {
void fill_cam(CamP& cp, const ovs_camera* cam, const ovs_grid_params* gp, const double* P) {
    cp.model = cam->model;
    cp.setup = cam->setup;
    cp.fx = cam->fx;
    cp.fy = cam->fy;
    cp.cx = cam->cx;
    cp.cy = cam->cy;
    cp.fxb = cam->focal_x_baseline;
    cp.cols = cam->cols;
    cp.rows = cam->rows;
    cp.min_x = gp->min_x;
    cp.min_y = gp->min_y;
    cp.max_x = gp->max_x;
    cp.max_y = gp->max_y;
    std::memcpy(cp.P, P, sizeof(double) * 12);
}
// Sim3_cw = [s R | t'] -> scale s = |first row of sR|, rot_cw = sR / s, trans_cw = t' / s, camera centre -R^T t (upstream decomposes the
// 4x4 the same way in fuse::detect_duplication / projection::match_by_Sim3_transform)
void decompose_sim3(const double* S, double* P, double* cc) {
    const double sc = std::sqrt((S[0] * S[0] + S[1] * S[1]) + S[2] * S[2]);
    for (int i = 0; i < 9; ++i) P[i] = S[i] / sc;
    for (int i = 0; i < 3; ++i) P[9 + i] = S[9 + i] / sc;
    cc[0] = -((P[0] * P[9] + P[3] * P[10]) + P[6] * P[11]);
    cc[1] = -((P[1] * P[9] + P[4] * P[10]) + P[7] * P[11]);
    cc[2] = -((P[2] * P[9] + P[5] * P[10]) + P[8] * P[11]);
}
}   // namespace

static ovs_status fuse_detect_duplication_impl(ovs_wmatcher* w, const ovs_frame_dev* res, const ovs_camera* cam, const ovs_grid_params* gp, const ovs_keypoint* kps,
                                       const uint8_t* desc, int32_t n, const double* sim3_cw, const double* lm_pos_w,
                                       const float* lm_dist_min_max, const double* lm_normal, const uint8_t* lm_desc, const uint8_t* lm_valid,
                                       int32_t m, const float* scale_factors, int32_t num_levels, float log_scale_factor, float margin,
                                       int32_t* best_idx, int32_t* num_found) {
    if (!w || !cam || !gp || !num_found || n < 0 || m < 0 || !sim3_cw || !scale_factors || num_levels < 1 || num_levels > OVS_MAX_LEVELS ||
        (cam->model != 0 && cam->model != 1))
        return OVS_ERR_INVALID;
    *num_found = 0;
    if (m == 0) return OVS_OK;
    if (!best_idx) return OVS_ERR_INVALID;
    for (int i = 0; i < m; ++i) best_idx[i] = -1;
    if (n == 0) return OVS_OK;
    if ((!res && (!kps || !desc)) || !lm_pos_w || !lm_dist_min_max || !lm_normal || !lm_desc) return OVS_ERR_INVALID;
    if (n > w->max_t || m > w->max_q) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    FuseArgs a{};
    double P[12];
    decompose_sim3(sim3_cw, P, a.cc);
    fill_cam(a.cam, cam, gp, P);
    for (int l = 0; l < OVS_MAX_LEVELS; ++l) {
        a.sf[l] = l < num_levels ? scale_factors[l] : 1.0f;
        a.ils[l] = 1.0f;
    }
    a.num_levels = num_levels;
    a.log_scale_factor = log_scale_factor;
    a.margin = margin;
    a.m = m;
    a.variant = kFuseDetect;
    a.max_dist = OVS_HAMMING_DIST_THR_LOW;
    Stager stg(w);   // the landmark side through the pinned buffer: one copy up
    a.lm_pos_w = stg.put(lm_pos_w, (size_t)3 * m);
    a.lm_normal = stg.put(lm_normal, (size_t)3 * m);
    a.lm_dist = stg.put(lm_dist_min_max, (size_t)2 * m);
    a.lm_desc = stg.put(lm_desc, (size_t)32 * m);
    a.lm_valid = stg.put(lm_valid, (size_t)m);
    if (stg.overflow) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(stg.flush(s));
    TargetRef tg;
    ovs_status st = stage_target(w, res, gp, kps, desc, nullptr, n, s, &tg);
    if (st != OVS_OK) return st;
    a.t_kps = tg.kps;
    a.t_desc = tg.desc;
    a.cell_start = tg.cell_start;
    a.items = tg.items;
    a.gp = tg.gp;
    OVS_HIP_TRY(hipMemsetAsync(w->d_num, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_fuse_best, dim3((unsigned)(((size_t)m * kFuseLanes + 255) / 256)), dim3(256), 0, s, a, w->d_assigned, w->d_num);
    OVS_HIP_TRY(hipGetLastError());
    return fetch_results(w, m, best_idx, num_found, s, false);
}

static ovs_status projection_match_by_sim3_transform_impl(ovs_wmatcher* w, const ovs_frame_dev* res, const ovs_camera* cam, const ovs_grid_params* gp, const ovs_keypoint* kps,
                                                  const uint8_t* desc, const uint8_t* occupied, int32_t n, const double* sim3_cw,
                                                  const double* lm_pos_w, const float* lm_dist_min_max, const double* lm_normal,
                                                  const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m, const float* scale_factors,
                                                  int32_t num_levels, float log_scale_factor, float margin, int32_t* assigned,
                                                  int32_t* num_matches) {
    if (!w || !cam || !gp || !num_matches || n < 0 || m < 0 || !sim3_cw || !scale_factors || num_levels < 1 || num_levels > OVS_MAX_LEVELS ||
        (cam->model != 0 && cam->model != 1))
        return OVS_ERR_INVALID;
    *num_matches = 0;
    if (m == 0) return OVS_OK;
    if (!assigned) return OVS_ERR_INVALID;
    for (int i = 0; i < m; ++i) assigned[i] = -1;
    if (n == 0) return OVS_OK;
    if ((!res && (!kps || !desc)) || !lm_pos_w || !lm_dist_min_max || !lm_normal || !lm_desc) return OVS_ERR_INVALID;
    if (n > w->max_t || m > w->max_q) return OVS_ERR_CAPACITY;
    // (min, max) distances then the normals ride in the key buffer; both are consumed before the lists are written
    if ((size_t)m * (2 * sizeof(float) + 3 * sizeof(double)) > (size_t)w->max_entries * sizeof(uint32_t)) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    CamP cp{};
    double P[12], cc[3];
    decompose_sim3(sim3_cw, P, cc);
    fill_cam(cp, cam, gp, P);
    float* d_dist = reinterpret_cast<float*>(w->d_keys);
    double* d_normal = reinterpret_cast<double*>(d_dist + 2 * (size_t)m);
    if (occupied) OVS_HIP_TRY(hipMemcpyAsync(w->d_t_flag, occupied, (size_t)n, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_pos, lm_pos_w, sizeof(double) * 3 * m, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(d_dist, lm_dist_min_max, sizeof(float) * 2 * m, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(d_normal, lm_normal, sizeof(double) * 3 * m, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_desc, lm_desc, (size_t)32 * m, hipMemcpyHostToDevice, s));
    uint8_t* d_valid = nullptr;
    if (lm_valid) {
        OVS_HIP_TRY(hipMemcpyAsync(w->d_q_flag, lm_valid, (size_t)m, hipMemcpyHostToDevice, s));
        d_valid = w->d_q_flag;
    }
    float sf16[OVS_MAX_LEVELS];
    for (int l = 0; l < OVS_MAX_LEVELS; ++l) sf16[l] = l < num_levels ? scale_factors[l] : 1.0f;
    OVS_HIP_TRY(hipMemcpyAsync(w->d_sf, sf16, sizeof(sf16), hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));   // sf16 is a stack array
    TargetRef tg;
    ovs_status st = stage_target(w, res, gp, kps, desc, nullptr, n, s, &tg);
    if (st != OVS_OK) return st;
    hipLaunchKernelGGL(k_reproject_queries, dim3((m + 255) / 256), dim3(256), 0, s, cp, (const ovs_keypoint*)nullptr, (const double*)w->d_q_pos,
                       (const uint8_t*)d_valid, m, margin, (const float*)w->d_sf, num_levels, 0, 0, (const float*)d_dist, cc[0], cc[1], cc[2],
                       log_scale_factor, w->d_q_xy, w->d_q_f, w->d_q_r, w->d_q_i, w->d_q_i2, w->d_q_flag, (const double*)d_normal, 0);
    OVS_HIP_TRY(hipGetLastError());
    WinArgs a{};
    a.t_kps = tg.kps;
    a.t_desc = tg.desc;
    a.t_occupied = occupied ? w->d_t_flag : nullptr;
    a.cell_start = tg.cell_start;
    a.items = tg.items;
    a.gp = tg.gp;
    a.n_q = m;
    a.q_xy = w->d_q_xy;
    a.q_x_right = w->d_q_f;
    a.q_valid = w->d_q_flag;
    a.q_radius = w->d_q_r;
    a.q_minl = w->d_q_i;
    a.q_maxl = w->d_q_i2;
    a.q_desc = w->d_q_desc;
    a.margin = margin;
    a.mode = kModeGeneric;
    a.dead_from = dead_from_thr(OVS_HAMMING_DIST_THR_LOW);
    st = build_lists(w, a, m, k_window_lists<false>, k_window_lists<true>, s);
    if (st != OVS_OK) return st;
    ResolveArgs ra{};
    ra.offsets = w->d_offsets;
    ra.keys = w->d_keys;
    ra.n_q = m;
    ra.n_t = n;
    ra.check_orientation = 0;
    ra.q_kps = nullptr;
    ra.t_kps = tg.kps;
    ra.assigned = w->d_assigned;
    ra.num_matches = w->d_num;
    ra.best_only_thr = OVS_HAMMING_DIST_THR_LOW;
    st = launch_resolve<kRuleBestOnly>(ra, s);
    if (st != OVS_OK) return st;
    uint32_t overflow = 0;
    OVS_HIP_TRY(hipMemcpyAsync(assigned, w->d_assigned, sizeof(int32_t) * m, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(num_matches, w->d_num, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(&overflow, w->d_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    if (!overflow) w->overflow_dirty = false;
    return overflow ? OVS_ERR_CAPACITY : OVS_OK;
}

// one direction of match_keyframes_mutually: the landmarks of keyframe A (pose P_a, world positions) against the keypoints of keyframe B
static ovs_status mutual_pass(ovs_wmatcher* w, const ovs_frame_dev* res_b, const ovs_camera* cam_b, const ovs_grid_params* gp_b, const ovs_keypoint* kps_b,
                              const uint8_t* desc_b, int n_b, const double* pose_cw_a, const double* sim_ba, const double* lm_pos_w_a,
                              const float* lm_dist_a, const uint8_t* lm_desc_a, const uint8_t* lm_valid_a, int n_a, const float* scale_factors,
                              int num_levels, float log_scale_factor, float margin, int32_t* d_out, hipStream_t s) {
    FuseArgs a{};
    fill_cam(a.cam, cam_b, gp_b, sim_ba);
    std::memcpy(a.P1, pose_cw_a, sizeof(double) * 12);
    for (int l = 0; l < OVS_MAX_LEVELS; ++l) {
        a.sf[l] = l < num_levels ? scale_factors[l] : 1.0f;
        a.ils[l] = 1.0f;
    }
    a.num_levels = num_levels;
    a.log_scale_factor = log_scale_factor;
    a.margin = margin;
    a.m = n_a;
    a.variant = kFuseMutual;
    a.max_dist = OVS_HAMMING_DIST_THR_HIGH;
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_pos, lm_pos_w_a, sizeof(double) * 3 * n_a, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_xy, lm_dist_a, sizeof(float) * 2 * n_a, hipMemcpyHostToDevice, s));
    OVS_HIP_TRY(hipMemcpyAsync(w->d_q_desc, lm_desc_a, (size_t)32 * n_a, hipMemcpyHostToDevice, s));
    if (lm_valid_a) OVS_HIP_TRY(hipMemcpyAsync(w->d_q_flag, lm_valid_a, (size_t)n_a, hipMemcpyHostToDevice, s));
    TargetRef tg;
    ovs_status st = stage_target(w, res_b, gp_b, kps_b, desc_b, nullptr, n_b, s, &tg);
    if (st != OVS_OK) return st;
    a.t_kps = tg.kps;
    a.t_desc = tg.desc;
    a.cell_start = tg.cell_start;
    a.items = tg.items;
    a.gp = tg.gp;
    a.lm_pos_w = w->d_q_pos;
    a.lm_dist = w->d_q_xy;
    a.lm_desc = w->d_q_desc;
    a.lm_valid = lm_valid_a ? w->d_q_flag : nullptr;
    hipLaunchKernelGGL(k_fuse_best, dim3((unsigned)(((size_t)n_a * kFuseLanes + 255) / 256)), dim3(256), 0, s, a, d_out, w->d_num + 1);   // per-direction count: scratch
    OVS_HIP_TRY(hipGetLastError());
    return OVS_OK;
}

static ovs_status projection_match_keyframes_mutually_impl(ovs_wmatcher* w, const ovs_frame_dev* res_1, const ovs_frame_dev* res_2, const ovs_camera* cam_1, const ovs_grid_params* gp_1,
                                                   const ovs_keypoint* kps_1, const uint8_t* desc_1, int32_t n1, const double* pose_cw_1,
                                                   const double* lm_pos_w_1, const float* lm_dist_1, const uint8_t* lm_desc_1,
                                                   const uint8_t* lm_valid_1, const ovs_camera* cam_2, const ovs_grid_params* gp_2,
                                                   const ovs_keypoint* kps_2, const uint8_t* desc_2, int32_t n2, const double* pose_cw_2,
                                                   const double* lm_pos_w_2, const float* lm_dist_2, const uint8_t* lm_desc_2,
                                                   const uint8_t* lm_valid_2, double s_12, const double* rot_12, const double* trans_12,
                                                   const float* scale_factors, int32_t num_levels, float log_scale_factor, float margin,
                                                   int32_t* matched_2_in_1, int32_t* num_matches) {
    if (!w || !cam_1 || !cam_2 || !gp_1 || !gp_2 || !num_matches || n1 < 0 || n2 < 0 || !pose_cw_1 || !pose_cw_2 || !rot_12 || !trans_12 ||
        !scale_factors || num_levels < 1 || num_levels > OVS_MAX_LEVELS || (cam_1->model != 0 && cam_1->model != 1) ||
        (cam_2->model != 0 && cam_2->model != 1) || !(s_12 > 0.0))
        return OVS_ERR_INVALID;
    *num_matches = 0;
    if (n1 == 0) return OVS_OK;
    if (!matched_2_in_1) return OVS_ERR_INVALID;
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
    if (n2 == 0) return OVS_OK;
    if ((!res_1 && (!kps_1 || !desc_1)) || !lm_pos_w_1 || !lm_dist_1 || !lm_desc_1 || (!res_2 && (!kps_2 || !desc_2)) || !lm_pos_w_2 || !lm_dist_2 || !lm_desc_2)
        return OVS_ERR_INVALID;
    const int nmax = std::max(n1, n2);
    if (nmax > w->max_t || nmax > w->max_q) return OVS_ERR_CAPACITY;
    if ((size_t)(n1 + n2) > (size_t)w->max_entries) return OVS_ERR_CAPACITY;   // the two one-way results ride in the key buffer
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    // Sim3 in both directions: [s_12 R_12 | t_12] takes keyframe-2 coordinates to keyframe 1; [R_12^T / s_12 | -(R_12^T / s_12) t_12] back
    double S12[12], S21[12];
    for (int i = 0; i < 9; ++i) S12[i] = s_12 * rot_12[i];
    for (int i = 0; i < 3; ++i) S12[9 + i] = trans_12[i];
    const double inv_s = 1.0 / s_12;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S21[3 * r + c] = inv_s * rot_12[3 * c + r];
    for (int r = 0; r < 3; ++r) S21[9 + r] = -((S21[3 * r] * trans_12[0] + S21[3 * r + 1] * trans_12[1]) + S21[3 * r + 2] * trans_12[2]);
    int32_t* d_2_in_1 = reinterpret_cast<int32_t*>(w->d_keys);
    int32_t* d_1_in_2 = d_2_in_1 + n1;
    ovs_status st = mutual_pass(w, res_2, cam_2, gp_2, kps_2, desc_2, n2, pose_cw_1, S21, lm_pos_w_1, lm_dist_1, lm_desc_1, lm_valid_1, n1, scale_factors,
                                num_levels, log_scale_factor, margin, d_2_in_1, s);
    if (st != OVS_OK) return st;
    st = mutual_pass(w, res_1, cam_1, gp_1, kps_1, desc_1, n1, pose_cw_2, S12, lm_pos_w_2, lm_dist_2, lm_desc_2, lm_valid_2, n2, scale_factors, num_levels,
                     log_scale_factor, margin, d_1_in_2, s);
    if (st != OVS_OK) return st;
    OVS_HIP_TRY(hipMemsetAsync(w->d_num, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_cross_check, dim3((n1 + 255) / 256), dim3(256), 0, s, (const int32_t*)d_2_in_1, (const int32_t*)d_1_in_2, n1, n2,
                       w->d_assigned, w->d_num);
    OVS_HIP_TRY(hipGetLastError());
    OVS_HIP_TRY(hipMemcpyAsync(matched_2_in_1, w->d_assigned, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(num_matches, w->d_num, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    return OVS_OK;
}

}   // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// Frame residency behind the class boundary (round 3): a frame's matcher-side data -- undistorted keypoints,
// descriptors, stereo_x_right and the keypoint grid data::assign_keypoints_to_grid builds in the frame's constructor -- is uploaded and
// indexed ONCE per frame (ovs_frame_dev), and the matchers that tracking_module calls two to four times on the same frame take the handle.
// What remains per call is what really changes per call: the landmark side, staged through ONE pinned buffer into ONE device arena (one
// copy up), and the result block (one copy down).
// ---------------------------------------------------------------------------------------------------------------------------

namespace {


// device arenas of destroyed frame handles, kept for the next frame (a tracker creates one handle per frame: a hipMalloc / hipFree pair
// per frame would cost more than the upload). Bounded; keyed by device and size.
struct FrameArenaPool {
    struct Item {
        int device;
        size_t bytes;
        unsigned char* p;
    };
    std::mutex mu;
    std::vector<Item> free_list;
    unsigned char* take(int device, size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].device == device && free_list[i].bytes == bytes) {
                unsigned char* p = free_list[i].p;
                free_list.erase(free_list.begin() + (long)i);
                return p;
            }
        return nullptr;
    }
    void give(int device, size_t bytes, unsigned char* p) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (free_list.size() < 8) {
                free_list.push_back(Item{device, bytes, p});
                return;
            }
        }
        (void)hipSetDevice(device);
        (void)hipFree(p);
    }
};
FrameArenaPool g_frame_pool;

// pinned staging for the once-per-frame upload, per thread (frames are created by the tracking thread; stereo rigs by two)
struct FrameStage {
    unsigned char* h = nullptr;
    size_t cap = 0;
    hipStream_t stream = nullptr;
    int device = -1;
    ~FrameStage() {
        if (h) (void)hipHostFree(h);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

}   // namespace

extern "C" {

ovs_status ovs_frame_dev_destroy(ovs_frame_dev* f) {
    if (!f) return OVS_OK;
    if (f->arena) g_frame_pool.give(f->device, f->arena_bytes, f->arena);
    if (f->d_bearings) (void)hipFree(f->d_bearings);
    delete f;
    return OVS_OK;
}

ovs_status ovs_frame_dev_create(int32_t device, const ovs_grid_params* gp, const ovs_keypoint* undist_kps, const uint8_t* desc,
                                const float* stereo_x_right, int32_t n, ovs_frame_dev** out) {
    if (!out || !gp || n < 0 || n > 65534 || (n > 0 && (!undist_kps || !desc))) return OVS_ERR_INVALID;
    *out = nullptr;
    if (gp->cols < 1 || gp->rows < 1 || gp->cols * gp->rows > kMaxGridCells || !(gp->max_x > gp->min_x) || !(gp->max_y > gp->min_y)) return OVS_ERR_INVALID;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    OVS_HIP_TRY(hipSetDevice(device));
    ovs_frame_dev* f = new (std::nothrow) ovs_frame_dev();
    if (!f) return OVS_ERR_INVALID;
    f->device = device;
    f->n = n;
    f->cap = std::max((n + 4095) & ~4095, 4096);   // capacity classes of 4096 keypoints: pooled arenas fit the next frame
    f->n_cells = gp->cols * gp->rows;
    f->has_stereo = stereo_x_right != nullptr;
    f->gpp = *gp;
    f->gp = make_gridp(*gp);
    const size_t C = (size_t)f->cap;
    const size_t o_kps = 0, o_desc = al256(o_kps + sizeof(ovs_keypoint) * C), o_xr = al256(o_desc + 32 * C), o_cellof = al256(o_xr + 4 * C),
                 o_items = al256(o_cellof + 4 * C), o_start = al256(o_items + 4 * C);
    f->arena_bytes = al256(o_start + sizeof(int32_t) * (kMaxGridCells + 1));
    f->arena = g_frame_pool.take(device, f->arena_bytes);
    if (!f->arena) {
        const hipError_t e = hipMalloc(&f->arena, f->arena_bytes);
        if (e != hipSuccess) {
            ovs::set_last_error("hipMalloc(frame arena)", e);
            delete f;
            return OVS_ERR_HIP;
        }
    }
    f->d_kps = reinterpret_cast<ovs_keypoint*>(f->arena + o_kps);
    f->d_desc = f->arena + o_desc;
    f->d_x_right = reinterpret_cast<float*>(f->arena + o_xr);
    f->d_cell_of = reinterpret_cast<int32_t*>(f->arena + o_cellof);
    f->d_items = reinterpret_cast<int32_t*>(f->arena + o_items);
    f->d_cell_start = reinterpret_cast<int32_t*>(f->arena + o_start);
    static thread_local FrameStage st;
    ovs_status rc = OVS_ERR_HIP;
    hipError_t er = hipSuccess;
    do {
#define F_TRY(expr)                           \
    if ((er = (expr)) != hipSuccess) {        \
        ovs::set_last_error(#expr, er);       \
        break;                                \
    }
        if (st.device != device) {
            if (st.stream) (void)hipStreamDestroy(st.stream);
            st.stream = nullptr;
            st.device = device;
        }
        if (!st.stream) F_TRY(hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking));
        // the three arrays keep their arena offsets in the staging buffer, so ONE copy of [0, end of the last array) uploads them
        const size_t up_bytes = n ? (stereo_x_right ? o_xr + 4 * (size_t)n : o_desc + 32 * (size_t)n) : 0;
        if (st.cap < up_bytes) {
            if (st.h) (void)hipHostFree(st.h);
            st.h = nullptr;
            st.cap = 0;
            const size_t want = std::max(up_bytes, al256(o_xr + 4 * (size_t)4096 * 4));
            F_TRY(hipHostMalloc(reinterpret_cast<void**>(&st.h), want, hipHostMallocDefault));
            st.cap = want;
        }
        if (n) {
            std::memcpy(st.h + o_kps, undist_kps, sizeof(ovs_keypoint) * (size_t)n);
            std::memcpy(st.h + o_desc, desc, 32 * (size_t)n);
            if (stereo_x_right) std::memcpy(st.h + o_xr, stereo_x_right, 4 * (size_t)n);
            F_TRY(hipMemcpyAsync(f->arena, st.h, up_bytes, hipMemcpyHostToDevice, st.stream));
        }
        const int in_lds = (n <= kGridItemsLds && (size_t)(2 * f->n_cells + 1 + n) * sizeof(int32_t) <= 144 * 1024) ? 1 : 0;
        const size_t lds = (size_t)(2 * f->n_cells + 1 + (in_lds ? n : 0)) * sizeof(int32_t);
        if (lds > 64 * 1024)
            F_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_assign), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_grid_assign, dim3(1), dim3(1024), lds, st.stream, (const ovs_keypoint*)f->d_kps, n, f->gp, f->d_cell_of, f->d_cell_start,
                           f->d_items, in_lds);
        F_TRY(hipGetLastError());
        F_TRY(hipStreamSynchronize(st.stream));   // the handle is used from other streams afterwards
        rc = OVS_OK;
#undef F_TRY
    } while (0);
    if (rc != OVS_OK) {
        ovs_frame_dev_destroy(f);
        return rc;
    }
    *out = f;
    return OVS_OK;
}

int32_t ovs_frame_dev_num_keypoints(const ovs_frame_dev* f) { return f ? f->n : -1; }

ovs_status ovs_projection_match_frame_and_landmarks_f(ovs_wmatcher* w, const ovs_frame_dev* frm, const uint8_t* occupied, const float* lm_xy,
                                                      const float* lm_x_right, const int32_t* lm_level, const uint8_t* lm_desc,
                                                      const uint8_t* lm_valid, int32_t m, const float* scale_factors, int32_t num_levels,
                                                      float margin, float lowe_ratio, int32_t* assigned, int32_t* num_matches) {
    if (!w || !frm || !num_matches || m < 0 || !scale_factors || num_levels < 1 || num_levels > OVS_MAX_LEVELS) return OVS_ERR_INVALID;
    *num_matches = 0;
    if (m == 0) return OVS_OK;
    if (!assigned) return OVS_ERR_INVALID;
    const int n = frm->n;
    if (n == 0) {
        for (int i = 0; i < m; ++i) assigned[i] = -1;
        return OVS_OK;
    }
    if (!lm_xy || !lm_level || !lm_desc || (frm->has_stereo && !lm_x_right)) return OVS_ERR_INVALID;
    if (frm->device != w->device) return OVS_ERR_INVALID;
    if (n > w->max_t || m > w->max_q) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    Stager stg(w);
    WinArgs a{};
    a.t_kps = frm->d_kps;
    a.t_desc = frm->d_desc;
    a.t_occupied = stg.put(occupied, (size_t)n);
    a.t_x_right = frm->has_stereo ? frm->d_x_right : nullptr;
    a.cell_start = frm->d_cell_start;
    a.items = frm->d_items;
    a.gp = frm->gp;
    a.n_q = m;
    a.q_xy = stg.put(lm_xy, (size_t)2 * m);
    a.q_x_right = frm->has_stereo ? stg.put(lm_x_right, (size_t)m) : nullptr;
    a.q_level = stg.put(lm_level, (size_t)m);
    a.q_valid = stg.put(lm_valid, (size_t)m);
    a.q_desc = stg.put(lm_desc, (size_t)32 * m);
    a.margin = margin;
    for (int l = 0; l < OVS_MAX_LEVELS; ++l) a.sf[l] = l < num_levels ? scale_factors[l] : 1.0f;
    a.mode = kModeProjection;
    a.dead_from = dead_from_ratio(OVS_HAMMING_DIST_THR_HIGH, lowe_ratio);
    if (stg.overflow) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(stg.flush(s));
    ovs_status st = build_lists(w, a, m, k_window_lists<false>, k_window_lists<true>, s);
    if (st != OVS_OK) return st;
    ResolveArgs ra{};
    ra.offsets = w->d_offsets;
    ra.keys = w->d_keys;
    ra.n_q = m;
    ra.n_t = n;
    ra.lowe_ratio = lowe_ratio;
    ra.assigned = w->d_assigned;
    ra.num_matches = w->d_num;
    st = launch_resolve<kRuleProjection>(ra, s);
    if (st != OVS_OK) return st;
    return fetch_results(w, m, assigned, num_matches, s);
}

ovs_status ovs_area_match_in_consistent_area_f(ovs_wmatcher* w, const ovs_frame_dev* frm_1, const ovs_frame_dev* frm_2, float* prev_matched_xy,
                                               int32_t* matched_2_in_1, int32_t margin, float lowe_ratio, int32_t check_orientation,
                                               int32_t* num_matches) {
    if (!w || !frm_1 || !frm_2 || !num_matches) return OVS_ERR_INVALID;
    *num_matches = 0;
    const int n1 = frm_1->n, n2 = frm_2->n;
    if (n1 == 0) return OVS_OK;
    if (!prev_matched_xy || !matched_2_in_1) return OVS_ERR_INVALID;
    if (n2 == 0) {
        for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
        return OVS_OK;
    }
    if (frm_1->device != w->device || frm_2->device != w->device) return OVS_ERR_INVALID;
    if (n2 > w->max_t || n1 > w->max_q) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    Stager stg(w);
    // prev_matched_pts are read AND updated by the resolver: they live in the staging arena and come back with their own copy
    const float* d_prev_c = stg.put(prev_matched_xy, (size_t)2 * n1);
    if (stg.overflow) return OVS_ERR_CAPACITY;
    float* d_prev = const_cast<float*>(d_prev_c);
    OVS_HIP_TRY(stg.flush(s));
    WinArgs a{};
    a.t_kps = frm_2->d_kps;
    a.t_desc = frm_2->d_desc;
    a.cell_start = frm_2->d_cell_start;
    a.items = frm_2->d_items;
    a.gp = frm_2->gp;
    a.n_q = n1;
    a.q_xy = d_prev;
    a.q_kps = frm_1->d_kps;
    a.q_desc = frm_1->d_desc;
    a.margin = (float)margin;
    a.mode = kModeArea;
    a.dead_from = dead_from_ratio(OVS_HAMMING_DIST_THR_LOW, lowe_ratio);
    ovs_status st = build_lists(w, a, n1, k_window_lists<false>, k_window_lists<true>, s);
    if (st != OVS_OK) return st;
    ResolveArgs ra{};
    ra.offsets = w->d_offsets;
    ra.keys = w->d_keys;
    ra.n_q = n1;
    ra.n_t = n2;
    ra.lowe_ratio = lowe_ratio;
    ra.check_orientation = check_orientation;
    ra.q_kps = frm_1->d_kps;
    ra.t_kps = frm_2->d_kps;
    ra.prev_matched_xy = d_prev;
    ra.assigned = w->d_assigned;
    ra.num_matches = w->d_num;
    st = launch_resolve<kRuleArea>(ra, s);
    if (st != OVS_OK) return st;
    OVS_HIP_TRY(hipMemcpyAsync(w->h_stage, d_prev, sizeof(float) * 2 * (size_t)n1, hipMemcpyDeviceToHost, s));
    st = fetch_results(w, n1, matched_2_in_1, num_matches, s);
    if (st != OVS_ERR_HIP) std::memcpy(prev_matched_xy, w->h_stage, sizeof(float) * 2 * (size_t)n1);   // only behind fetch_results' synchronisation
    return st;
}

ovs_status ovs_projection_match_current_and_last_frames_f(ovs_wmatcher* w, const ovs_camera* cam, const ovs_frame_dev* curr,
                                                          const uint8_t* curr_occupied, const double* pose_cw_curr, const ovs_keypoint* last_kps,
                                                          const double* last_pos_w, const uint8_t* last_lm_desc, const uint8_t* last_valid,
                                                          int32_t n_last, const double* pose_cw_last, const float* scale_factors,
                                                          int32_t num_levels, float margin, int32_t check_orientation, int32_t* assigned,
                                                          int32_t* num_matches) {
    if (!w || !cam || !curr || !num_matches || n_last < 0 || !pose_cw_curr || !pose_cw_last || !scale_factors || num_levels < 1 ||
        num_levels > OVS_MAX_LEVELS || (cam->model != 0 && cam->model != 1))
        return OVS_ERR_INVALID;
    *num_matches = 0;
    if (n_last == 0) return OVS_OK;
    if (!assigned) return OVS_ERR_INVALID;
    for (int i = 0; i < n_last; ++i) assigned[i] = -1;
    const int n_curr = curr->n;
    if (n_curr == 0) return OVS_OK;
    if (!last_kps || !last_pos_w || !last_lm_desc) return OVS_ERR_INVALID;
    if (curr->device != w->device) return OVS_ERR_INVALID;
    if (n_curr > w->max_t || n_last > w->max_q) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(w->device));
    hipStream_t s = w->stream;
    // motion direction (host, double): trans_wc = -rot_cw^T trans_cw; trans_lc = rot_lw trans_wc + trans_lw
    const double* Rc = pose_cw_curr;
    const double* tc = pose_cw_curr + 9;
    const double twc[3] = {-((Rc[0] * tc[0] + Rc[3] * tc[1]) + Rc[6] * tc[2]), -((Rc[1] * tc[0] + Rc[4] * tc[1]) + Rc[7] * tc[2]),
                           -((Rc[2] * tc[0] + Rc[5] * tc[1]) + Rc[8] * tc[2])};
    const double* Rl = pose_cw_last;
    const double tlc_z = ((Rl[6] * twc[0] + Rl[7] * twc[1]) + Rl[8] * twc[2]) + pose_cw_last[11];
    const int forward = cam->setup == 0 ? 0 : (tlc_z > cam->true_baseline);
    const int backward = cam->setup == 0 ? 0 : (-tlc_z > cam->true_baseline);
    CamP cp{};
    cp.model = cam->model;
    cp.setup = cam->setup;
    cp.fx = cam->fx;
    cp.fy = cam->fy;
    cp.cx = cam->cx;
    cp.cy = cam->cy;
    cp.fxb = cam->focal_x_baseline;
    cp.cols = cam->cols;
    cp.rows = cam->rows;
    cp.min_x = curr->gpp.min_x;
    cp.min_y = curr->gpp.min_y;
    cp.max_x = curr->gpp.max_x;
    cp.max_y = curr->gpp.max_y;
    std::memcpy(cp.P, pose_cw_curr, sizeof(double) * 12);
    float sf16[OVS_MAX_LEVELS];
    for (int l = 0; l < OVS_MAX_LEVELS; ++l) sf16[l] = l < num_levels ? scale_factors[l] : 1.0f;
    Stager stg(w);
    const uint8_t* d_occ = stg.put(curr_occupied, (size_t)n_curr);
    const ovs_keypoint* d_q_kps = stg.put(last_kps, (size_t)n_last);
    const double* d_q_pos = stg.put(last_pos_w, (size_t)3 * n_last);
    const uint8_t* d_q_desc = stg.put(last_lm_desc, (size_t)32 * n_last);
    const float* d_sf = stg.put(sf16, (size_t)OVS_MAX_LEVELS);
    // the query flags are written by k_reproject_queries (in place over last_valid when given: same index, read-then-write by one lane)
    uint8_t* d_q_flag = last_valid ? const_cast<uint8_t*>(stg.put(last_valid, (size_t)n_last)) : stg.reserve<uint8_t>((size_t)n_last);
    float* d_q_xy = stg.reserve<float>((size_t)2 * n_last);
    float* d_q_f = stg.reserve<float>((size_t)n_last);
    float* d_q_r = stg.reserve<float>((size_t)n_last);
    int32_t* d_q_i = stg.reserve<int32_t>((size_t)n_last);
    int32_t* d_q_i2 = stg.reserve<int32_t>((size_t)n_last);
    if (stg.overflow) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(stg.flush(s));
    hipLaunchKernelGGL(k_reproject_queries, dim3((n_last + 255) / 256), dim3(256), 0, s, cp, d_q_kps, d_q_pos, (const uint8_t*)(last_valid ? d_q_flag : nullptr),
                       n_last, margin, d_sf, num_levels, forward, backward, (const float*)nullptr, 0.0, 0.0, 0.0, 0.0f, d_q_xy, d_q_f, d_q_r, d_q_i, d_q_i2,
                       d_q_flag);
    OVS_HIP_TRY(hipGetLastError());
    WinArgs a{};
    a.t_kps = curr->d_kps;
    a.t_desc = curr->d_desc;
    a.t_occupied = d_occ;
    a.t_x_right = curr->has_stereo ? curr->d_x_right : nullptr;
    a.cell_start = curr->d_cell_start;
    a.items = curr->d_items;
    a.gp = curr->gp;
    a.n_q = n_last;
    a.q_xy = d_q_xy;
    a.q_x_right = d_q_f;
    a.q_valid = d_q_flag;
    a.q_radius = d_q_r;
    a.q_minl = d_q_i;
    a.q_maxl = d_q_i2;
    a.q_desc = d_q_desc;
    a.margin = margin;
    a.mode = kModeGeneric;
    a.dead_from = dead_from_thr(OVS_HAMMING_DIST_THR_HIGH);
    ovs_status st = build_lists(w, a, n_last, k_window_lists<false>, k_window_lists<true>, s);
    if (st != OVS_OK) return st;
    ResolveArgs ra{};
    ra.offsets = w->d_offsets;
    ra.keys = w->d_keys;
    ra.n_q = n_last;
    ra.n_t = n_curr;
    ra.check_orientation = check_orientation;
    ra.q_kps = d_q_kps;
    ra.t_kps = curr->d_kps;
    ra.assigned = w->d_assigned;
    ra.num_matches = w->d_num;
    ra.best_only_thr = OVS_HAMMING_DIST_THR_HIGH;
    st = launch_resolve<kRuleBestOnly>(ra, s);
    if (st != OVS_OK) return st;
    return fetch_results(w, n_last, assigned, num_matches, s);
}

}   // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// Keyframe-side matchers: the host-array forms and their resident twins (_f, round 4). A twin takes an ovs_frame_dev handle wherever the
// host form takes (grid parameters, keypoints, descriptors, stereo_x_right, n) of the TARGET frame / keyframe; everything else -- the
// landmark side, poses, thresholds, outputs -- is unchanged, and so are the results (tests/test_gpu_resident.py: bit-equal to the host forms).
// ---------------------------------------------------------------------------------------------------------------------------
extern "C" {

ovs_status ovs_fuse_replace_duplication(ovs_wmatcher* w, const ovs_camera* cam, const ovs_grid_params* gp, const ovs_keypoint* kps, const uint8_t* desc,
                                        const float* stereo_x_right, int32_t n, const double* pose_cw, const double* lm_pos_w, const float* lm_dist_min_max,
                                        const double* lm_normal, const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m, const float* scale_factors,
                                        const float* inv_level_sigma_sq, int32_t num_levels, float log_scale_factor, float margin, int32_t* best_idx,
                                        int32_t* num_fused) {
    return fuse_replace_duplication_impl(w, nullptr, cam, gp, kps, desc, stereo_x_right, n, pose_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc, lm_valid, m,
                                         scale_factors, inv_level_sigma_sq, num_levels, log_scale_factor, margin, best_idx, num_fused);
}
ovs_status ovs_fuse_replace_duplication_f(ovs_wmatcher* w, const ovs_camera* cam, const ovs_frame_dev* keyfrm, const double* pose_cw, const double* lm_pos_w,
                                          const float* lm_dist_min_max, const double* lm_normal, const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m,
                                          const float* scale_factors, const float* inv_level_sigma_sq, int32_t num_levels, float log_scale_factor,
                                          float margin, int32_t* best_idx, int32_t* num_fused) {
    if (!keyfrm) return OVS_ERR_INVALID;
    return fuse_replace_duplication_impl(w, keyfrm, cam, &keyfrm->gpp, nullptr, nullptr, nullptr, keyfrm->n, pose_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc,
                                         lm_valid, m, scale_factors, inv_level_sigma_sq, num_levels, log_scale_factor, margin, best_idx, num_fused);
}

ovs_status ovs_projection_match_frame_and_keyframe(ovs_wmatcher* w, const ovs_camera* cam, const ovs_grid_params* gp, const ovs_keypoint* curr_kps,
                                                   const uint8_t* curr_desc, const uint8_t* curr_occupied, int32_t n_curr, const double* pose_cw_curr,
                                                   const ovs_keypoint* kf_kps, const double* kf_pos_w, const float* kf_dist_min_max, const uint8_t* kf_lm_desc,
                                                   const uint8_t* kf_valid, int32_t n_kf, const float* scale_factors, int32_t num_levels,
                                                   float log_scale_factor, float margin, uint32_t hamm_dist_thr, int32_t check_orientation, int32_t* assigned,
                                                   int32_t* num_matches) {
    return projection_match_frame_and_keyframe_impl(w, nullptr, cam, gp, curr_kps, curr_desc, curr_occupied, n_curr, pose_cw_curr, kf_kps, kf_pos_w, kf_dist_min_max,
                                                    kf_lm_desc, kf_valid, n_kf, scale_factors, num_levels, log_scale_factor, margin, hamm_dist_thr,
                                                    check_orientation, assigned, num_matches);
}
ovs_status ovs_projection_match_frame_and_keyframe_f(ovs_wmatcher* w, const ovs_camera* cam, const ovs_frame_dev* curr, const uint8_t* curr_occupied,
                                                     const double* pose_cw_curr, const ovs_keypoint* kf_kps, const double* kf_pos_w,
                                                     const float* kf_dist_min_max, const uint8_t* kf_lm_desc, const uint8_t* kf_valid, int32_t n_kf,
                                                     const float* scale_factors, int32_t num_levels, float log_scale_factor, float margin,
                                                     uint32_t hamm_dist_thr, int32_t check_orientation, int32_t* assigned, int32_t* num_matches) {
    if (!curr) return OVS_ERR_INVALID;
    return projection_match_frame_and_keyframe_impl(w, curr, cam, &curr->gpp, nullptr, nullptr, curr_occupied, curr->n, pose_cw_curr, kf_kps, kf_pos_w, kf_dist_min_max,
                                                    kf_lm_desc, kf_valid, n_kf, scale_factors, num_levels, log_scale_factor, margin, hamm_dist_thr,
                                                    check_orientation, assigned, num_matches);
}

ovs_status ovs_fuse_detect_duplication(ovs_wmatcher* w, const ovs_camera* cam, const ovs_grid_params* gp, const ovs_keypoint* kps, const uint8_t* desc, int32_t n,
                                       const double* sim3_cw, const double* lm_pos_w, const float* lm_dist_min_max, const double* lm_normal,
                                       const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m, const float* scale_factors, int32_t num_levels,
                                       float log_scale_factor, float margin, int32_t* best_idx, int32_t* num_found) {
    return fuse_detect_duplication_impl(w, nullptr, cam, gp, kps, desc, n, sim3_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc, lm_valid, m, scale_factors, num_levels,
                                        log_scale_factor, margin, best_idx, num_found);
}
ovs_status ovs_fuse_detect_duplication_f(ovs_wmatcher* w, const ovs_camera* cam, const ovs_frame_dev* keyfrm, const double* sim3_cw, const double* lm_pos_w,
                                         const float* lm_dist_min_max, const double* lm_normal, const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m,
                                         const float* scale_factors, int32_t num_levels, float log_scale_factor, float margin, int32_t* best_idx,
                                         int32_t* num_found) {
    if (!keyfrm) return OVS_ERR_INVALID;
    return fuse_detect_duplication_impl(w, keyfrm, cam, &keyfrm->gpp, nullptr, nullptr, keyfrm->n, sim3_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc, lm_valid, m,
                                        scale_factors, num_levels, log_scale_factor, margin, best_idx, num_found);
}

ovs_status ovs_projection_match_by_sim3_transform(ovs_wmatcher* w, const ovs_camera* cam, const ovs_grid_params* gp, const ovs_keypoint* kps, const uint8_t* desc,
                                                  const uint8_t* occupied, int32_t n, const double* sim3_cw, const double* lm_pos_w,
                                                  const float* lm_dist_min_max, const double* lm_normal, const uint8_t* lm_desc, const uint8_t* lm_valid,
                                                  int32_t m, const float* scale_factors, int32_t num_levels, float log_scale_factor, float margin,
                                                  int32_t* assigned, int32_t* num_matches) {
    return projection_match_by_sim3_transform_impl(w, nullptr, cam, gp, kps, desc, occupied, n, sim3_cw, lm_pos_w, lm_dist_min_max, lm_normal, lm_desc, lm_valid, m,
                                                   scale_factors, num_levels, log_scale_factor, margin, assigned, num_matches);
}
ovs_status ovs_projection_match_by_sim3_transform_f(ovs_wmatcher* w, const ovs_camera* cam, const ovs_frame_dev* keyfrm, const uint8_t* occupied,
                                                    const double* sim3_cw, const double* lm_pos_w, const float* lm_dist_min_max, const double* lm_normal,
                                                    const uint8_t* lm_desc, const uint8_t* lm_valid, int32_t m, const float* scale_factors,
                                                    int32_t num_levels, float log_scale_factor, float margin, int32_t* assigned, int32_t* num_matches) {
    if (!keyfrm) return OVS_ERR_INVALID;
    return projection_match_by_sim3_transform_impl(w, keyfrm, cam, &keyfrm->gpp, nullptr, nullptr, occupied, keyfrm->n, sim3_cw, lm_pos_w, lm_dist_min_max, lm_normal,
                                                   lm_desc, lm_valid, m, scale_factors, num_levels, log_scale_factor, margin, assigned, num_matches);
}

ovs_status ovs_projection_match_keyframes_mutually(ovs_wmatcher* w, const ovs_camera* cam_1, const ovs_grid_params* gp_1, const ovs_keypoint* kps_1,
                                                   const uint8_t* desc_1, int32_t n1, const double* pose_cw_1, const double* lm_pos_w_1,
                                                   const float* lm_dist_1, const uint8_t* lm_desc_1, const uint8_t* lm_valid_1, const ovs_camera* cam_2,
                                                   const ovs_grid_params* gp_2, const ovs_keypoint* kps_2, const uint8_t* desc_2, int32_t n2,
                                                   const double* pose_cw_2, const double* lm_pos_w_2, const float* lm_dist_2, const uint8_t* lm_desc_2,
                                                   const uint8_t* lm_valid_2, double s_12, const double* rot_12, const double* trans_12,
                                                   const float* scale_factors, int32_t num_levels, float log_scale_factor, float margin,
                                                   int32_t* matched_2_in_1, int32_t* num_matches) {
    return projection_match_keyframes_mutually_impl(w, nullptr, nullptr, cam_1, gp_1, kps_1, desc_1, n1, pose_cw_1, lm_pos_w_1, lm_dist_1, lm_desc_1, lm_valid_1, cam_2,
                                                    gp_2, kps_2, desc_2, n2, pose_cw_2, lm_pos_w_2, lm_dist_2, lm_desc_2, lm_valid_2, s_12, rot_12, trans_12,
                                                    scale_factors, num_levels, log_scale_factor, margin, matched_2_in_1, num_matches);
}
ovs_status ovs_projection_match_keyframes_mutually_f(ovs_wmatcher* w, const ovs_camera* cam_1, const ovs_frame_dev* keyfrm_1, const double* pose_cw_1,
                                                     const double* lm_pos_w_1, const float* lm_dist_1, const uint8_t* lm_desc_1, const uint8_t* lm_valid_1,
                                                     const ovs_camera* cam_2, const ovs_frame_dev* keyfrm_2, const double* pose_cw_2,
                                                     const double* lm_pos_w_2, const float* lm_dist_2, const uint8_t* lm_desc_2, const uint8_t* lm_valid_2,
                                                     double s_12, const double* rot_12, const double* trans_12, const float* scale_factors,
                                                     int32_t num_levels, float log_scale_factor, float margin, int32_t* matched_2_in_1,
                                                     int32_t* num_matches) {
    if (!keyfrm_1 || !keyfrm_2) return OVS_ERR_INVALID;
    return projection_match_keyframes_mutually_impl(w, keyfrm_1, keyfrm_2, cam_1, &keyfrm_1->gpp, nullptr, nullptr, keyfrm_1->n, pose_cw_1, lm_pos_w_1, lm_dist_1,
                                                    lm_desc_1, lm_valid_1, cam_2, &keyfrm_2->gpp, nullptr, nullptr, keyfrm_2->n, pose_cw_2, lm_pos_w_2, lm_dist_2,
                                                    lm_desc_2, lm_valid_2, s_12, rot_12, trans_12, scale_factors, num_levels, log_scale_factor, margin,
                                                    matched_2_in_1, num_matches);
}

// bow_tree / robust::match_for_triangulation with both sides resident: only the per-call flags and the BoW feature vectors (node CSRs) travel
ovs_status ovs_bow_match_frame_and_keyframe_f(ovs_wmatcher* w, const ovs_frame_dev* keyfrm, const uint8_t* kf_valid, const int32_t* kf_node_ids,
                                              const int32_t* kf_node_start, const int32_t* kf_items, int32_t kf_nodes, const ovs_frame_dev* frm,
                                              const int32_t* frm_node_ids, const int32_t* frm_node_start, const int32_t* frm_items, int32_t frm_nodes,
                                              float lowe_ratio, int32_t check_orientation, int32_t* matched_kf_in_frm, int32_t* num_matches) {
    if (!keyfrm || !frm) return OVS_ERR_INVALID;
    return bow_match_impl(w, keyfrm, frm, 0, nullptr, nullptr, nullptr, nullptr, kf_valid, keyfrm->n, kf_node_ids, kf_node_start, kf_items, kf_nodes, nullptr, nullptr,
                          frm->n, frm_node_ids, frm_node_start, frm_items, frm_nodes, lowe_ratio, check_orientation, matched_kf_in_frm, num_matches);
}
ovs_status ovs_bow_match_keyframes_f(ovs_wmatcher* w, const ovs_frame_dev* keyfrm_1, const uint8_t* valid_1, const int32_t* node_ids_1,
                                     const int32_t* node_start_1, const int32_t* items_1, int32_t nodes_1, const ovs_frame_dev* keyfrm_2,
                                     const uint8_t* valid_2, const int32_t* node_ids_2, const int32_t* node_start_2, const int32_t* items_2, int32_t nodes_2,
                                     float lowe_ratio, int32_t check_orientation, int32_t* matched_2_in_1, int32_t* num_matches) {
    if (!keyfrm_1 || !keyfrm_2) return OVS_ERR_INVALID;
    return bow_match_impl(w, keyfrm_1, keyfrm_2, 1, valid_2, nullptr, nullptr, nullptr, valid_1, keyfrm_1->n, node_ids_1, node_start_1, items_1, nodes_1, nullptr, nullptr,
                          keyfrm_2->n, node_ids_2, node_start_2, items_2, nodes_2, lowe_ratio, check_orientation, matched_2_in_1, num_matches);
}
ovs_status ovs_robust_match_for_triangulation_f(ovs_wmatcher* w, const ovs_frame_dev* keyfrm_1, const uint8_t* has_lm_1, const int32_t* node_ids_1,
                                                const int32_t* node_start_1, const int32_t* items_1, int32_t nodes_1, const ovs_frame_dev* keyfrm_2,
                                                const uint8_t* has_lm_2, const int32_t* node_ids_2, const int32_t* node_start_2, const int32_t* items_2,
                                                int32_t nodes_2, const double* E_12, const double* epipole_in_2, const float* scale_factors,
                                                int32_t num_levels, int32_t check_orientation, int32_t* matched_2_in_1, int32_t* num_matches) {
    if (!keyfrm_1 || !keyfrm_2 || !keyfrm_1->d_bearings || !keyfrm_2->d_bearings || !E_12 || !epipole_in_2 || !scale_factors || num_levels < 1 ||
        num_levels > OVS_MAX_LEVELS)
        return OVS_ERR_INVALID;
    const int n1 = keyfrm_1->n, n2 = keyfrm_2->n;
    std::vector<uint8_t> v1((size_t)std::max(n1, 1), 1), v2((size_t)std::max(n2, 1), 1);   // "valid" for the bow kernels = NO landmark yet
    if (has_lm_1)
        for (int i = 0; i < n1; ++i) v1[i] = has_lm_1[i] ? 0 : 1;
    if (has_lm_2)
        for (int i = 0; i < n2; ++i) v2[i] = has_lm_2[i] ? 0 : 1;
    TriParams tp{nullptr, nullptr, nullptr, nullptr, E_12, epipole_in_2, scale_factors, num_levels};
    return bow_match_impl(w, keyfrm_1, keyfrm_2, 1, v2.data(), &tp, nullptr, nullptr, v1.data(), n1, node_ids_1, node_start_1, items_1, nodes_1, nullptr, nullptr, n2,
                          node_ids_2, node_start_2, items_2, nodes_2, 0.0f, check_orientation, matched_2_in_1, num_matches);
}

// 3 doubles per keypoint (data::keyframe::bearings_), uploaded once. Call it before the handle is shared between threads.
ovs_status ovs_frame_dev_attach_bearings(ovs_frame_dev* f, const double* bearings) {
    if (!f || (f->n > 0 && !bearings)) return OVS_ERR_INVALID;
    if (f->n == 0) return OVS_OK;
    OVS_HIP_TRY(hipSetDevice(f->device));
    if (!f->d_bearings) OVS_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&f->d_bearings), sizeof(double) * 3 * (size_t)f->cap));
    OVS_HIP_TRY(hipMemcpy(f->d_bearings, bearings, sizeof(double) * 3 * (size_t)f->n, hipMemcpyHostToDevice));
    return OVS_OK;
}
int32_t ovs_frame_dev_device(const ovs_frame_dev* f) { return f ? f->device : -1; }

ovs_status ovs_match_set_variant(int32_t which, int32_t value) {
    if (value != 0 && value != 1) return OVS_ERR_INVALID;
    switch (which) {
        case OVS_MATCH_VARIANT_ANGLE_KEEP_RULE: g_angle_keep_rule.store(value, std::memory_order_relaxed); return OVS_OK;
        case OVS_MATCH_VARIANT_ANGLE_TIE_ORDER: g_angle_tie_order.store(value, std::memory_order_relaxed); return OVS_OK;
        case OVS_MATCH_VARIANT_BF_FRAME_MASK: g_bf_frame_mask.store(value, std::memory_order_relaxed); return OVS_OK;
        default: return OVS_ERR_INVALID;
    }
}
int32_t ovs_match_get_variant(int32_t which) {
    switch (which) {
        case OVS_MATCH_VARIANT_ANGLE_KEEP_RULE: return g_angle_keep_rule.load(std::memory_order_relaxed);
        case OVS_MATCH_VARIANT_ANGLE_TIE_ORDER: return g_angle_tie_order.load(std::memory_order_relaxed);
        case OVS_MATCH_VARIANT_BF_FRAME_MASK: return g_bf_frame_mask.load(std::memory_order_relaxed);
        default: return -1;
    }
}

}   // extern "C"
