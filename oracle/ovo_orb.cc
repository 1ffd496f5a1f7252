// ovo_orb.cc -- CPU ORACLE (test infrastructure, see ovo_oracle.h): feature::orb_extractor restated from spec.
// PARITY UNPINNED: upstream sources are absent (/root/reference/README.md:1-4); citations name the EXPECTED
// upstream path and the SURVEY.md section 8(a) row that describes the behaviour being restated.
#include "ovo_oracle.h"
#include "../include/ovs_detmath.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <list>
#include <utility>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int kFastPatchSize = 31;      // orb_extractor::fast_patch_size_
constexpr int kFastHalfPatch = 15;      // fast_half_patch_size_
constexpr int kOrbPatchRadius = 19;     // orb_extractor::orb_patch_radius_
constexpr int kCellSize = 64;           // compute_fast_keypoints: cell_size
constexpr int kCellOverlap = 6;         // compute_fast_keypoints: overlap

// round-half-to-even of a float, as cvRound (SSE cvtss2si under the default MXCSR mode / lrintf).
inline int cv_round(float v) { return (int)std::nearbyintf(v); }
inline int cv_round(double v) { return (int)std::nearbyint(v); }
inline int cv_floor(float v) { return (int)std::floor(v); }

static const int8_t kPattern[256 * 4] = {
#include "orb_pattern.inc"
};

// ------------------------------------------------------------------------------------------------------------
// A0  orb_params tables (expected: src/openvslam/feature/orb_params.cc, orb_extractor.cc ctor/initialize())
// ------------------------------------------------------------------------------------------------------------
void calc_tables(const ovo_orb_params& p, std::vector<float>& sf, std::vector<float>& isf, std::vector<float>& ls,
                 std::vector<float>& ils, std::vector<int>& npl, int* u_max) {
    const int L = p.num_levels;
    sf.assign(L, 1.0f);
    isf.assign(L, 1.0f);
    ls.assign(L, 1.0f);
    ils.assign(L, 1.0f);
    // cumulative FLOAT product, not pow
    for (int l = 1; l < L; ++l) sf[l] = p.scale_factor * sf[l - 1];
    for (int l = 0; l < L; ++l) {
        isf[l] = 1.0f / sf[l];
        ls[l] = sf[l] * sf[l];
        ils[l] = 1.0f / ls[l];
    }
    // geometric share per level, round(); last level takes the remainder
    npl.assign(L, 0);
    double desired = p.max_num_keypts * (1.0 - 1.0 / p.scale_factor) /
                     (1.0 - std::pow(1.0 / p.scale_factor, static_cast<double>(L)));
    int total = 0;
    for (int l = 0; l < L - 1; ++l) {
        npl[l] = (int)std::round(desired);
        total += npl[l];
        desired *= 1.0 / p.scale_factor;
    }
    npl[L - 1] = std::max(p.max_num_keypts - total, 0);
    // u_max: circular patch half-widths, as OpenCV ORB / ORB-SLAM2
    if (u_max) {
        for (int i = 0; i < 16; ++i) u_max[i] = 0;
        const int vmax = (int)std::floor(kFastHalfPatch * std::sqrt(2.0) / 2 + 1);
        const int vmin = (int)std::ceil(kFastHalfPatch * std::sqrt(2.0) / 2);
        for (int v = 0; v <= vmax; ++v)
            u_max[v] = (int)std::round(std::sqrt((double)kFastHalfPatch * kFastHalfPatch - (double)v * v));
        for (int v = kFastHalfPatch, v0 = 0; v >= vmin; --v) {
            while (u_max[v0] == u_max[v0 + 1]) ++v0;
            u_max[v] = v0;
            ++v0;
        }
    }
}

void pyramid_sizes(const std::vector<float>& sf, int rows, int cols, std::vector<int>& lr, std::vector<int>& lc) {
    // compute_image_pyramid: size from the ORIGINAL image and the float scale factor promoted to double
    const int L = (int)sf.size();
    lr.assign(L, rows);
    lc.assign(L, cols);
    for (int l = 1; l < L; ++l) {
        const double scale = sf[l];
        lc[l] = (int)std::round(cols * 1.0 / scale);
        lr[l] = (int)std::round(rows * 1.0 / scale);
    }
}

// ------------------------------------------------------------------------------------------------------------
// A1  cv::resize INTER_LINEAR, CV_8UC1 (OpenCV imgproc resize.cpp: resizeGeneric_ + HResizeLinear<uchar,int,short,
//     INTER_RESIZE_COEF_SCALE=2048> + VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>)
// ------------------------------------------------------------------------------------------------------------
struct ResizeTab {
    std::vector<int> ofs;        // source index of the first tap
    std::vector<short> a0, a1;   // 11-bit coefficients
};
ResizeTab make_resize_tab(int ssize, int dsize) {
    ResizeTab t;
    t.ofs.resize(dsize);
    t.a0.resize(dsize);
    t.a1.resize(dsize);
    const double scale = (double)ssize / dsize;
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        t.ofs[d] = s;
        const float c0 = 1.f - f, c1 = f;
        t.a0[d] = (short)cv_round(c0 * 2048.f);
        t.a1[d] = (short)cv_round(c1 * 2048.f);
    }
    return t;
}
void resize_linear_u8(const uint8_t* src, int srows, int scols, size_t sstride, uint8_t* dst, int drows, int dcols,
                      size_t dstride) {
    const ResizeTab tx = make_resize_tab(scols, dcols);
    const ResizeTab ty = make_resize_tab(srows, drows);
    std::vector<int> r0(dcols), r1(dcols);
    for (int dy = 0; dy < drows; ++dy) {
        const int sy0 = ty.ofs[dy];
        const int sy1 = std::min(sy0 + 1, srows - 1);
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dcols; ++dx) {
            const int sx0 = tx.ofs[dx];
            const int sx1 = std::min(sx0 + 1, scols - 1);
            r0[dx] = S0[sx0] * tx.a0[dx] + S0[sx1] * tx.a1[dx];
            r1[dx] = S1[sx0] * tx.a0[dx] + S1[sx1] * tx.a1[dx];
        }
        const int b0 = ty.a0[dy], b1 = ty.a1[dy];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dcols; ++dx) {
            const int v = (((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// A3  cv::FAST TYPE_9_16 with non-max suppression (OpenCV features2d fast.cpp FAST_t<16>, fast_score.cpp
//     cornerScore<16>)
// ------------------------------------------------------------------------------------------------------------
const int kRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int corner_score_16(const uint8_t* ptr, const int* pixel, int threshold) {
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

struct FastKp {
    int x, y, score;
};

void fast9_16(const uint8_t* img, int rows, int cols, size_t stride, int threshold, bool nonmax,
              std::vector<FastKp>& out) {
    out.clear();
    if (rows < 7 || cols < 7) return;
    const int K = 8, N = 16 + K + 1;
    int pixel[25];
    for (int k = 0; k < 16; ++k) pixel[k] = kRing[k][0] + kRing[k][1] * (int)stride;
    for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t tab[512];
    for (int i = -255; i <= 255; ++i) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

    // score plane of the (sub-)image: 0 where not a corner; the 3-px frame is never tested (stays 0). This is the
    // same data OpenCV keeps in its 3-row rolling buffer.
    std::vector<int> score((size_t)rows * cols, 0);
    std::vector<uint8_t> is_corner((size_t)rows * cols, 0);
    for (int i = 3; i < rows - 3; ++i) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        for (int j = 3; j < cols - 3; ++j, ++ptr) {
            const int v = ptr[0];
            const uint8_t* t = &tab[0] - v + 255;
            int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
            if (d == 0) continue;
            d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
            d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
            d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
            if (d == 0) continue;
            d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
            d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
            d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
            d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
            bool corner = false;
            if (d & 1) {
                const int vt = v - threshold;
                int count = 0;
                for (int k = 0; k < N; ++k) {
                    const int x = ptr[pixel[k]];
                    if (x < vt) {
                        if (++count > K) { corner = true; break; }
                    } else
                        count = 0;
                }
            }
            if (!corner && (d & 2)) {
                const int vt = v + threshold;
                int count = 0;
                for (int k = 0; k < N; ++k) {
                    const int x = ptr[pixel[k]];
                    if (x > vt) {
                        if (++count > K) { corner = true; break; }
                    } else
                        count = 0;
                }
            }
            if (corner) {
                is_corner[(size_t)i * cols + j] = 1;
                score[(size_t)i * cols + j] = corner_score_16(ptr, pixel, threshold);
            }
        }
    }
    for (int i = 3; i < rows - 3; ++i) {
        for (int j = 3; j < cols - 3; ++j) {
            if (!is_corner[(size_t)i * cols + j]) continue;
            const int* c = &score[(size_t)i * cols + j];
            const int s = c[0];
            if (!nonmax || (s > c[1] && s > c[-1] && s > c[-cols - 1] && s > c[-cols] && s > c[-cols + 1] &&
                            s > c[cols - 1] && s > c[cols] && s > c[cols + 1]))
                out.push_back({j, i, s});
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// A4  distribute_keypoints_via_tree / initialize_nodes / assign_child_nodes / find_keypoints_with_max_response
//     (expected: src/openvslam/feature/orb_extractor.cc, orb_extractor_node.{h,cc}); lineage: ORB-SLAM2
//     ORBextractor::DistributeOctTree.
//     TIE RULE (upstream sorts std::pair<int,node*> so equal counts are ordered by heap address, which is
//     implementation-defined): equal counts are ordered LATER-CREATED NODE FIRST, i.e. the order a monotone bump
//     allocator would give upstream's descending pointer comparison.
// ------------------------------------------------------------------------------------------------------------
struct Cand {
    float x, y, response;
    int idx;
};
struct Node {
    int bx = 0, by = 0, ex = 0, ey = 0;   // pt_begin_, pt_end_
    std::vector<Cand> keypts;
    bool is_leaf = false;
    std::list<Node>::iterator iter;
    long seq = 0;   // creation sequence number (tie rule)
};

std::array<Node, 4> divide_node(const Node& n) {
    const int half_x = (int)std::ceil((n.ex - n.bx) / 2.0);
    const int half_y = (int)std::ceil((n.ey - n.by) / 2.0);
    std::array<Node, 4> c;
    const int cx = n.bx + half_x, cy = n.by + half_y;
    c[0].bx = n.bx; c[0].by = n.by; c[0].ex = cx;   c[0].ey = cy;
    c[1].bx = cx;   c[1].by = n.by; c[1].ex = n.ex; c[1].ey = cy;
    c[2].bx = n.bx; c[2].by = cy;   c[2].ex = cx;   c[2].ey = n.ey;
    c[3].bx = cx;   c[3].by = cy;   c[3].ex = n.ex; c[3].ey = n.ey;
    for (const auto& k : n.keypts) {
        unsigned idx = 0;
        if ((float)cx <= k.x) idx += 1;
        if ((float)cy <= k.y) idx += 2;
        c[idx].keypts.push_back(k);
    }
    for (auto& ch : c)
        if (ch.keypts.size() == 1) ch.is_leaf = true;
    return c;
}

void assign_child_nodes(const std::array<Node, 4>& children, std::list<Node>& nodes,
                        std::vector<std::pair<int, Node*>>& pool, long& seq) {
    for (const auto& ch : children) {
        if (ch.keypts.empty()) continue;
        nodes.push_front(ch);
        nodes.front().seq = seq++;
        nodes.front().iter = nodes.begin();
        if (ch.keypts.size() == 1) continue;
        pool.emplace_back((int)ch.keypts.size(), &nodes.front());
    }
}

// switch_factor / tie_earlier_first: the two implementation-dependent choices of this stage as run-time variants (ORACLE_SPEC rules 6, 7):
// ORB-SLAM2's `N < nodes + 3 * splittable` vs a factor of 1, and the order of equal-count nodes in the sorted phase
std::vector<int> distribute_via_tree(const std::vector<Cand>& cands, int min_x, int max_x, int min_y, int max_y,
                                     unsigned num_keypts, unsigned switch_factor = 3, bool tie_earlier_first = false) {
    std::vector<int> result;
    if (cands.empty()) return result;
    // ---- initialize_nodes: a row (landscape) or column (portrait) of near-square root patches
    const double ratio = static_cast<double>(max_x - min_x) / (max_y - min_y);
    double delta_x, delta_y;
    unsigned num_x_grid, num_y_grid;
    if (ratio > 1) {
        num_x_grid = (unsigned)std::round(ratio);
        num_y_grid = 1;
        delta_x = static_cast<double>(max_x - min_x) / num_x_grid;
        delta_y = max_y - min_y;
    } else {
        num_x_grid = 1;
        num_y_grid = (unsigned)std::round(1 / ratio);
        delta_x = max_x - min_x;
        delta_y = static_cast<double>(max_y - min_y) / num_y_grid;
    }
    const unsigned num_initial = num_x_grid * num_y_grid;
    std::list<Node> nodes;
    std::vector<Node*> initial(num_initial);
    long seq = 0;
    for (unsigned i = 0; i < num_initial; ++i) {
        Node n;
        const unsigned ix = i % num_x_grid, iy = i / num_x_grid;
        n.bx = (int)(delta_x * ix);   // cv::Point2i(double,double): truncation
        n.by = (int)(delta_y * iy);
        n.ex = (int)(delta_x * (ix + 1));
        n.ey = (int)(delta_y * (iy + 1));
        n.seq = seq++;
        nodes.push_back(n);
        initial[i] = &nodes.back();
    }
    for (const auto& k : cands) {
        unsigned ix = (unsigned)(k.x / delta_x);
        unsigned iy = (unsigned)(k.y / delta_y);
        ix = std::min(ix, num_x_grid - 1);   // guard only; cannot trigger for in-range candidates
        iy = std::min(iy, num_y_grid - 1);
        initial[ix + iy * num_x_grid]->keypts.push_back(k);
    }
    for (auto it = nodes.begin(); it != nodes.end();) {
        if (it->keypts.size() == 1) {
            it->is_leaf = true;
            ++it;
        } else if (it->keypts.empty())
            it = nodes.erase(it);
        else
            ++it;
    }

    std::vector<std::pair<int, Node*>> pool;
    bool is_filled = false;
    while (true) {
        const size_t prev_size = nodes.size();
        auto it = nodes.begin();
        pool.clear();
        while (it != nodes.end()) {
            if (it->is_leaf) { ++it; continue; }
            const auto children = divide_node(*it);
            assign_child_nodes(children, nodes, pool, seq);
            it = nodes.erase(it);
        }
        if (num_keypts <= nodes.size() || nodes.size() == prev_size) { is_filled = true; break; }
        if (num_keypts < nodes.size() + switch_factor * pool.size()) { is_filled = false; break; }
    }
    while (!is_filled) {
        const size_t prev_size = nodes.size();
        auto prev_pool = pool;
        pool.clear();
        // descending by keypoint count; ties: later-created node first (see TIE RULE above)
        std::sort(prev_pool.begin(), prev_pool.end(), [tie_earlier_first](const std::pair<int, Node*>& a, const std::pair<int, Node*>& b) {
            if (a.first != b.first) return a.first > b.first;
            return tie_earlier_first ? a.second->seq < b.second->seq : a.second->seq > b.second->seq;
        });
        for (const auto& pn : prev_pool) {
            const auto children = divide_node(*pn.second);
            assign_child_nodes(children, nodes, pool, seq);
            nodes.erase(pn.second->iter);
            if (num_keypts <= nodes.size()) { is_filled = true; break; }
        }
        if (is_filled || num_keypts <= nodes.size() || nodes.size() == prev_size) { is_filled = true; break; }
    }
    // ---- find_keypoints_with_max_response: list order; first maximum wins
    result.reserve(nodes.size());
    for (const auto& n : nodes) {
        const Cand* best = &n.keypts[0];
        float max_resp = best->response;
        for (size_t k = 1; k < n.keypts.size(); ++k)
            if (n.keypts[k].response > max_resp) { best = &n.keypts[k]; max_resp = best->response; }
        result.push_back(best->idx);
    }
    return result;
}

// ------------------------------------------------------------------------------------------------------------
// A5  ic_angle (expected: orb_extractor.cc) + cv::fastAtan2 scalar form (OpenCV core mathfuncs_core)
// ------------------------------------------------------------------------------------------------------------
float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

float ic_angle(const uint8_t* img, size_t stride, int x, int y, const int* u_max) {
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int u = -kFastHalfPatch; u <= kFastHalfPatch; ++u) m_10 += u * center[u];
    const int step = (int)stride;
    for (int v = 1; v <= kFastHalfPatch; ++v) {
        int v_sum = 0;
        const int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

// ------------------------------------------------------------------------------------------------------------
// A6  cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) for CV_8U: OpenCV >= 3.4.1 fixed-point path
//     (ufixedpoint16 8.8 taps that sum to 256; row pass u8*tap -> 8.8, column pass 8.8*tap -> 16.16,
//     round-half-up to u8). Taps: OpenCV's getGaussianKernelFixedPoint_ED rule -- from the outside in,
//     v_i = cvRound(256*g_i + err), err carried to the next tap, mirrored, the centre tap takes 256 - 2*sum:
//     256*g = 17.96, 33.55, 48.82, 55.32 -> 18 (err -.04), 34 (err -.49), 48, centre 56
//     (tests/test_oracle_kat.py::test_blur_taps_and_rounding re-derives them). OpenCV-version dependent upstream
//     (3.4.1..3.4.6 round every tap independently): this definition is the oracle's.
// ------------------------------------------------------------------------------------------------------------
const int kGauss7[7] = {18, 34, 48, 56, 48, 34, 18};
inline const int* ovo_kGauss7Default() { return kGauss7; }
// variant 1 (ORACLE_SPEC rule 10): every tap rounded on its own, cvRound(256 g_i) -- OpenCV's 8-bit fixed-point separable filter before
// 3.4.1's getGaussianKernelFixedPoint_ED (and 3.4.1 .. 3.4.6 as recalled). The taps sum to 257, so the result can reach 257: saturate_cast.
const int kGauss7Indep[7] = {18, 34, 49, 55, 49, 34, 18};
inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}
void gaussian_blur_7x7(const uint8_t* src, int rows, int cols, size_t sstride, uint8_t* dst, size_t dstride, int taps_variant = 0) {
    const int* const kGauss7 = taps_variant == 1 ? kGauss7Indep : ovo_kGauss7Default();
    std::vector<uint16_t> h((size_t)rows * cols);
    for (int y = 0; y < rows; ++y) {
        const uint8_t* S = src + (size_t)y * sstride;
        for (int x = 0; x < cols; ++x) {
            unsigned acc = 0;
            for (int k = -3; k <= 3; ++k) acc += (unsigned)kGauss7[k + 3] * S[reflect101(x + k, cols)];
            h[(size_t)y * cols + x] = (uint16_t)acc;   // <= 255 * 257 = 65535
        }
    }
    for (int y = 0; y < rows; ++y) {
        uint8_t* D = dst + (size_t)y * dstride;
        for (int x = 0; x < cols; ++x) {
            uint32_t acc = 0;
            for (int k = -3; k <= 3; ++k) acc += (uint32_t)kGauss7[k + 3] * h[(size_t)reflect101(y + k, rows) * cols + x];
            const uint32_t v = (acc + 32768u) >> 16;
            D[x] = (uint8_t)(v > 255u ? 255u : v);   // (only the 257-sum variant can exceed 255)
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// A7  util::cos / util::sin (expected: src/openvslam/util/trigonometric.h -- 3-term even polynomial after range
//     reduction) and compute_orb_descriptor (expected: orb_extractor.cc). Deterministic float ops only; this is
//     the "shared sin/cos" of SURVEY section 7 hard part 4: the HIP kernel evaluates the identical op sequence.
// ------------------------------------------------------------------------------------------------------------
constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr float kHalfPi = 1.57079632679489661923f;
constexpr float kThreeHalfPi = 4.71238898038468985769f;
constexpr float kInvTwoPi = 0.15915494309189533577f;

inline float poly_cos(float v) {
    const float c1 = 0.99940307f, c2 = -0.49558072f, c3 = 0.03679168f;
    const float v2 = v * v;
    return c1 + v2 * (c2 + c3 * v2);
}
float util_cos(float v) {
    v = v - std::floor(v * kInvTwoPi) * kTwoPi;
    v = (0.0f < v) ? v : -v;
    if (v < kHalfPi) return poly_cos(v);
    if (v < kPi) return -poly_cos(kPi - v);
    if (v < kThreeHalfPi) return -poly_cos(v - kPi);
    return poly_cos(kTwoPi - v);
}
float util_sin(float v) { return util_cos(kHalfPi - v); }

// trig_variant (ORACLE_SPEC rule 11): 0 util::cos / util::sin (OpenVSLAM's util/trigonometric.h polynomial), 1 std::cos / std::sin on the float
// angle as glibc computes them (ovs_det_cosf / ovs_det_sinf of include/ovs_detmath.h, exhaustively equal to glibc's on [0, 6.3])
void orb_descriptor(const uint8_t* blurred, size_t stride, int x, int y, float angle_deg, uint8_t* desc, int trig_variant = 0) {
    // upstream: `const float angle = keypt.angle * M_PI / 180.0;` -- float * double / double evaluated in double, rounded to float ONCE
    // (ORACLE_SPEC rule 11; not ORB-SLAM2's single float multiply by factorPI, which differs by 1 ulp for a fraction of the angles)
    const float angle = (float)((double)angle_deg * M_PI / 180.0);
    const float cos_a = trig_variant ? ovs_det_cosf(angle) : util_cos(angle), sin_a = trig_variant ? ovs_det_sinf(angle) : util_sin(angle);
    const uint8_t* center = blurred + (size_t)y * stride + x;
    const int step = (int)stride;
    auto value = [&](int px, int py) -> int {
        const float fx = (float)px, fy = (float)py;
        const int dy = cv_round(fx * sin_a + fy * cos_a);
        const int dx = cv_round(fx * cos_a - fy * sin_a);
        return center[dy * step + dx];
    };
    for (int i = 0; i < 32; ++i) {
        unsigned val = 0;
        for (int b = 0; b < 8; ++b) {
            const int8_t* p = kPattern + (i * 8 + b) * 4;
            const int t0 = value(p[0], p[1]), t1 = value(p[2], p[3]);
            val |= (unsigned)(t0 < t1) << b;
        }
        desc[i] = (uint8_t)val;
    }
}

}   // namespace

// ============================================================================================================
// A9  orb_extractor (expected: src/openvslam/feature/orb_extractor.{h,cc})
// ============================================================================================================
struct ovo_orb {
    ovo_orb_params p;
    std::vector<float> sf, isf, ls, ils;
    std::vector<int> npl;
    int u_max[16];
    int threads = 1;
    int tree_switch_factor = 3, tree_tie_earlier_first = 0, blur_taps = 0, trig = 0;   // ORACLE_SPEC rules 6, 7, 10, 11 as run-time variants
    // observables of the last extract
    std::vector<int> lrows, lcols;
    std::vector<std::vector<uint8_t>> pyr, blurred;
    std::vector<std::vector<FastKp>> cands;   // level-image coordinates
    std::vector<int> nkp;
};

namespace {

void compute_level(ovo_orb* h, int level, const uint8_t* mask, size_t mask_stride, std::vector<ovo_keypoint>& kps) {
    const int rows = h->lrows[level], cols = h->lcols[level];
    const uint8_t* img = h->pyr[level].data();
    const float scale_factor = h->sf[level];
    auto is_in_mask = [&](unsigned y, unsigned x) {
        return mask[(size_t)(unsigned)(y * scale_factor) * mask_stride + (unsigned)(x * scale_factor)] == 0;
    };
    // ---- compute_fast_keypoints: cell loop
    std::vector<FastKp>& all = h->cands[level];
    all.clear();
    kps.clear();
    const int min_border_x = kOrbPatchRadius, min_border_y = kOrbPatchRadius;
    const int max_border_x = cols - kOrbPatchRadius, max_border_y = rows - kOrbPatchRadius;
    if (max_border_x <= min_border_x || max_border_y <= min_border_y) return;
    const int width = max_border_x - min_border_x, height = max_border_y - min_border_y;
    const int num_cols = width / kCellSize + 1, num_rows = height / kCellSize + 1;
    std::vector<FastKp> in_cell;
    std::vector<Cand> to_distribute;   // coordinates relative to (min_border_x, min_border_y)
    for (int i = 0; i < num_rows; ++i) {
        const int min_y = min_border_y + i * kCellSize;
        if (max_border_y - kCellOverlap <= min_y) continue;
        int max_y = min_y + kCellSize + kCellOverlap;
        if (max_border_y < max_y) max_y = max_border_y;
        for (int j = 0; j < num_cols; ++j) {
            const int min_x = min_border_x + j * kCellSize;
            if (max_border_x - kCellOverlap <= min_x) continue;
            int max_x = min_x + kCellSize + kCellOverlap;
            if (max_border_x < max_x) max_x = max_border_x;
            if (mask) {
                if (is_in_mask(min_y, min_x) || is_in_mask(max_y, min_x) || is_in_mask(min_y, max_x) ||
                    is_in_mask(max_y, max_x))
                    continue;
            }
            const uint8_t* cell = img + (size_t)min_y * cols + min_x;
            fast9_16(cell, max_y - min_y, max_x - min_x, cols, h->p.ini_fast_thr, true, in_cell);
            if (in_cell.empty()) fast9_16(cell, max_y - min_y, max_x - min_x, cols, h->p.min_fast_thr, true, in_cell);
            if (in_cell.empty()) continue;
            for (const auto& k : in_cell) {
                const int rx = k.x + j * kCellSize, ry = k.y + i * kCellSize;   // relative to min_border
                if (mask && is_in_mask(min_border_y + ry, min_border_x + rx)) continue;
                to_distribute.push_back({(float)rx, (float)ry, (float)k.score, (int)to_distribute.size()});
                all.push_back({min_border_x + rx, min_border_y + ry, k.score});
            }
        }
    }
    // ---- distribute, then translate / octave / size
    const std::vector<int> sel =
        distribute_via_tree(to_distribute, min_border_x, max_border_x, min_border_y, max_border_y, (unsigned)h->npl[level],
                            (unsigned)h->tree_switch_factor, h->tree_tie_earlier_first != 0);
    const unsigned scaled_patch_size = (unsigned)(kFastPatchSize * scale_factor);
    kps.reserve(sel.size());
    for (int idx : sel) {
        const Cand& c = to_distribute[idx];
        ovo_keypoint k;
        k.x = c.x + min_border_x;
        k.y = c.y + min_border_y;
        k.size = (float)scaled_patch_size;
        k.response = c.response;
        k.octave = level;
        k.class_id = -1;
        // ---- compute_orientation (on the UNBLURRED level)
        k.angle = ic_angle(img, cols, cv_round(k.x), cv_round(k.y), h->u_max);
        kps.push_back(k);
    }
}

}   // namespace

extern "C" {

int ovo_orb_tables(const ovo_orb_params* p, float* sf, float* isf, float* ls, float* ils, int32_t* npl, int32_t* u_max16) {
    if (!p || p->num_levels < 1 || p->num_levels > OVO_MAX_LEVELS) return -1;
    std::vector<float> a, b, c, d;
    std::vector<int> n;
    int um[16];
    calc_tables(*p, a, b, c, d, n, um);
    for (int l = 0; l < p->num_levels; ++l) {
        if (sf) sf[l] = a[l];
        if (isf) isf[l] = b[l];
        if (ls) ls[l] = c[l];
        if (ils) ils[l] = d[l];
        if (npl) npl[l] = n[l];
    }
    if (u_max16) std::memcpy(u_max16, um, sizeof(um));
    return 0;
}

int ovo_pyramid_sizes(const ovo_orb_params* p, int rows, int cols, int32_t* lr, int32_t* lc) {
    if (!p || p->num_levels < 1 || p->num_levels > OVO_MAX_LEVELS) return -1;
    std::vector<float> a, b, c, d;
    std::vector<int> n, r, cc;
    calc_tables(*p, a, b, c, d, n, nullptr);
    pyramid_sizes(a, rows, cols, r, cc);
    for (int l = 0; l < p->num_levels; ++l) { lr[l] = r[l]; lc[l] = cc[l]; }
    return 0;
}

int ovo_resize_linear_u8(const uint8_t* src, int srows, int scols, size_t sstride, uint8_t* dst, int drows, int dcols,
                         size_t dstride) {
    if (!src || !dst || srows < 1 || scols < 1 || drows < 1 || dcols < 1) return -1;
    resize_linear_u8(src, srows, scols, sstride, dst, drows, dcols, dstride);
    return 0;
}

int ovo_fast9_16(const uint8_t* img, int rows, int cols, size_t stride, int threshold, int nonmax, int32_t* xs, int32_t* ys,
                 int32_t* scores, int cap) {
    std::vector<FastKp> out;
    fast9_16(img, rows, cols, stride, threshold, nonmax != 0, out);
    const int n = (int)out.size();
    for (int i = 0; i < n && i < cap; ++i) { xs[i] = out[i].x; ys[i] = out[i].y; scores[i] = out[i].score; }
    return n;
}

int ovo_distribute_via_tree(const float* xs, const float* ys, const float* responses, int n, int min_x, int max_x, int min_y,
                            int max_y, int num_keypts, int32_t* out_idx, int cap) {
    std::vector<Cand> c(n);
    for (int i = 0; i < n; ++i) c[i] = {xs[i], ys[i], responses[i], i};
    const auto sel = distribute_via_tree(c, min_x, max_x, min_y, max_y, (unsigned)num_keypts);
    for (size_t i = 0; i < sel.size() && (int)i < cap; ++i) out_idx[i] = sel[i];
    return (int)sel.size();
}

int ovo_distribute_via_tree_v(const float* xs, const float* ys, const float* responses, int n, int min_x, int max_x, int min_y, int max_y,
                              int num_keypts, int switch_factor, int tie_earlier_first, int32_t* out_idx, int cap) {
    std::vector<Cand> c(n);
    for (int i = 0; i < n; ++i) c[i] = {xs[i], ys[i], responses[i], i};
    const auto sel = distribute_via_tree(c, min_x, max_x, min_y, max_y, (unsigned)num_keypts, (unsigned)switch_factor, tie_earlier_first != 0);
    for (size_t i = 0; i < sel.size() && (int)i < cap; ++i) out_idx[i] = sel[i];
    return (int)sel.size();
}

float ovo_fast_atan2(float y, float x) { return fast_atan2(y, x); }
float ovo_ic_angle(const uint8_t* img, size_t stride, int x, int y, const int32_t* u_max16) { return ic_angle(img, stride, x, y, u_max16); }
int ovo_gaussian_blur_7x7(const uint8_t* src, int rows, int cols, size_t sstride, uint8_t* dst, size_t dstride) {
    if (!src || !dst || rows < 1 || cols < 1) return -1;
    gaussian_blur_7x7(src, rows, cols, sstride, dst, dstride);
    return 0;
}
int ovo_gaussian_blur_7x7_v(const uint8_t* src, int rows, int cols, size_t sstride, uint8_t* dst, size_t dstride, int taps_variant) {
    if (!src || !dst || rows < 1 || cols < 1 || taps_variant < 0 || taps_variant > 1) return -1;
    gaussian_blur_7x7(src, rows, cols, sstride, dst, dstride, taps_variant);
    return 0;
}
float ovo_util_cos(float v) { return util_cos(v); }
float ovo_util_sin(float v) { return util_sin(v); }
int ovo_orb_descriptor(const uint8_t* blurred, size_t stride, int x, int y, float angle_deg, uint8_t* desc32) {
    orb_descriptor(blurred, stride, x, y, angle_deg, desc32);
    return 0;
}
int ovo_orb_descriptor_v(const uint8_t* blurred, size_t stride, int x, int y, float angle_deg, uint8_t* desc32, int trig_variant) {
    if (trig_variant < 0 || trig_variant > 1) return -1;
    orb_descriptor(blurred, stride, x, y, angle_deg, desc32, trig_variant);
    return 0;
}
// the shared sinf / cosf restatement against THIS machine's libm over the float bit patterns [lo_bits, hi_bits]: number of values where either differs
long ovo_trig_mismatches_vs_libm(uint32_t lo_bits, uint32_t hi_bits) {
    long bad = 0;
    for (uint64_t u = lo_bits; u <= hi_bits; ++u) {
        const uint32_t v = (uint32_t)u;
        float x;
        std::memcpy(&x, &v, 4);
        const float a = ovs_det_sinf(x), b = sinf(x), c = ovs_det_cosf(x), d = cosf(x);
        bad += std::memcmp(&a, &b, 4) != 0 || std::memcmp(&c, &d, 4) != 0;
    }
    return bad;
}
// the kernels' one-multiply deg -> rad against upstream's literal expression over the float bit patterns [lo_bits, hi_bits]: number of differing values
long ovo_deg2rad_mismatches(uint32_t lo_bits, uint32_t hi_bits) {
    long bad = 0;
    for (uint64_t u = lo_bits; u <= hi_bits; ++u) {
        const uint32_t v = (uint32_t)u;
        float a;
        std::memcpy(&a, &v, 4);
        const float lit = (float)((double)a * M_PI / 180.0), one = ovs_det_deg2rad(a);
        bad += std::memcmp(&lit, &one, 4) != 0;
    }
    return bad;
}
float ovo_det_sinf(float v) { return ovs_det_sinf(v); }
float ovo_det_cosf(float v) { return ovs_det_cosf(v); }
const int8_t* ovo_orb_pattern(void) { return kPattern; }

ovo_orb* ovo_orb_create(const ovo_orb_params* p) {
    if (!p || p->num_levels < 1 || p->num_levels > OVO_MAX_LEVELS || p->scale_factor <= 1.0f) return nullptr;
    ovo_orb* h = new ovo_orb();
    h->p = *p;
    calc_tables(*p, h->sf, h->isf, h->ls, h->ils, h->npl, h->u_max);
    return h;
}
void ovo_orb_destroy(ovo_orb* h) { delete h; }
void ovo_orb_set_threads(ovo_orb* h, int n) { h->threads = std::max(1, n); }
// which: 0 quad-tree switch factor (3 | 1), 1 equal-count tie order (0 later-created first | 1 earlier-created first), 2 blur taps (0 | 1),
// 3 steering trigonometry (0 util::cos / util::sin | 1 libm's cosf / sinf)
int ovo_orb_set_variant(ovo_orb* h, int which, int value) {
    if (!h) return -1;
    if (which == 0 && (value == 3 || value == 1)) h->tree_switch_factor = value;
    else if (which == 1 && (value == 0 || value == 1)) h->tree_tie_earlier_first = value;
    else if (which == 2 && (value == 0 || value == 1)) h->blur_taps = value;
    else if (which == 3 && (value == 0 || value == 1)) h->trig = value;
    else return -1;
    return 0;
}

int ovo_orb_extract(ovo_orb* h, const uint8_t* img, int rows, int cols, size_t stride, const uint8_t* mask, size_t mask_stride,
                    ovo_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    if (!h || !n_out) return -1;
    *n_out = 0;
    if (!img || rows <= 0 || cols <= 0) return 0;   // empty image: early return (SURVEY 8(b) errors)
    const int L = h->p.num_levels;
    pyramid_sizes(h->sf, rows, cols, h->lrows, h->lcols);
    h->pyr.assign(L, {});
    h->blurred.assign(L, {});
    h->cands.assign(L, {});
    h->nkp.assign(L, 0);
    // ---- compute_image_pyramid (level 0 is the caller's image; each level from the PREVIOUS level)
    h->pyr[0].resize((size_t)rows * cols);
    for (int y = 0; y < rows; ++y) std::memcpy(&h->pyr[0][(size_t)y * cols], img + (size_t)y * stride, cols);
    for (int l = 1; l < L; ++l) {
        if (h->lrows[l] < 1 || h->lcols[l] < 1) return -2;
        h->pyr[l].resize((size_t)h->lrows[l] * h->lcols[l]);
        resize_linear_u8(h->pyr[l - 1].data(), h->lrows[l - 1], h->lcols[l - 1], h->lcols[l - 1], h->pyr[l].data(), h->lrows[l],
                         h->lcols[l], h->lcols[l]);
    }
    std::vector<std::vector<ovo_keypoint>> all_kps(L);
    std::vector<std::vector<uint8_t>> all_desc(L);
#pragma omp parallel for schedule(dynamic, 1) num_threads(h->threads)
    for (int l = 0; l < L; ++l) {
        compute_level(h, l, mask, mask_stride, all_kps[l]);
        auto& k = all_kps[l];
        h->nkp[l] = (int)k.size();
        if (k.empty()) continue;
        // ---- blur a copy of the level, describe, then correct_keypoint_scale
        h->blurred[l].resize(h->pyr[l].size());
        gaussian_blur_7x7(h->pyr[l].data(), h->lrows[l], h->lcols[l], h->lcols[l], h->blurred[l].data(), h->lcols[l], h->blur_taps);
        all_desc[l].resize(k.size() * 32);
        for (size_t i = 0; i < k.size(); ++i) {
            orb_descriptor(h->blurred[l].data(), h->lcols[l], cv_round(k[i].x), cv_round(k[i].y), k[i].angle, &all_desc[l][i * 32], h->trig);
            k[i].x *= h->sf[l];
            k[i].y *= h->sf[l];
        }
    }
    int n = 0;
    for (int l = 0; l < L; ++l) {
        for (size_t i = 0; i < all_kps[l].size(); ++i) {
            if (n < cap) {
                if (kps) kps[n] = all_kps[l][i];
                if (desc) std::memcpy(desc + (size_t)n * 32, &all_desc[l][i * 32], 32);
            }
            ++n;
        }
    }
    *n_out = n;
    return n > cap ? 1 : 0;   // 1: truncated (count still reported)
}

int ovo_orb_level_size(const ovo_orb* h, int level, int* rows, int* cols) {
    if (!h || level < 0 || level >= (int)h->lrows.size()) return -1;
    *rows = h->lrows[level];
    *cols = h->lcols[level];
    return 0;
}
const uint8_t* ovo_orb_level_image(const ovo_orb* h, int level) {
    if (!h || level < 0 || level >= (int)h->pyr.size()) return nullptr;
    return h->pyr[level].data();
}
const uint8_t* ovo_orb_level_blurred(const ovo_orb* h, int level) {
    if (!h || level < 0 || level >= (int)h->blurred.size() || h->blurred[level].empty()) return nullptr;
    return h->blurred[level].data();
}
int ovo_orb_level_num_candidates(const ovo_orb* h, int level) {
    if (!h || level < 0 || level >= (int)h->cands.size()) return -1;
    return (int)h->cands[level].size();
}
int ovo_orb_level_candidates(const ovo_orb* h, int level, int32_t* xs, int32_t* ys, int32_t* scores, int cap) {
    if (!h || level < 0 || level >= (int)h->cands.size()) return -1;
    const auto& c = h->cands[level];
    for (size_t i = 0; i < c.size() && (int)i < cap; ++i) { xs[i] = c[i].x; ys[i] = c[i].y; scores[i] = c[i].score; }
    return (int)c.size();
}
int ovo_orb_level_num_keypts(const ovo_orb* h, int level) {
    if (!h || level < 0 || level >= (int)h->nkp.size()) return -1;
    return h->nkp[level];
}

}   // extern "C"
