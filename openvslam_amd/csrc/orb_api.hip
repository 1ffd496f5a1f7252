// orb_api.hip -- host side of the ORB extractor behind the C ABI (include/ovslam_hip.h): orb_params tables (A0), level
// geometry, cv::resize coefficient tables, buffer ownership, launch sequence. Replaces the body of
// feature::orb_extractor (expected: src/openvslam/feature/orb_extractor.{h,cc}, orb_params.{h,cc}).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ovs_common.h"

namespace ovs {

static thread_local std::string g_last_error;
void set_last_error(const char* what, hipError_t e) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
}
void set_last_error_text(const std::string& what) { g_last_error = what; }
std::atomic<int32_t> g_injected_hip_failures{0}, g_injected_hip_skip{0};

const Tuning& tuning() {
    static const Tuning t = [] {
        auto num = [](const char* name, int dflt) {
            const char* e = std::getenv(name);
            return e && e[0] ? std::atoi(e) : dflt;
        };
        Tuning v;
        v.fast_impl = num("OVS_FAST_IMPL", 1);
        v.fast_map = num("OVS_FAST_MAP", 1);
        v.fast_pf = std::max(0, num("OVS_FAST_PF", 0));
        v.fast_cells = std::max(0, num("OVS_FAST_CELLS", 0));
        v.fast_pad_lds = std::max(0, num("OVS_FAST_PAD_LDS", 0));
        v.fast_timing = std::getenv("OVS_FAST_TIMING") != nullptr;
        v.describe_xcd = num("OVS_DESCRIBE_XCD", 1) != 0;
        v.resolve_wide_from = num("OVS_RESOLVE_WIDE_FROM", 1024);
        v.pose_threads = num("OVS_POSE_THREADS", 0);
        v.pose_groups = std::max(0, num("OVS_POSE_GROUPS", 0));
        v.ba_trace = std::getenv("OVS_BA_TRACE") != nullptr;
        v.pyr_chain = std::max(0, num("OVS_PYR_CHAIN", 2));
        v.pyr_pair = num("OVS_PYR_PAIR", 1) != 0;
        v.chol_resident = num("OVS_CHOL_RESIDENT", 1) != 0;
        v.ba_backsub_edges = num("OVS_BA_BACKSUB_EDGES", 1) != 0;
        v.ba_dev_outliers = num("OVS_BA_DEV_OUTLIERS", 1) != 0;
        return v;
    }();
    return t;
}

static inline int cv_round_f(float v) { return (int)std::nearbyintf(v); }

// cv::resize INTER_LINEAR tables for one axis (OpenCV resize.cpp: fx = (float)((dx+0.5)*scale - 0.5); sx = cvFloor(fx);
// fx -= sx; clamp; ialpha = saturate_cast<short>(cbuf * INTER_RESIZE_COEF_SCALE)).
static void build_taps(int ssize, int dsize, ResizeTap* out) {
    const double scale = (double)ssize / dsize;
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        out[d].o0 = (uint16_t)s;
        out[d].o1 = (uint16_t)std::min(s + 1, ssize - 1);
        out[d].a0 = (int16_t)cv_round_f((1.f - f) * 2048.f);
        out[d].a1 = (int16_t)cv_round_f(f * 2048.f);
    }
}

// Conditions under which k_resize_linear_u8 (v4, orb_pyramid.hip) may run a level: every 128 x 32 output tile's source rectangle fits the
// kernel's LDS tile (48 words x 44 rows, the rectangle starting on a 16-byte boundary), and the four source bytes of every column pair
// (x even, min(x + 1, dcols - 1)) lie inside the two aligned words that start at o0(x) & ~3.
static bool resize_windows_ok(const ResizeTap* xt, int dcols, const ResizeTap* yt, int drows) {
    constexpr int kTw = 128, kTh = 32, kWords = 48, kRows = 44;
    for (int x0 = 0; x0 < dcols; x0 += kTw) {
        const int x1 = std::min(x0 + kTw, dcols) - 1;
        const int lo = xt[x0].o0 & ~15, hi = xt[x1].o1;
        if (hi < lo || (hi - lo) / 4 + 1 > kWords) return false;
    }
    for (int y0 = 0; y0 < drows; y0 += kTh) {
        const int y1 = std::min(y0 + kTh, drows) - 1;
        if (yt[y1].o1 < yt[y0].o0 || yt[y1].o1 - yt[y0].o0 + 1 > kRows) return false;
    }
    for (int x = 0; x < dcols; x += 2) {
        const int xb = std::min(x + 1, dcols - 1);
        const int base = xt[x].o0 & ~3;
        const int o[4] = {xt[x].o0, xt[x].o1, xt[xb].o0, xt[xb].o1};
        for (int k = 0; k < 4; ++k)
            if (o[k] < base || o[k] > base + 7) return false;
        if (xt[x].a0 < 0 || xt[x].a1 < 0 || xt[xb].a0 < 0 || xt[xb].a1 < 0 || xt[x].a0 > 2048 || xt[x].a1 > 2048 || xt[xb].a0 > 2048 || xt[xb].a1 > 2048)
            return false;
    }
    for (int y = 0; y < drows; ++y)
        if (yt[y].a0 < 0 || yt[y].a1 < 0 || yt[y].a0 > 2048 || yt[y].a1 > 2048) return false;
    return true;
}


// The tile grid and the per-level regions of k_pyramid_chain (orb_pyramid.hip) for one geometry. Both axes separately: tile t of T owns
// [t size_l / T, (t + 1) size_l / T) of level l (a partition of every level); it computes, bottom-up, R_{L-1} = owned and
// R_{l-1} = hull(owned_{l-1}, tap footprint of R_l); R_0 is the footprint alone (level 0 is only read). ok = false when a region exceeds what
// the kernel stages (unusual scale factors or level counts): the per-level launches run then.
struct ChainPlanHost {
    bool ok = false;
    int TX = 0, TY = 0, bufA = 0, bufB = 0, tap_cap = 0;
    std::vector<ChainSpan> spans;   // [level][TX x-spans, TY y-spans]
};
static void build_chain_plan(const FrameGeo& geo, const std::vector<ResizeTap>& taps, ChainPlanHost& out) {
    out = ChainPlanHost();
    const int L = geo.num_levels;
    if (L < 2) return;
    const int TX = std::max(1, (geo.lv[0].cols + 127) / 128), TY = std::max(1, (geo.lv[0].rows + 95) / 96);
    out.TX = TX;
    out.TY = TY;
    out.spans.assign((size_t)L * (TX + TY), ChainSpan{0, 0, 0, 0});
    bool ok = true;
    for (int axis = 0; axis < 2; ++axis) {
        const int T = axis ? TY : TX;
        for (int t = 0; t < T; ++t) {
            int c0[OVS_MAX_LEVELS], c1[OVS_MAX_LEVELS], o0[OVS_MAX_LEVELS], o1[OVS_MAX_LEVELS];
            for (int l = 0; l < L; ++l) {
                const int64_t size = axis ? geo.lv[l].rows : geo.lv[l].cols;
                o0[l] = (int)(t * size / T);
                o1[l] = (int)((t + 1) * size / T);
            }
            c0[L - 1] = o0[L - 1];
            c1[L - 1] = o1[L - 1];
            for (int l = L - 1; l >= 1; --l) {
                const ResizeTap* tab = taps.data() + (axis ? geo.lv[l].ytab_off : geo.lv[l].xtab_off);
                int f0 = 0, f1 = 0;
                if (c1[l] > c0[l]) {
                    f0 = tab[c0[l]].o0;
                    f1 = tab[c1[l] - 1].o1 + 1;
                    for (int i = c0[l]; i < c1[l]; ++i) {   // (the tables are monotone; this does not rely on it)
                        f0 = std::min(f0, (int)tab[i].o0);
                        f1 = std::max(f1, (int)tab[i].o1 + 1);
                    }
                }
                if (l - 1 == 0) {
                    o0[0] = o1[0] = 0;
                    c0[0] = axis ? f0 : (f0 & ~3);
                    c1[0] = f1;
                } else if (o1[l - 1] > o0[l - 1] && f1 > f0) {
                    c0[l - 1] = std::min(o0[l - 1], f0);
                    c1[l - 1] = std::max(o1[l - 1], f1);
                } else if (f1 > f0) {
                    c0[l - 1] = f0;
                    c1[l - 1] = f1;
                } else {
                    c0[l - 1] = o0[l - 1];
                    c1[l - 1] = o1[l - 1];
                }
            }
            for (int l = 0; l < L; ++l) {
                if (c1[l] > 32767) ok = false;
                out.spans[(size_t)l * (TX + TY) + (axis ? TX + t : t)] = ChainSpan{(int16_t)c0[l], (int16_t)c1[l], (int16_t)o0[l], (int16_t)o1[l]};
                const int ext = c1[l] - c0[l];
                if (l == 0 ? (axis ? ext > kChainMaxH0 : ext > kChainMaxW0) : (!axis && ext > kChainMaxW)) ok = false;
            }
        }
    }
    // LDS: level 0, 2, 4, .. regions in buffer A, the odd levels in buffer B; the taps of the tile with the most of them
    int bufA = 16, bufB = 16, tap_cap = 0;
    for (int tj = 0; tj < TY; ++tj)
        for (int ti = 0; ti < TX; ++ti) {
            int ntap = 0;
            for (int l = 0; l < L; ++l) {
                const ChainSpan X = out.spans[(size_t)l * (TX + TY) + ti], Y = out.spans[(size_t)l * (TX + TY) + TX + tj];
                const int W = X.c1 - X.c0, H = Y.c1 - Y.c0, bytes = (((W + 3) & ~3) * H + 15) & ~15;
                (l & 1 ? bufB : bufA) = std::max(l & 1 ? bufB : bufA, bytes);
                if (l >= 1) ntap += W + H;
            }
            tap_cap = std::max(tap_cap, ntap);
        }
    out.bufA = bufA;
    out.bufB = bufB;
    out.tap_cap = tap_cap;
    if (chain_lds_bytes(bufA, bufB, tap_cap) > (size_t)kChainMaxLds) ok = false;
    out.ok = ok;
}

// HTapRec of every column pair of a level (ovs_common.h): make_htaps of orb_pyramid.hip with absolute offsets -- the selectors do not depend on the
// tile origin (a multiple of 4), the first source word becomes relative by one subtraction in the kernel.
static void build_htaps(const ResizeTap* xt, int cols, std::vector<HTapRec>& out) {
    for (int x = 0; x < cols; x += 2) {
        const ResizeTap ta = xt[x], tb = xt[std::min(x + 1, cols - 1)];
        HTapRec r = {};
        r.wa = (uint32_t)(ta.o0 >> 2);
        const int base = 4 * (int)r.wa;
        const uint32_t sa0 = (uint32_t)(ta.o0 - base) & 7u, sa1 = (uint32_t)(ta.o1 - base) & 7u;
        const uint32_t sb0 = (uint32_t)(tb.o0 - base) & 7u, sb1 = (uint32_t)(tb.o1 - base) & 7u;
        r.sel_a = sa0 | 0x0c00u | (sa1 << 16) | 0x0c000000u;
        r.sel_b = sb0 | 0x0c00u | (sb1 << 16) | 0x0c000000u;
        r.ca = ((uint32_t)(uint16_t)ta.a0 << 4) | ((uint32_t)(uint16_t)ta.a1 << 20);
        r.cb = ((uint32_t)(uint16_t)tb.a0 << 4) | ((uint32_t)(uint16_t)tb.a1 << 20);
        out.push_back(r);
    }
}

// The regions of k_resize_pair_u8 (orb_pyramid.hip) for the level pair (l2 - 1, l2) computed from level l2 - 2: one PairSpan per tile column and
// per tile row of level l2's 128 x 32 grid, x spans first. Returns false -- the pair then runs as two per-level launches -- unless both levels
// pass v4's window checks, every region fits the kernel's LDS buffers, and the owned ranges partition the middle level.
static bool build_pair_plan(const FrameGeo& geo, const std::vector<ResizeTap>& taps, int l2, std::vector<PairSpan>& out) {
    constexpr int kTw = 128, kTh = 32, kTileWords = 48, kTileRows = 44, kGroups = 40, kSrcWords = 56, kSrcRows = 52;   // orb_pyramid.hip's constants
    out.clear();
    if (l2 < 2 || l2 >= geo.num_levels) return false;
    const LevelGeo &g1 = geo.lv[l2 - 1], &g2 = geo.lv[l2];
    if (!g1.resize_hwin_ok || !g2.resize_hwin_ok) return false;
    for (int axis = 0; axis < 2; ++axis) {
        const ResizeTap* t1 = taps.data() + (axis ? g1.ytab_off : g1.xtab_off);
        const ResizeTap* t2 = taps.data() + (axis ? g2.ytab_off : g2.xtab_off);
        const int size1 = axis ? g1.rows : g1.cols, size2 = axis ? g2.rows : g2.cols, step = axis ? kTh : kTw;
        int expect_b0 = 0;   // the owned ranges must follow each other from 0
        for (int p0 = 0; p0 < size2; p0 += step) {
            const int p1 = std::min(p0 + step, size2) - 1;
            const bool last = p0 + step >= size2;
            PairSpan sp = {};
            const int first = t2[p0].o0;
            const int b0 = axis ? first : (first & ~3), b1 = last ? size1 - 1 : t2[p1].o1;
            if (b1 < t2[p1].o1 || b0 != expect_b0) return false;
            sp.s_lo = (int16_t)(axis ? first : (first & ~15));
            sp.b0 = (int16_t)b0;
            int n, own, hi;   // computed extent (groups / rows), owned extent, last middle index that exists
            if (!axis) {
                n = (b1 - b0) / 4 + 1;
                hi = std::min(b0 + 4 * n - 1, size1 - 1);
                const int next_b0 = last ? b0 + 4 * n : (t2[p0 + step].o0 & ~3);
                own = (next_b0 - b0) / 4;
                if (n > kGroups || (b0 - sp.s_lo) / 4 + n > kTileWords) return false;
                expect_b0 = next_b0;
            } else {
                n = b1 - b0 + 1;
                hi = b1;
                const int next_b0 = last ? b0 + n : t2[p0 + step].o0;
                own = next_b0 - b0;
                if (n > kTileRows) return false;
                expect_b0 = next_b0;
            }
            if (own < 1 || own > n) return false;
            int a_lo = t1[b0].o0, a_hi = t1[hi].o1;
            for (int i = b0; i <= hi; ++i) {   // (the tables are monotone; this does not rely on it)
                a_lo = std::min(a_lo, (int)t1[i].o0);
                a_hi = std::max(a_hi, (int)t1[i].o1);
            }
            if (!axis) a_lo &= ~15;
            const int an = axis ? a_hi - a_lo + 1 : (a_hi - a_lo) / 4 + 1;
            if (an > (axis ? kSrcRows : kSrcWords)) return false;
            sp.n = (int16_t)n;
            sp.own = (int16_t)own;
            sp.a0 = (int16_t)a_lo;
            sp.an = (int16_t)an;
            out.push_back(sp);
        }
        if (expect_b0 < size1) return false;   // the last tile's region ends where the middle level ends
    }
    return true;
}

}   // namespace ovs

using namespace ovs;

struct ovs_orb {
    ovs_orb_params p;
    int device = 0;
    int max_rows = 0, max_cols = 0, max_batch = 0;
    hipStream_t stream = nullptr;
    // A0 tables
    std::vector<float> sf, isf, ls, ils;
    std::vector<int> npl;
    // geometry of the current image size
    int cur_rows = 0, cur_cols = 0;
    FrameGeo geo;
    std::vector<ResizeTap> taps;
    // device memory
    FrameGeo* d_geo = nullptr;
    ResizeTap* d_taps = nullptr;
    size_t taps_cap = 0;
    CellDesc* d_cells = nullptr;
    size_t cells_cap = 0;
    ChainPlanHost chain;             // k_pyramid_chain's tile grid and regions for the current geometry (chain.ok: usable)
    ChainSpan* d_chain = nullptr;    // its spans on the device
    size_t chain_cap = 0;
    // k_resize_pair_u8's regions for the current geometry: pair_off[l2] = offset of the spans of the level pair (l2 - 1, l2) in d_pair, -1 = no plan
    int pair_off[OVS_MAX_LEVELS];
    PairSpan* d_pair = nullptr;
    size_t pair_cap = 0;
    int htab_off[OVS_MAX_LEVELS];    // offset of level l's column-pair records in d_htaps (levels with a pair plan only)
    HTapRec* d_htaps = nullptr;
    size_t htaps_cap = 0;
    int32_t variant = 0;   // FrameGeo::variant
    DevBuffers d{};
    size_t pyr_cap = 0, cand_cap = 0, node_cap = 0, kps_cap = 0;
    // host-API staging: two slots (double buffering). A slot owns a device image (+ mask), a pinned input staging buffer, a pinned
    // output block [counts 16 B | keypoints | descriptors] and, on demand, a pinned copy of the pyramid block. H2D copies run on
    // copy_stream, kernels and the D2H of the results on `stream`: the upload of frame k+1 overlaps the kernels of frame k.
    struct HostSlot {
        uint8_t* d_img = nullptr;
        uint8_t* d_mask = nullptr;
        uint8_t* h_in = nullptr;     // pinned, img_pitch x max_rows
        uint8_t* h_mask = nullptr;   // pinned, lazily
        uint8_t* h_out = nullptr;    // pinned
        uint8_t* h_pyr = nullptr;    // pinned, lazily (ovs_orb_set_host_pyramid)
        hipEvent_t ev_h2d = nullptr, ev_done = nullptr;
        hipEvent_t t[4] = {};        // profiling: h2d begin / h2d end / kernels end / d2h end
        bool pending = false, has_pyr = false, timed = false;
        int rows = 0, cols = 0;
    };
    HostSlot slot[2];
    // ovs_orb_extract_pair (stereo left / right in one call): two device images (+ masks), one output block for both frames, allocated on
    // first use
    struct PairBuf {
        uint8_t *d_img = nullptr, *d_mask = nullptr, *d_out = nullptr, *h_out = nullptr;
        size_t frame_bytes = 0, off_kps = 0, off_desc = 0, out_bytes = 0;
        int cap = 0;
        hipEvent_t ev_h2d = nullptr;
    } pair;
    hipStream_t copy_stream = nullptr;
    unsigned n_submitted = 0, n_collected = 0;
    int last_slot = -1;              // slot of the last COLLECTED frame (host pyramid getter)
    bool host_pyr = false;
    uint8_t* mirror_out = nullptr;     // set by ovs_orb_extract_submit around its run_extract: the pinned block k_describe also writes the results into
    bool cand_count_cleared = false;   // ovs_orb_extract_submit has already enqueued the clearing of the candidate counters for the next run_extract
    int host_mode = 0;               // 0: hipMemcpy2DAsync straight from the caller's (pageable) rows (measured 0.054 ms per 1080p frame);
                                     // 1: banded copy through pinned staging (0.12 ms: the CPU memcpy costs more than the runtime's own staging)
    float host_ms[3] = {0, 0, 0};    // h2d | kernels | d2h of the last collected frame (profiling enabled)
    size_t img_pitch = 0;
    ovs_keypoint* d_out_kps = nullptr;
    uint8_t* d_out_desc = nullptr;
    int32_t* d_out_counts = nullptr;
    int out_cap = 0;
    int out_cap_variant[2] = {0, 0};   // by FrameGeo::variant bit 0; buffers are allocated for [1], the layout follows the current variant
    size_t out_block_bytes = 0, out_off_desc = 0;
    void set_out_layout() {
        out_cap = out_cap_variant[variant & 1];
        out_off_desc = (16 + sizeof(ovs_keypoint) * (size_t)out_cap + 31) & ~(size_t)31;
        out_block_bytes = out_off_desc + (size_t)32 * out_cap;
        d_out_desc = reinterpret_cast<uint8_t*>(d_out_counts) + out_off_desc;
    }
    hipStream_t last_stream = nullptr;   // stream of the last extract (device-batch form: the caller's)
    int32_t last_host_count = -1;        // keypoints the last HOST-form extract returned (they are still in d_out_kps / d_out_desc)
    // last extract (for pyramid / debug getters)
    const uint8_t* last_img0 = nullptr;
    size_t last_stride0 = 0, last_frame_stride0 = 0;
    int last_batch = 0;
    StageProfiler<4> prof;
    // optional sub-batch pipelining of the device-batch path (ovs_orb_set_pipeline)
    static constexpr int kMaxSub = 4;
    int pipeline = 1;
    hipStream_t sub_stream[kMaxSub] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxSub] = {};
    StageProfiler<4> prof_sub[kMaxSub];
    // FAST on level 0 needs no pyramid: it runs on aux_stream BESIDE the seven resize launches (VALU-bound next to latency / bandwidth-bound),
    // the remaining levels follow the pyramid on the main stream (ovs_orb_set_fast_split; default on)
    bool fast_split = true;
    int pyr_chain_max = tuning().pyr_chain;   // frames per launch up to which the pyramid is one k_pyramid_chain launch (ovs_orb_set_pyramid_chain; 0 = never)
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_aux_fork = nullptr, ev_aux_join = nullptr;
    StageProfiler<1> prof_aux;
};

namespace {

void calc_tables(ovs_orb* h) {
    const ovs_orb_params& p = h->p;
    const int L = p.num_levels;
    h->sf.assign(L, 1.0f);
    h->isf.assign(L, 1.0f);
    h->ls.assign(L, 1.0f);
    h->ils.assign(L, 1.0f);
    for (int l = 1; l < L; ++l) h->sf[l] = p.scale_factor * h->sf[l - 1];   // cumulative float product
    for (int l = 0; l < L; ++l) {
        h->isf[l] = 1.0f / h->sf[l];
        h->ls[l] = h->sf[l] * h->sf[l];
        h->ils[l] = 1.0f / h->ls[l];
    }
    h->npl.assign(L, 0);
    double desired = p.max_num_keypts * (1.0 - 1.0 / p.scale_factor) / (1.0 - std::pow(1.0 / p.scale_factor, (double)L));
    int total = 0;
    for (int l = 0; l < L - 1; ++l) {
        h->npl[l] = (int)std::round(desired);
        total += h->npl[l];
        desired *= 1.0 / p.scale_factor;
    }
    h->npl[L - 1] = std::max(p.max_num_keypts - total, 0);
}

// Level sizes, cell grids, root grids, capacities and offsets for an image of rows x cols. Returns false if the geometry
// cannot be processed (level too small for the packed formats, too many root patches).
bool build_geometry(const ovs_orb* h, int rows, int cols, FrameGeo& geo, std::vector<ResizeTap>& taps, size_t& pyr_bytes,
                    size_t& cand_entries, size_t& node_entries, std::vector<CellDesc>* cells = nullptr) {
    const int L = h->p.num_levels;
    std::memset(&geo, 0, sizeof(geo));
    geo.num_levels = L;
    geo.variant = h->variant;
    geo.ini_thr = std::min(std::max(h->p.ini_fast_thr, 0), 255);
    geo.min_thr = std::min(std::max(h->p.min_fast_thr, 0), 255);
    taps.clear();
    pyr_bytes = 0;
    cand_entries = 0;
    node_entries = 0;
    int cell_base = 0, kp_base = 0;
    int prev_rows = rows, prev_cols = cols;
    if (rows > 8191 || cols > 8191) return false;
    for (int l = 0; l < L; ++l) {
        LevelGeo& g = geo.lv[l];
        if (l == 0) {
            g.rows = rows;
            g.cols = cols;
        } else {
            const double scale = h->sf[l];   // compute_image_pyramid: size from the ORIGINAL image
            g.cols = (int)std::round(cols * 1.0 / scale);
            g.rows = (int)std::round(rows * 1.0 / scale);
            if (g.rows < 1 || g.cols < 1) return false;
            g.pitch = (g.cols + 255) & ~255;
            g.plane_off = (int64_t)pyr_bytes;
            pyr_bytes += (size_t)g.pitch * g.rows;
            g.xtab_off = (int64_t)taps.size();
            taps.resize(taps.size() + g.cols);
            build_taps(prev_cols, g.cols, &taps[g.xtab_off]);
            g.ytab_off = (int64_t)taps.size();
            taps.resize(taps.size() + g.rows);
            build_taps(prev_rows, g.rows, &taps[g.ytab_off]);
            g.resize_hwin_ok = resize_windows_ok(&taps[g.xtab_off], g.cols, &taps[g.ytab_off], g.rows) ? 1 : 0;
        }
        prev_rows = g.rows;
        prev_cols = g.cols;
        g.scale = h->sf[l];
        g.kp_size = (float)(unsigned)(kFastPatchSize * h->sf[l]);
        g.max_bx = g.cols - kOrbPatchRadius;
        g.max_by = g.rows - kOrbPatchRadius;
        const int W = g.max_bx - kOrbPatchRadius, H = g.max_by - kOrbPatchRadius;
        // valid cells: 19 + 64*i < max_border - overlap
        g.ncx = W > kCellOverlap ? (W - kCellOverlap + kCellSize - 1) / kCellSize : 0;
        g.ncy = H > kCellOverlap ? (H - kCellOverlap + kCellSize - 1) / kCellSize : 0;
        if (g.ncx == 0 || g.ncy == 0) g.ncx = g.ncy = 0;
        g.inv_ncx = g.ncx ? 1.0f / (float)g.ncx : 0.0f;
        g.ncx_magic = g.ncx > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)g.ncx - 1) / (uint64_t)g.ncx) : 0u;   // ncx == 1: see k_fast_cells
        g.cell_base = cell_base;
        cell_base += g.ncx * g.ncy;
        g.n_keypts = h->npl[l];
        g.gx = g.gy = 1;
        g.dx = g.dy = 1.0;
        if (g.ncx) {
            // initialize_nodes: a row (landscape) or column (portrait) of near-square root patches
            const double ratio = (double)W / H;
            if (ratio > 1) {
                g.gx = (int)std::round(ratio);
                g.gy = 1;
                g.dx = (double)W / g.gx;
                g.dy = H;
            } else {
                g.gx = 1;
                g.gy = (int)std::round(1 / ratio);
                g.dx = W;
                g.dy = (double)H / g.gy;
            }
            if (g.gx * g.gy > 4096) return false;   // (16-bit node ids: max_nodes below is the binding limit)
        }
        const int nc = std::max(g.n_keypts, g.gx * g.gy) + 16;
        g.max_nodes = 4 * nc;
        if (g.max_nodes > 65535) return false;
        // the tree ends with fewer than N + 3 nodes when the pool is split one at a time from N < size + 3 * pool on (rule 6 as fixed);
        // with the factor-1 variant the last all-at-once pass starts from size + pool <= N and may end with up to size + 3 * pool <= 2 N
        g.kp_cap = std::max(((geo.variant & 1) ? 2 * g.n_keypts : g.n_keypts) + 3, 4 * g.gx * g.gy);
        g.kp_base = kp_base;
        kp_base += g.kp_cap;
        g.cand_cap = g.ncx * g.ncy * 1024;   // NMS leaves at most one survivor per 2x2 block of a 64x64 cell
        g.cand_off = (int64_t)cand_entries;
        cand_entries += (size_t)g.cand_cap;
        g.node_off = (int64_t)node_entries;
        node_entries += (size_t)2 * g.max_nodes;
    }
    geo.total_cells = cell_base;
    if (cells) {
        cells->clear();
        cells->reserve((size_t)cell_base);
        for (int l = 0; l < L; ++l) {
            const LevelGeo& g = geo.lv[l];
            for (int ci = 0; ci < g.ncy; ++ci)
                for (int cj = 0; cj < g.ncx; ++cj) {
                    CellDesc c{};
                    const int min_x = kOrbPatchRadius + cj * kCellSize, min_y = kOrbPatchRadius + ci * kCellSize;
                    c.min_x = (uint16_t)min_x;
                    c.min_y = (uint16_t)min_y;
                    c.cw = (uint8_t)(std::min(min_x + kCellSize + kCellOverlap, (int)g.max_bx) - min_x);
                    c.ch = (uint8_t)(std::min(min_y + kCellSize + kCellOverlap, (int)g.max_by) - min_y);
                    c.level = (uint8_t)l;
                    cells->push_back(c);
                }
        }
    }
    geo.total_kp_cap = kp_base;
    for (int l = 0; l < OVS_MAX_LEVELS; ++l) geo.cell_base_tab[l] = l < L ? geo.lv[l].cell_base : INT32_MAX;
    if (tree_lds_bytes_for(geo) > kMaxLdsPerWorkgroup) return false;   // too many keypoints on one level for the quad-tree's LDS arrays
    return true;
}

ovs_status ensure_geometry(ovs_orb* h, int rows, int cols) {
    if (rows == h->cur_rows && cols == h->cur_cols) return OVS_OK;
    if (rows > h->max_rows || cols > h->max_cols) return OVS_ERR_CAPACITY;
    // Build into temporaries and commit only after every check and both uploads succeeded: a refused size (e.g. 60 x 1920 on a handle created for 480 x 1920:
    // its 85 root patches need more node / keypoint capacity than the handle has) must leave the handle exactly as it was, so that the next call with the previous, valid size still finds host
    // geometry, device geometry and cur_rows / cur_cols in agreement.
    size_t pyr_bytes, cand_entries, node_entries;
    FrameGeo geo;
    std::vector<ResizeTap> taps;
    std::vector<CellDesc> cells;
    if (!build_geometry(h, rows, cols, geo, taps, pyr_bytes, cand_entries, node_entries, &cells)) return OVS_ERR_INVALID;
    if (pyr_bytes > h->d.pyr_frame_bytes || cand_entries > h->d.cand_frame_entries || node_entries > h->d.node_frame_entries ||
        taps.size() > h->taps_cap || (size_t)geo.total_kp_cap > h->kps_cap || cells.size() > h->cells_cap)
        return OVS_ERR_CAPACITY;
    // the buffers keep the strides they were allocated with (max geometry); only offsets inside a frame block change
    OVS_HIP_TRY(hipDeviceSynchronize());   // rare: kernels of an earlier geometry may still read d_geo / d_taps
    // from here on the device copy is being replaced: if an upload fails half way, no size is current any more
    h->cur_rows = h->cur_cols = 0;
    OVS_HIP_TRY(hipMemcpy(h->d_geo, &geo, sizeof(FrameGeo), hipMemcpyHostToDevice));
    if (!taps.empty()) OVS_HIP_TRY(hipMemcpy(h->d_taps, taps.data(), taps.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
    if (!cells.empty()) OVS_HIP_TRY(hipMemcpy(h->d_cells, cells.data(), cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice));
    ChainPlanHost chain;
    build_chain_plan(geo, taps, chain);
    if (chain.ok && chain.spans.size() <= h->chain_cap)
        OVS_HIP_TRY(hipMemcpy(h->d_chain, chain.spans.data(), chain.spans.size() * sizeof(ChainSpan), hipMemcpyHostToDevice));
    else
        chain.ok = false;
    h->chain = std::move(chain);
    {
        std::vector<PairSpan> all, one;
            for (int l = 0; l < OVS_MAX_LEVELS; ++l) h->pair_off[l] = -1;
        for (int l2 = 2; l2 < geo.num_levels; ++l2)
            if (build_pair_plan(geo, taps, l2, one) && all.size() + one.size() <= h->pair_cap) {
                h->pair_off[l2] = (int)all.size();
                all.insert(all.end(), one.begin(), one.end());
            }
        if (!all.empty()) OVS_HIP_TRY(hipMemcpy(h->d_pair, all.data(), all.size() * sizeof(PairSpan), hipMemcpyHostToDevice));
        std::vector<HTapRec> recs;
        for (int l = 0; l < OVS_MAX_LEVELS; ++l) h->htab_off[l] = -1;
        for (int l = 1; l < geo.num_levels; ++l)
            if ((l + 1 < geo.num_levels && h->pair_off[l + 1] >= 0) || h->pair_off[l] >= 0) {
                h->htab_off[l] = (int)recs.size();
                build_htaps(taps.data() + geo.lv[l].xtab_off, geo.lv[l].cols, recs);
            }
        if (recs.size() > h->htaps_cap)
            for (int l = 0; l < OVS_MAX_LEVELS; ++l) h->pair_off[l] = -1;   // (cannot happen: the capacity covers every level of the largest image)
        else if (!recs.empty())
            OVS_HIP_TRY(hipMemcpy(h->d_htaps, recs.data(), recs.size() * sizeof(HTapRec), hipMemcpyHostToDevice));
    }
    h->geo = geo;
    h->taps.swap(taps);
    h->cur_rows = rows;
    h->cur_cols = cols;
    return OVS_OK;
}

// pyramid -> FAST -> quad-tree -> describe for frames [f0, f0 + nb) of the batch, on stream s
ovs_status run_chain(ovs_orb* h, StageProfiler<4>& prof, const uint8_t* d_images, int f0, int nb, size_t stride, size_t frame_stride,
                     const uint8_t* d_masks, ovs_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts, int cap, int rows, hipStream_t s) {
    const FrameGeo& geo = h->geo;
    const int L = geo.num_levels;
    DevBuffers d = h->d;   // views of this sub-batch
    d.pyr += (size_t)f0 * d.pyr_frame_bytes;
    d.cand += (size_t)f0 * d.cand_frame_entries;
    d.cand_count += (size_t)f0 * L;
    d.nodes += (size_t)f0 * d.node_frame_entries * 4;
    d.lvl_kps += (size_t)f0 * geo.total_kp_cap;
    d.lvl_count += (size_t)f0 * L;
    const uint8_t* img = d_images + (size_t)f0 * frame_stride;
    const uint8_t* msk = d_masks ? d_masks + (size_t)f0 * frame_stride : nullptr;
    OVS_HIP_TRY(prof.begin(s));
    const bool split = h->fast_split && L > 1 && &prof == &h->prof && geo.lv[1].cell_base > 0;
    if (split) {
        OVS_HIP_TRY(hipEventRecord(h->ev_aux_fork, s));
        OVS_HIP_TRY(hipStreamWaitEvent(h->aux_stream, h->ev_aux_fork, 0));
        h->prof_aux.enabled = prof.enabled;
        OVS_HIP_TRY(h->prof_aux.begin(h->aux_stream));
        OVS_HIP_TRY(launch_fast(geo, d, img, stride, frame_stride, msk, rows, nb, h->aux_stream, 0, geo.lv[1].cell_base));
        OVS_HIP_TRY(h->prof_aux.mark(1, h->aux_stream));
        // level 0's quad-tree (the longest pole: ~20 k candidates per frame, latency-bound) follows on the same stream, under the FAST of
        // the other levels
        OVS_HIP_TRY(launch_tree(geo, d, nb, h->aux_stream, 0, 1));
        OVS_HIP_TRY(hipEventRecord(h->ev_aux_join, h->aux_stream));
    }
    // A1: each level from the previous one -- for a tracker's single frame (or a stereo pair) all levels in ONE launch (k_pyramid_chain: the
    // seven dependent launches cost 40 us of a 0.24 ms extract), for batches level by level (the throughput form)
    const bool chained = h->chain.ok && L > 1 && nb <= h->pyr_chain_max && !((uintptr_t)img & 3) && !(frame_stride & 3) && !(stride & 3);
    if (chained)
        OVS_HIP_TRY(launch_pyramid_chain(img, frame_stride, (int)stride, d.pyr, d.pyr_frame_bytes, d.geo, h->d_taps, h->d_chain, h->chain.TX, h->chain.TY,
                                         h->chain.bufA, h->chain.bufB, h->chain.tap_cap, nb, s));
    for (int l = 1; l < L && !chained; ++l) {
        const LevelGeo& g = geo.lv[l];
        const LevelGeo& gp = geo.lv[l - 1];
        const uint8_t* src = (l == 1) ? img : d.pyr + gp.plane_off;
        const size_t src_fs = (l == 1) ? frame_stride : d.pyr_frame_bytes;
        const int src_pitch = (l == 1) ? (int)stride : gp.pitch;
        // batches: levels l and l + 1 in one launch where the geometry has a plan for the pair (the middle level is written once, never re-read)
        if (tuning().pyr_pair && l + 1 < L && h->pair_off[l + 1] >= 0 && resize_pair_launchable(src, src_fs, src_pitch, geo.lv[l + 1].rows, geo.lv[l + 1].cols, nb)) {
            const LevelGeo& g2 = geo.lv[l + 1];
            OVS_HIP_TRY(launch_resize_pair(src, src_fs, src_pitch, d.pyr + g.plane_off, g.pitch, g.rows, g.cols, d.pyr + g2.plane_off, g2.pitch, g2.rows, g2.cols,
                                           d.pyr_frame_bytes, h->d_taps + g.ytab_off, h->d_taps + g2.ytab_off, h->d_htaps + h->htab_off[l], h->d_htaps + h->htab_off[l + 1],
                                           h->d_pair + h->pair_off[l + 1], nb, s));
            ++l;
            continue;
        }
        OVS_HIP_TRY(launch_resize(src, src_fs, src_pitch, gp.rows, gp.cols, d.pyr + g.plane_off, d.pyr_frame_bytes, g.pitch, g.rows, g.cols,
                                  h->d_taps + g.xtab_off, h->d_taps + g.ytab_off, nb, s, g.resize_hwin_ok));
    }
    OVS_HIP_TRY(prof.mark(1, s));
    if (split) {
        OVS_HIP_TRY(launch_fast(geo, d, img, stride, frame_stride, msk, rows, nb, s, geo.lv[1].cell_base, -1));
    } else {
        OVS_HIP_TRY(launch_fast(geo, d, img, stride, frame_stride, msk, rows, nb, s));
    }
    OVS_HIP_TRY(prof.mark(2, s));
    if (split) {
        OVS_HIP_TRY(launch_tree(geo, d, nb, s, 1, -1));
        OVS_HIP_TRY(hipStreamWaitEvent(s, h->ev_aux_join, 0));   // describe needs every level's keypoints
    } else {
        OVS_HIP_TRY(launch_tree(geo, d, nb, s));
    }
    OVS_HIP_TRY(prof.mark(3, s));
    // a one-frame host call: the results also go straight into its pinned block (h->mirror_out: [counts 16 B | keypoints | descriptors])
    uint8_t* const mo = (h->mirror_out && nb == 1 && f0 == 0 && d_kps == h->d_out_kps && cap == h->out_cap) ? h->mirror_out : nullptr;
    OVS_HIP_TRY(launch_describe(geo, d, img, stride, frame_stride, d_kps + (size_t)f0 * cap, d_desc + (size_t)f0 * cap * 32, d_counts + f0, cap, nb, s,
                                mo ? reinterpret_cast<ovs_keypoint*>(mo + 16) : nullptr, mo ? mo + h->out_off_desc : nullptr,
                                mo ? reinterpret_cast<int32_t*>(mo) : nullptr));
    OVS_HIP_TRY(prof.mark(4, s));
    return OVS_OK;
}

ovs_status run_extract(ovs_orb* h, const uint8_t* d_images, int batch, int rows, int cols, size_t stride, size_t frame_stride,
                       const uint8_t* d_masks, ovs_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts, int cap, hipStream_t s) {
    ovs_status st = ensure_geometry(h, rows, cols);
    if (st != OVS_OK) return st;
    const int L = h->geo.num_levels;
    if (!h->cand_count_cleared) OVS_HIP_TRY(hipMemsetAsync(h->d.cand_count, 0, sizeof(uint32_t) * (size_t)batch * L, s));
    h->cand_count_cleared = false;
    const int nsub = std::min(h->pipeline, batch);
    if (nsub <= 1) {
        st = run_chain(h, h->prof, d_images, 0, batch, stride, frame_stride, d_masks, d_kps, d_desc, d_counts, cap, rows, s);
        if (st != OVS_OK) return st;
    } else {
        // fork: the sub-batches run their chains on the handle's own streams, so the latency-bound stages of one (pyramid, quad-tree,
        // describe) overlap the VALU-bound FAST pass of another; join: the caller's stream waits for all of them
        OVS_HIP_TRY(hipEventRecord(h->ev_fork, s));
        const int per = (batch + nsub - 1) / nsub;
        for (int k = 0; k < nsub; ++k) {
            const int f0 = k * per, nb = std::min(per, batch - f0);
            if (nb <= 0) break;
            OVS_HIP_TRY(hipStreamWaitEvent(h->sub_stream[k], h->ev_fork, 0));
            h->prof_sub[k].enabled = h->prof.enabled;
            st = run_chain(h, h->prof_sub[k], d_images, f0, nb, stride, frame_stride, d_masks, d_kps, d_desc, d_counts, cap, rows, h->sub_stream[k]);
            if (st != OVS_OK) return st;
            OVS_HIP_TRY(hipEventRecord(h->ev_join[k], h->sub_stream[k]));
            OVS_HIP_TRY(hipStreamWaitEvent(s, h->ev_join[k], 0));
        }
    }
    h->last_img0 = d_images;
    h->last_stride0 = stride;
    h->last_frame_stride0 = frame_stride;
    h->last_batch = batch;
    h->last_stream = s;
    return OVS_OK;
}

}   // namespace

namespace ovs {
bool orb_pyramid_view(const ovs_orb* h, int frame, PyrView* out) {
    if (!h || !out || !h->last_img0 || frame < 0 || frame >= h->last_batch) return false;
    const int L = h->geo.num_levels;
    out->num_levels = L;
    for (int l = 0; l < L; ++l) {
        const LevelGeo& g = h->geo.lv[l];
        if (l == 0) {
            out->base[0] = h->last_img0 + (size_t)frame * h->last_frame_stride0;
            out->pitch[0] = (int32_t)h->last_stride0;
        } else {
            out->base[l] = h->d.pyr + (size_t)frame * h->d.pyr_frame_bytes + g.plane_off;
            out->pitch[l] = g.pitch;
        }
        out->rows[l] = g.rows;
        out->cols[l] = g.cols;
        out->scale[l] = h->sf[l];
        out->inv_scale[l] = h->isf[l];
    }
    return true;
}
int orb_device(const ovs_orb* h) { return h ? h->device : -1; }
hipStream_t orb_last_stream(const ovs_orb* h) { return h ? h->last_stream : nullptr; }

// If (kps, desc, n) are byte-for-byte what the handle's last HOST-form extract returned -- the pinned block the results were downloaded into
// is still there to compare with (120 KB: a few microseconds) -- the same data is also still in the device output block: hand that out.
bool orb_host_outputs_equal(const ovs_orb* h, const ovs_keypoint* kps, const uint8_t* desc, int32_t n, const ovs_keypoint** d_kps,
                            const uint8_t** d_desc) {
    if (!h || h->last_host_count < 0 || h->last_host_count != n || h->n_submitted != h->n_collected || h->last_slot < 0) return false;
    const ovs_orb::HostSlot& sl = h->slot[h->last_slot];
    if (!sl.h_out) return false;
    if (n > 0 && (std::memcmp(kps, sl.h_out + 16, sizeof(ovs_keypoint) * (size_t)n) != 0 || std::memcmp(desc, sl.h_out + h->out_off_desc, (size_t)32 * n) != 0))
        return false;
    *d_kps = h->d_out_kps;
    *d_desc = h->d_out_desc;
    return true;
}
}   // namespace ovs

extern "C" {

const char* ovs_last_error(void) { return g_last_error.c_str(); }

int32_t ovs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

ovs_status ovs_device_arch(int32_t dev, char* buf, size_t buflen) {
    if (!buf || buflen == 0) return OVS_ERR_INVALID;
    hipDeviceProp_t prop;
    OVS_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    std::snprintf(buf, buflen, "%s", prop.gcnArchName);
    return OVS_OK;
}

ovs_status ovs_orb_create(const ovs_orb_params* params, int32_t max_rows, int32_t max_cols, int32_t max_batch, int32_t device,
                          ovs_orb** out) {
    if (!params || !out || params->num_levels < 1 || params->num_levels > OVS_MAX_LEVELS || !(params->scale_factor > 1.0f) ||
        params->max_num_keypts < 1 || max_rows < 1 || max_cols < 1 || max_batch < 1)
        return OVS_ERR_INVALID;
    *out = nullptr;
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    ovs_orb* h = new (std::nothrow) ovs_orb();
    if (!h) return OVS_ERR_INVALID;
    h->p = *params;
    h->device = device;
    h->max_rows = max_rows;
    h->max_cols = max_cols;
    h->max_batch = max_batch;
    calc_tables(h);
    FrameGeo geo;
    std::vector<ResizeTap> taps;
    std::vector<CellDesc> cells_max;
    size_t pyr_bytes, cand_entries, node_entries;
    if (!build_geometry(h, max_rows, max_cols, geo, taps, pyr_bytes, cand_entries, node_entries, &cells_max)) {
        delete h;
        return OVS_ERR_INVALID;
    }
#define CREATE_TRY(expr)                       \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) {                \
            ovs::set_last_error(#expr, _e);    \
            ovs_orb_destroy(h);                \
            return OVS_ERR_HIP;                \
        }                                      \
    } while (0)
    CREATE_TRY(hipSetDevice(device));
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    const int L = params->num_levels;
    const size_t B = (size_t)max_batch;
    h->d.pyr_frame_bytes = (pyr_bytes + 255) & ~(size_t)255;
    h->d.cand_frame_entries = cand_entries;
    // The per-level node and keypoint capacities depend on the root grid (round(W / H) x 1 patches, at most 64), i.e. on the ASPECT
    // RATIO of the image, not on its size: a small elongated image can need more of both than the largest image the handle was created
    // for (found by tools/fuzz_parity.py: 1664 x 257 on a 2000 x 1300 handle). Size them for the worst grid.
    {
        // (round 6: more than 64 root patches run when the handle was CREATED for such a shape -- the largest image's own root grids count
        // here; a 64:1 strip on a handle created for 4:3 images is refused with OVS_ERR_CAPACITY, not OVS_ERR_INVALID)
        size_t node_worst = 0, kp_worst[2] = {0, 0};
        for (int l = 0; l < L; ++l) {
            const int roots = std::max(64, geo.lv[l].gx * geo.lv[l].gy);
            node_worst += (size_t)2 * 4 * (std::max(geo.lv[l].n_keypts, roots) + 16);
            kp_worst[0] += (size_t)std::max(geo.lv[l].n_keypts + 3, 4 * roots);
            kp_worst[1] += (size_t)std::max(2 * geo.lv[l].n_keypts + 3, 4 * roots);   // ovs_orb_set_variant(TREE_SWITCH_FACTOR, 1)
        }
        node_entries = std::max(node_entries, node_worst);
        h->out_cap_variant[0] = (int)kp_worst[0];
        h->out_cap_variant[1] = (int)kp_worst[1];
    }
    h->d.node_frame_entries = node_entries;
    h->taps_cap = taps.size() + 64;
    h->kps_cap = (size_t)h->out_cap_variant[1];
    CREATE_TRY(hipMalloc(&h->d_geo, sizeof(FrameGeo)));
    CREATE_TRY(hipMalloc(&h->d_taps, h->taps_cap * sizeof(ResizeTap)));
    h->cells_cap = cells_max.size() + 64;   // (the cell count is monotone in rows and cols: the largest image has the most cells)
    CREATE_TRY(hipMalloc(&h->d_cells, h->cells_cap * sizeof(CellDesc)));
    h->chain_cap = (size_t)L * ((size_t)(max_cols + 127) / 128 + (size_t)(max_rows + 95) / 96 + 2);   // (the tile grid is monotone in rows and cols)
    CREATE_TRY(hipMalloc(&h->d_chain, h->chain_cap * sizeof(ChainSpan)));
    h->pair_cap = (size_t)L * ((size_t)(max_cols + 127) / 128 + (size_t)(max_rows + 31) / 32 + 2);
    CREATE_TRY(hipMalloc(&h->d_pair, h->pair_cap * sizeof(PairSpan)));
    for (int l = 0; l < OVS_MAX_LEVELS; ++l) h->pair_off[l] = h->htab_off[l] = -1;
    h->htaps_cap = h->taps_cap;   // one record per column pair of every level: fewer than there are taps
    CREATE_TRY(hipMalloc(&h->d_htaps, h->htaps_cap * sizeof(HTapRec)));
    CREATE_TRY(hipMalloc(&h->d.pyr, std::max<size_t>(h->d.pyr_frame_bytes * B, 256)));
    CREATE_TRY(hipMalloc(&h->d.cand, std::max<size_t>(cand_entries * B * sizeof(uint64_t), 256)));
    CREATE_TRY(hipMalloc(&h->d.cand_count, sizeof(uint32_t) * B * L));
    CREATE_TRY(hipMalloc(&h->d.nodes, std::max<size_t>(node_entries * B * 16, 256)));
    CREATE_TRY(hipMalloc(&h->d.lvl_kps, std::max<size_t>(h->kps_cap * B * sizeof(uint64_t), 256)));
    CREATE_TRY(hipMalloc(&h->d.lvl_count, sizeof(uint32_t) * B * L));
    CREATE_TRY(hipMemset(h->d.lvl_count, 0, sizeof(uint32_t) * B * L));
    h->d.geo = h->d_geo;
    h->d.taps = h->d_taps;
    h->d.cells = h->d_cells;
    // host-API staging: one frame
    h->img_pitch = ((size_t)max_cols + 255) & ~(size_t)255;
    // one contiguous device output block [counts | keypoints | descriptors] so that ONE D2H brings a frame's results back; allocated
    // for the larger of the two capacities, laid out (and copied) for the current one
    const size_t out_block_alloc = ((16 + sizeof(ovs_keypoint) * (size_t)h->out_cap_variant[1] + 31) & ~(size_t)31) + (size_t)32 * h->out_cap_variant[1];
    {
        uint8_t* blk = nullptr;
        CREATE_TRY(hipMalloc(&blk, out_block_alloc));
        h->d_out_counts = reinterpret_cast<int32_t*>(blk);
        h->d_out_kps = reinterpret_cast<ovs_keypoint*>(blk + 16);
    }
    h->set_out_layout();
    CREATE_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_aux_fork, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_aux_join, hipEventDisableTiming));
    for (auto& sl : h->slot) {
        CREATE_TRY(hipMalloc(&sl.d_img, h->img_pitch * max_rows));
        CREATE_TRY(hipHostMalloc(&sl.h_in, h->img_pitch * max_rows, hipHostMallocDefault));
        CREATE_TRY(hipHostMalloc(&sl.h_out, out_block_alloc, hipHostMallocDefault));
        CREATE_TRY(hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming));
        CREATE_TRY(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
    }
#undef CREATE_TRY
    *out = h;
    return OVS_OK;
}

int32_t ovs_orb_device(const ovs_orb* h) { return ovs::orb_device(h); }

ovs_status ovs_orb_destroy(ovs_orb* h) {
    if (!h) return OVS_OK;
    if (h->stream) hipStreamSynchronize(h->stream);
    hipFree(h->d_geo);
    hipFree(h->d_taps);
    hipFree(h->d_cells);
    hipFree(h->d_chain);
    hipFree(h->d_pair);
    hipFree(h->d_htaps);
    hipFree(h->d.pyr);
    hipFree(h->d.cand);
    hipFree(h->d.cand_count);
    hipFree(h->d.nodes);
    hipFree(h->d.lvl_kps);
    hipFree(h->d.lvl_count);
    if (h->copy_stream) hipStreamSynchronize(h->copy_stream);
    hipFree(h->pair.d_img);
    hipFree(h->pair.d_mask);
    hipFree(h->pair.d_out);
    if (h->pair.h_out) hipHostFree(h->pair.h_out);
    if (h->pair.ev_h2d) hipEventDestroy(h->pair.ev_h2d);
    for (auto& sl : h->slot) {
        hipFree(sl.d_img);
        hipFree(sl.d_mask);
        if (sl.h_in) hipHostFree(sl.h_in);
        if (sl.h_mask) hipHostFree(sl.h_mask);
        if (sl.h_out) hipHostFree(sl.h_out);
        if (sl.h_pyr) hipHostFree(sl.h_pyr);
        if (sl.ev_h2d) hipEventDestroy(sl.ev_h2d);
        if (sl.ev_done) hipEventDestroy(sl.ev_done);
        for (auto& e : sl.t)
            if (e) hipEventDestroy(e);
    }
    hipFree(h->d_out_counts);   // base of the [counts | keypoints | descriptors] block
    if (h->aux_stream) {
        hipStreamSynchronize(h->aux_stream);
        hipStreamDestroy(h->aux_stream);
    }
    if (h->ev_aux_fork) hipEventDestroy(h->ev_aux_fork);
    if (h->ev_aux_join) hipEventDestroy(h->ev_aux_join);
    h->prof_aux.destroy();
    if (h->copy_stream) hipStreamDestroy(h->copy_stream);
    h->prof.destroy();
    for (int k = 0; k < ovs_orb::kMaxSub; ++k) {
        h->prof_sub[k].destroy();
        if (h->sub_stream[k]) {
            hipStreamSynchronize(h->sub_stream[k]);
            hipStreamDestroy(h->sub_stream[k]);
        }
        if (h->ev_join[k]) hipEventDestroy(h->ev_join[k]);
    }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return OVS_OK;
}

ovs_status ovs_orb_tables(const ovs_orb* h, float* sf, float* isf, float* ls, float* ils, int32_t* npl) {
    if (!h) return OVS_ERR_INVALID;
    for (int l = 0; l < h->p.num_levels; ++l) {
        if (sf) sf[l] = h->sf[l];
        if (isf) isf[l] = h->isf[l];
        if (ls) ls[l] = h->ls[l];
        if (ils) ils[l] = h->ils[l];
        if (npl) npl[l] = h->npl[l];
    }
    return OVS_OK;
}

int32_t ovs_orb_max_keypoints(const ovs_orb* h) { return h ? h->out_cap : 0; }

ovs_status ovs_orb_profile_enable(ovs_orb* h, int32_t enable) {
    if (!h) return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(h->device));
    h->prof.enabled = enable != 0;
    if (enable) OVS_HIP_TRY(h->prof.ensure());
    for (int k = 0; k < h->pipeline && h->pipeline > 1; ++k) {
        h->prof_sub[k].enabled = h->prof.enabled;
        if (enable) OVS_HIP_TRY(h->prof_sub[k].ensure());
    }
    return OVS_OK;
}

ovs_status ovs_orb_profile_read(ovs_orb* h, float* stage_ms, int32_t* ncalls) {
    if (!h || !stage_ms || !ncalls) return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(h->device));
    OVS_HIP_TRY(h->prof.read(stage_ms, ncalls));
    // pipelined calls: every sub-batch recorded its own chain on its own stream; a stage's time is the sum of its launches' own
    // durations (they overlap launches of other stages in wall time)
    for (int k = 0; k < ovs_orb::kMaxSub; ++k) {
        if (!h->prof_sub[k].created) continue;
        float ms[4];
        int32_t nc = 0;
        OVS_HIP_TRY(h->prof_sub[k].read(ms, &nc));
        for (int j = 0; j < 4; ++j) stage_ms[j] += ms[j];
        if (k == 0) *ncalls += nc;
    }
    return OVS_OK;
}

ovs_status ovs_orb_set_variant(ovs_orb* h, int32_t which, int32_t value) {
    if (!h) return OVS_ERR_INVALID;
    int32_t v = h->variant;
    if (which == OVS_VARIANT_TREE_SWITCH_FACTOR && (value == 3 || value == 1)) v = (v & ~1) | (value == 1 ? 1 : 0);
    else if (which == OVS_VARIANT_TREE_TIE_ORDER && (value == 0 || value == 1)) v = (v & ~2) | (value ? 2 : 0);
    else if (which == OVS_VARIANT_BLUR_TAPS && (value == 0 || value == 1)) v = (v & ~4) | (value ? 4 : 0);
    else if (which == OVS_VARIANT_TRIG && (value == 0 || value == 1)) v = (v & ~8) | (value ? 8 : 0);
    else return OVS_ERR_INVALID;
    if (v != h->variant) {
        OVS_HIP_TRY(hipSetDevice(h->device));
        OVS_HIP_TRY(hipDeviceSynchronize());   // the output block's layout follows the variant's capacity: nothing may be in flight
        h->variant = v;
        h->cur_rows = h->cur_cols = 0;   // the device copy of the geometry carries the flags: rebuilt by the next extract
        h->last_host_count = -1;
        h->set_out_layout();             // ovs_orb_max_keypoints changes with TREE_SWITCH_FACTOR
    }
    return OVS_OK;
}

ovs_status ovs_debug_inject_hip_failures(int32_t skip_calls, int32_t n_calls) {
    ovs::g_injected_hip_failures.store(0, std::memory_order_relaxed);
    ovs::g_injected_hip_skip.store(skip_calls > 0 ? skip_calls : 0, std::memory_order_relaxed);
    ovs::g_injected_hip_failures.store(n_calls > 0 ? n_calls : 0, std::memory_order_relaxed);
    return OVS_OK;
}

ovs_status ovs_orb_set_fast_split(ovs_orb* h, int32_t enable) {
    if (!h) return OVS_ERR_INVALID;
    h->fast_split = enable != 0;
    return OVS_OK;
}

ovs_status ovs_orb_set_pyramid_chain(ovs_orb* h, int32_t max_frames) {
    if (!h || max_frames < 0) return OVS_ERR_INVALID;
    h->pyr_chain_max = max_frames;
    return OVS_OK;
}

ovs_status ovs_orb_profile_read_aux(ovs_orb* h, float* fast_level0_ms, int32_t* ncalls) {
    if (!h || !fast_level0_ms || !ncalls) return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(h->device));
    *fast_level0_ms = 0;
    *ncalls = 0;
    if (!h->prof_aux.created) return OVS_OK;
    OVS_HIP_TRY(h->prof_aux.read(fast_level0_ms, ncalls));
    return OVS_OK;
}

ovs_status ovs_orb_set_pipeline(ovs_orb* h, int32_t n_sub) {
    if (!h || n_sub < 1 || n_sub > ovs_orb::kMaxSub) return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(h->device));
    if (n_sub > 1) {
        if (!h->ev_fork) OVS_HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (int k = 0; k < n_sub; ++k) {
            if (!h->sub_stream[k]) OVS_HIP_TRY(hipStreamCreateWithFlags(&h->sub_stream[k], hipStreamNonBlocking));
            if (!h->ev_join[k]) OVS_HIP_TRY(hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming));
            if (h->prof.enabled) OVS_HIP_TRY(h->prof_sub[k].ensure());
        }
    }
    h->pipeline = n_sub;
    return OVS_OK;
}

ovs_status ovs_orb_extract_batch_dev(ovs_orb* h, const uint8_t* d_images, int32_t batch, int32_t rows, int32_t cols, size_t stride,
                                     size_t frame_stride, const uint8_t* d_masks, ovs_keypoint* d_kps, uint8_t* d_desc,
                                     int32_t* d_counts, int32_t cap, void* stream) {
    if (!h || !d_images || !d_kps || !d_desc || !d_counts || batch < 1 || rows < 1 || cols < 1 || cap < 1) return OVS_ERR_INVALID;
    if (batch > h->max_batch) return OVS_ERR_CAPACITY;
    if (((uintptr_t)d_images & 3) || (stride & 3) || (frame_stride & 3) || stride < (size_t)cols) return OVS_ERR_ALIGN;
    if (d_masks && ((uintptr_t)d_masks & 3)) return OVS_ERR_ALIGN;
    OVS_HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;   // verbatim: NULL is HIP's default stream (torch's default stream handle is 0)
    return run_extract(h, d_images, batch, rows, cols, stride, frame_stride, d_masks, d_kps, d_desc, d_counts, cap, s);
}

// ---- host-pointer path: submit / collect over two slots; ovs_orb_extract = submit + collect ----------------------------------------
namespace {

// rows of `cols` bytes from the caller's (pageable) buffer to the device image of a slot, on copy_stream
ovs_status upload_plane(ovs_orb* h, const uint8_t* src, size_t stride, int rows, int cols, uint8_t* staging, uint8_t* d_dst) {
    hipStream_t cs = h->copy_stream;
    const size_t pitch = h->img_pitch;
    if (h->host_mode == 0 || !staging) {
        OVS_HIP_TRY(hipMemcpy2DAsync(d_dst, pitch, src, stride, (size_t)cols, (size_t)rows, hipMemcpyHostToDevice, cs));
        return OVS_OK;
    }
    // banded: the CPU copies band k+1 into pinned memory while the DMA engine moves band k (a pageable hipMemcpy does the same inside
    // the runtime, but through its own small staging chunks and with a blocking wait at the end)
    const int band = std::max(32, (rows + 7) / 8);
    for (int r0 = 0; r0 < rows; r0 += band) {
        const int nr = std::min(band, rows - r0);
        uint8_t* st = staging + (size_t)r0 * pitch;
        if (stride == (size_t)cols && pitch == (size_t)cols) {
            std::memcpy(st, src + (size_t)r0 * stride, (size_t)nr * cols);
        } else {
            for (int r = 0; r < nr; ++r) std::memcpy(st + (size_t)r * pitch, src + (size_t)(r0 + r) * stride, (size_t)cols);
        }
        OVS_HIP_TRY(hipMemcpyAsync(d_dst + (size_t)r0 * pitch, st, (size_t)nr * pitch, hipMemcpyHostToDevice, cs));
    }
    return OVS_OK;
}

}   // namespace

ovs_status ovs_orb_extract_submit(ovs_orb* h, const uint8_t* image, int32_t rows, int32_t cols, size_t stride, const uint8_t* mask,
                                  size_t mask_stride) {
    if (!h || !image || rows <= 0 || cols <= 0 || stride < (size_t)cols || (mask && mask_stride < (size_t)cols)) return OVS_ERR_INVALID;
    if (rows > h->max_rows || cols > h->max_cols) return OVS_ERR_CAPACITY;
    if (h->n_submitted - h->n_collected >= 2) return OVS_ERR_CAPACITY;   // both slots in flight: collect first
    OVS_HIP_TRY(hipSetDevice(h->device));
    ovs_orb::HostSlot& sl = h->slot[h->n_submitted & 1];
    hipStream_t s = h->stream, cs = h->copy_stream;
    // a geometry change synchronises the device (ensure_geometry): do it before anything of this frame is enqueued
    ovs_status st = ensure_geometry(h, rows, cols);
    if (st != OVS_OK) return st;
    const bool timed = h->prof.enabled;
    if (timed)
        for (auto& e : sl.t)
            if (!e) OVS_HIP_TRY(hipEventCreate(&e));
    // the candidate counters are cleared while the image is still on its way (a 4 us fill kernel that used to run between the upload and the pyramid)
    OVS_HIP_TRY(hipMemsetAsync(h->d.cand_count, 0, sizeof(uint32_t) * (size_t)h->geo.num_levels, s));
    h->cand_count_cleared = true;
    struct ClearedGuard {   // the flag lives inside this call: whatever path leaves it, the next run_extract clears its own counters
        ovs_orb* h;
        ~ClearedGuard() {
            h->cand_count_cleared = false;
            h->mirror_out = nullptr;
        }
    } cleared_guard{h};
    // OVS_ORB_ZERO_COPY_OUT=0: results through a D2H copy command after the kernels (rounds 1-5)
    static const bool zero_copy_out = [] {
        const char* e = std::getenv("OVS_ORB_ZERO_COPY_OUT");
        return !(e && e[0] == '0');
    }();
    h->mirror_out = zero_copy_out ? sl.h_out : nullptr;
    if (timed) OVS_HIP_TRY(hipEventRecord(sl.t[0], cs));
    st = upload_plane(h, image, stride, rows, cols, sl.h_in, sl.d_img);
    if (st != OVS_OK) return st;
    if (mask) {
        if (!sl.d_mask) OVS_HIP_TRY(hipMalloc(&sl.d_mask, h->img_pitch * h->max_rows));
        if (!sl.h_mask) OVS_HIP_TRY(hipHostMalloc(&sl.h_mask, h->img_pitch * h->max_rows, hipHostMallocDefault));
        st = upload_plane(h, mask, mask_stride, rows, cols, sl.h_mask, sl.d_mask);
        if (st != OVS_OK) return st;
    }
    if (timed) OVS_HIP_TRY(hipEventRecord(sl.t[1], cs));
    OVS_HIP_TRY(hipEventRecord(sl.ev_h2d, cs));
    OVS_HIP_TRY(hipStreamWaitEvent(s, sl.ev_h2d, 0));
    st = run_extract(h, sl.d_img, 1, rows, cols, h->img_pitch, h->img_pitch * (size_t)rows, mask ? sl.d_mask : nullptr, h->d_out_kps,
                     h->d_out_desc, h->d_out_counts, h->out_cap, s);
    if (st != OVS_OK) return st;
    if (timed) OVS_HIP_TRY(hipEventRecord(sl.t[2], s));
    // ONE D2H for counts + keypoints + descriptors (120 KB at 2000 features: cheaper than a count round trip followed by two copies)
    if (!h->mirror_out) OVS_HIP_TRY(hipMemcpyAsync(sl.h_out, h->d_out_counts, h->out_block_bytes, hipMemcpyDeviceToHost, s));
    sl.has_pyr = false;
    if (h->host_pyr && h->p.num_levels > 1) {
        if (!sl.h_pyr) OVS_HIP_TRY(hipHostMalloc(&sl.h_pyr, h->d.pyr_frame_bytes, hipHostMallocDefault));
        const LevelGeo& gl = h->geo.lv[h->p.num_levels - 1];
        const size_t used = (size_t)gl.plane_off + (size_t)gl.pitch * gl.rows;   // levels 1 .. L-1 are contiguous in the frame block
        OVS_HIP_TRY(hipMemcpyAsync(sl.h_pyr, h->d.pyr, used, hipMemcpyDeviceToHost, s));
        sl.has_pyr = true;
    }
    if (timed) OVS_HIP_TRY(hipEventRecord(sl.t[3], s));
    OVS_HIP_TRY(hipEventRecord(sl.ev_done, s));
    sl.pending = true;
    sl.timed = timed;
    sl.rows = rows;
    sl.cols = cols;
    ++h->n_submitted;
    return OVS_OK;
}

ovs_status ovs_orb_extract_collect(ovs_orb* h, ovs_keypoint* kps, uint8_t* desc, int32_t cap, int32_t* n_out) {
    if (!h || !n_out || cap < 0 || (cap > 0 && (!kps || !desc))) return OVS_ERR_INVALID;
    *n_out = 0;
    if (h->n_collected == h->n_submitted) return OVS_ERR_INVALID;   // nothing in flight
    OVS_HIP_TRY(hipSetDevice(h->device));
    const int si = (int)(h->n_collected & 1);
    ovs_orb::HostSlot& sl = h->slot[si];
    OVS_HIP_TRY(hipEventSynchronize(sl.ev_done));   // the only wait of a frame
    sl.pending = false;
    ++h->n_collected;
    h->last_slot = si;
    if (sl.timed) {
        OVS_HIP_TRY(hipEventElapsedTime(&h->host_ms[0], sl.t[0], sl.t[1]));
        OVS_HIP_TRY(hipEventElapsedTime(&h->host_ms[1], sl.t[1], sl.t[2]));
        OVS_HIP_TRY(hipEventElapsedTime(&h->host_ms[2], sl.t[2], sl.t[3]));
    }
    const int32_t n = *reinterpret_cast<const int32_t*>(sl.h_out);
    const int32_t m = std::min(n, cap);
    if (m > 0) {
        std::memcpy(kps, sl.h_out + 16, sizeof(ovs_keypoint) * (size_t)m);
        std::memcpy(desc, sl.h_out + h->out_off_desc, (size_t)32 * m);
    }
    *n_out = m;
    h->last_host_count = m;
    return n > cap ? OVS_ERR_CAPACITY : OVS_OK;
}

ovs_status ovs_orb_extract(ovs_orb* h, const uint8_t* image, int32_t rows, int32_t cols, size_t stride, const uint8_t* mask,
                           size_t mask_stride, ovs_keypoint* kps, uint8_t* desc, int32_t cap, int32_t* n_out) {
    if (!h || !n_out) return OVS_ERR_INVALID;
    *n_out = 0;
    if (!image || rows <= 0 || cols <= 0) return OVS_OK;   // upstream: empty image -> early return
    if (cap < 0 || (cap > 0 && (!kps || !desc))) return OVS_ERR_INVALID;
    if (h->n_submitted != h->n_collected) return OVS_ERR_INVALID;   // frames submitted asynchronously must be collected first
    const ovs_status st = ovs_orb_extract_submit(h, image, rows, cols, stride, mask, mask_stride);
    if (st != OVS_OK) return st;
    return ovs_orb_extract_collect(h, kps, desc, cap, n_out);
}

ovs_status ovs_orb_extract_pair(ovs_orb* h, const uint8_t* left, const uint8_t* right, int32_t rows, int32_t cols, size_t stride,
                                const uint8_t* mask_left, const uint8_t* mask_right, size_t mask_stride, ovs_keypoint* kps_left,
                                uint8_t* desc_left, int32_t* n_left, ovs_keypoint* kps_right, uint8_t* desc_right, int32_t* n_right, int32_t cap) {
    if (!h || !n_left || !n_right) return OVS_ERR_INVALID;
    *n_left = *n_right = 0;
    if (!left || !right || rows <= 0 || cols <= 0) return OVS_OK;   // upstream: empty image -> early return
    if (cap < 0 || (cap > 0 && (!kps_left || !desc_left || !kps_right || !desc_right)) || stride < (size_t)cols) return OVS_ERR_INVALID;
    if ((mask_left == nullptr) != (mask_right == nullptr) || (mask_left && mask_stride < (size_t)cols)) return OVS_ERR_INVALID;
    if (h->max_batch < 2 || rows > h->max_rows || cols > h->max_cols) return OVS_ERR_CAPACITY;
    if (h->n_submitted != h->n_collected) return OVS_ERR_INVALID;   // frames submitted asynchronously must be collected first
    OVS_HIP_TRY(hipSetDevice(h->device));
    ovs_status st = ensure_geometry(h, rows, cols);
    if (st != OVS_OK) return st;
    ovs_orb::PairBuf& pb = h->pair;
    if (!pb.d_img) {
        pb.frame_bytes = h->img_pitch * (size_t)h->max_rows;
        OVS_HIP_TRY(hipMalloc(&pb.d_img, 2 * pb.frame_bytes));
        OVS_HIP_TRY(hipEventCreateWithFlags(&pb.ev_h2d, hipEventDisableTiming));
    }
    if (pb.cap < h->out_cap_variant[1]) {   // [counts | keypoints of both frames | descriptors of both frames], for the larger capacity
        if (pb.d_out) (void)hipFree(pb.d_out);
        if (pb.h_out) (void)hipHostFree(pb.h_out);
        pb.d_out = pb.h_out = nullptr;
        pb.cap = h->out_cap_variant[1];
        const size_t bytes = 32 + ((2 * sizeof(ovs_keypoint) * (size_t)pb.cap + 31) & ~(size_t)31) + (size_t)64 * pb.cap;
        OVS_HIP_TRY(hipMalloc(&pb.d_out, bytes));
        OVS_HIP_TRY(hipHostMalloc(&pb.h_out, bytes, hipHostMallocDefault));
    }
    const int fcap = h->out_cap;   // per frame, current variant
    pb.off_kps = 32;
    pb.off_desc = 32 + ((2 * sizeof(ovs_keypoint) * (size_t)fcap + 31) & ~(size_t)31);
    pb.out_bytes = pb.off_desc + (size_t)64 * fcap;
    hipStream_t s = h->stream, cs = h->copy_stream;
    st = upload_plane(h, left, stride, rows, cols, nullptr, pb.d_img);
    if (st == OVS_OK) st = upload_plane(h, right, stride, rows, cols, nullptr, pb.d_img + pb.frame_bytes);
    if (st != OVS_OK) return st;
    if (mask_left) {
        if (!pb.d_mask) OVS_HIP_TRY(hipMalloc(&pb.d_mask, 2 * pb.frame_bytes));
        st = upload_plane(h, mask_left, mask_stride, rows, cols, nullptr, pb.d_mask);
        if (st == OVS_OK) st = upload_plane(h, mask_right, mask_stride, rows, cols, nullptr, pb.d_mask + pb.frame_bytes);
        if (st != OVS_OK) return st;
    }
    OVS_HIP_TRY(hipEventRecord(pb.ev_h2d, cs));
    OVS_HIP_TRY(hipStreamWaitEvent(s, pb.ev_h2d, 0));
    st = run_extract(h, pb.d_img, 2, rows, cols, h->img_pitch, pb.frame_bytes, mask_left ? pb.d_mask : nullptr,
                     reinterpret_cast<ovs_keypoint*>(pb.d_out + pb.off_kps), pb.d_out + pb.off_desc, reinterpret_cast<int32_t*>(pb.d_out), fcap, s);
    if (st != OVS_OK) return st;
    OVS_HIP_TRY(hipMemcpyAsync(pb.h_out, pb.d_out, pb.out_bytes, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    const int32_t* cnt = reinterpret_cast<const int32_t*>(pb.h_out);
    ovs_keypoint* const kout[2] = {kps_left, kps_right};
    uint8_t* const dout[2] = {desc_left, desc_right};
    int32_t* const nout[2] = {n_left, n_right};
    bool over = false;
    for (int f = 0; f < 2; ++f) {
        const int32_t m = std::min(cnt[f], cap);
        if (m > 0) {
            std::memcpy(kout[f], pb.h_out + pb.off_kps + sizeof(ovs_keypoint) * (size_t)fcap * f, sizeof(ovs_keypoint) * (size_t)m);
            std::memcpy(dout[f], pb.h_out + pb.off_desc + (size_t)32 * fcap * f, (size_t)32 * m);
        }
        *nout[f] = m;
        over |= cnt[f] > cap;
    }
    h->last_host_count = *n_left;
    return over ? OVS_ERR_CAPACITY : OVS_OK;
}

ovs_status ovs_orb_set_host_pyramid(ovs_orb* h, int32_t enable) {
    if (!h) return OVS_ERR_INVALID;
    h->host_pyr = enable != 0;
    return OVS_OK;
}

ovs_status ovs_orb_set_host_mode(ovs_orb* h, int32_t mode) {
    if (!h || mode < 0 || mode > 1) return OVS_ERR_INVALID;
    h->host_mode = mode;
    return OVS_OK;
}

ovs_status ovs_orb_host_pyramid_level(const ovs_orb* h, int32_t level, const uint8_t** base, int32_t* rows, int32_t* cols, int32_t* pitch) {
    if (!h || level < 0 || level >= h->p.num_levels || h->last_slot < 0) return OVS_ERR_INVALID;
    const ovs_orb::HostSlot& sl = h->slot[h->last_slot];
    const LevelGeo& g = h->geo.lv[level];
    if (rows) *rows = g.rows;
    if (cols) *cols = g.cols;
    if (level == 0) {   // level 0 is the caller's own image (upstream aliases it too)
        if (base) *base = nullptr;
        if (pitch) *pitch = 0;
        return OVS_OK;
    }
    if (!sl.has_pyr || sl.rows != h->cur_rows || sl.cols != h->cur_cols) return OVS_ERR_INVALID;   // enable ovs_orb_set_host_pyramid first
    if (base) *base = sl.h_pyr + g.plane_off;
    if (pitch) *pitch = g.pitch;
    return OVS_OK;
}

ovs_status ovs_orb_host_profile_read(const ovs_orb* h, float* h2d_kernels_d2h_ms) {
    if (!h || !h2d_kernels_d2h_ms) return OVS_ERR_INVALID;
    for (int i = 0; i < 3; ++i) h2d_kernels_d2h_ms[i] = h->host_ms[i];
    return OVS_OK;
}

ovs_status ovs_orb_pyramid_level(ovs_orb* h, int32_t frame, int32_t level, uint8_t* host_dst, int32_t* rows, int32_t* cols) {
    if (!h || level < 0 || level >= h->p.num_levels || frame < 0 || frame >= h->last_batch || !h->last_img0) return OVS_ERR_INVALID;
    const LevelGeo& g = h->geo.lv[level];
    if (rows) *rows = g.rows;
    if (cols) *cols = g.cols;
    if (!host_dst) return OVS_OK;
    OVS_HIP_TRY(hipSetDevice(h->device));
    // only the stream the last extract ran on is waited for -- never the whole device: a second extractor on another thread (stereo
    // left / right) must not be stalled by this getter
    OVS_HIP_TRY(hipStreamSynchronize(h->last_stream));
    hipStream_t s = h->stream;
    if (level == 0)
        OVS_HIP_TRY(hipMemcpy2DAsync(host_dst, g.cols, h->last_img0 + (size_t)frame * h->last_frame_stride0, h->last_stride0, g.cols, g.rows,
                                     hipMemcpyDeviceToHost, s));
    else
        OVS_HIP_TRY(hipMemcpy2DAsync(host_dst, g.cols, h->d.pyr + (size_t)frame * h->d.pyr_frame_bytes + g.plane_off, g.pitch, g.cols, g.rows,
                                     hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    return OVS_OK;
}

ovs_status ovs_orb_debug_candidates(ovs_orb* h, int32_t frame, int32_t level, int32_t* xs, int32_t* ys, int32_t* scores, int32_t cap,
                                    int32_t* n_out) {
    if (!h || !n_out || level < 0 || level >= h->p.num_levels || frame < 0 || frame >= h->last_batch) return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(h->device));
    OVS_HIP_TRY(hipDeviceSynchronize());
    const LevelGeo& g = h->geo.lv[level];
    const int L = h->p.num_levels;
    uint32_t n = 0;
    OVS_HIP_TRY(hipMemcpy(&n, h->d.cand_count + (size_t)frame * L + level, sizeof(uint32_t), hipMemcpyDeviceToHost));
    n = std::min<uint32_t>(n, (uint32_t)g.cand_cap);
    std::vector<uint64_t> c(n);
    if (n)
        OVS_HIP_TRY(hipMemcpy(c.data(), h->d.cand + (size_t)frame * h->d.cand_frame_entries + g.cand_off, n * sizeof(uint64_t),
                              hipMemcpyDeviceToHost));
    const uint32_t ncx = (uint32_t)g.ncx;
    std::sort(c.begin(), c.end(), [ncx](uint64_t a, uint64_t b) {
        return cand_order(cand_x(a), cand_y(a), ncx) < cand_order(cand_x(b), cand_y(b), ncx);
    });
    for (uint32_t i = 0; i < n && (int32_t)i < cap; ++i) {
        if (xs) xs[i] = (int32_t)cand_x(c[i]);
        if (ys) ys[i] = (int32_t)cand_y(c[i]);
        if (scores) scores[i] = (int32_t)cand_score(c[i]);
    }
    *n_out = (int32_t)n;
    return OVS_OK;
}

ovs_status ovs_orb_debug_level_counts(ovs_orb* h, int32_t frame, int32_t* counts) {
    if (!h || !counts || frame < 0 || frame >= h->max_batch) return OVS_ERR_INVALID;
    OVS_HIP_TRY(hipSetDevice(h->device));
    OVS_HIP_TRY(hipDeviceSynchronize());
    std::vector<uint32_t> c(h->p.num_levels);
    OVS_HIP_TRY(hipMemcpy(c.data(), h->d.lvl_count + (size_t)frame * h->p.num_levels, sizeof(uint32_t) * c.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < c.size(); ++i) counts[i] = (int32_t)c[i];
    return OVS_OK;
}

}   // extern "C"
