// ovs_common.h -- internal declarations shared by the HIP translation units of libovslam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <string>
#include <stdint.h>

#include "../../include/ovslam_hip.h"
#include "../../include/ovs_detmath.h"

namespace ovs {

constexpr int kOrbPatchRadius = 19;   // orb_extractor::orb_patch_radius_
constexpr int kFastPatchSize = 31;    // fast_patch_size_
constexpr int kCellSize = 64;         // compute_fast_keypoints cell_size
constexpr int kCellOverlap = 6;       // compute_fast_keypoints overlap
constexpr int kDetOrigin = kOrbPatchRadius + 3;   // first pixel FAST can test (cell's own 3-px dead frame)

// One bilinear tap pair of cv::resize's 11-bit fixed-point tables (computed on the host, double/float as OpenCV does).
struct ResizeTap {
    uint16_t o0, o1;   // source indices (o1 = min(o0+1, size-1))
    int16_t a0, a1;    // 11-bit coefficients
};

// Geometry of one pyramid level for the handle's current (rows, cols). Lives in device memory; kernels read it uniformly.
struct LevelGeo {
    int32_t rows, cols;
    int32_t pitch;            // bytes per row of this level's plane (levels >= 1; level 0 uses the caller's stride)
    int32_t ncx, ncy;         // valid FAST cells
    int32_t cell_base;        // prefix sum of ncx*ncy over lower levels
    int32_t max_bx, max_by;   // cols-19, rows-19
    int32_t n_keypts;         // N_level (num_keypts_per_level_)
    int32_t kp_cap;           // N_level + 3
    int32_t kp_base;          // prefix sum of kp_cap over lower levels
    int32_t cand_cap;         // capacity of this level's candidate list
    int32_t gx, gy;           // root grid of the quad-tree
    int32_t max_nodes;        // 4*N_level + 16
    float inv_ncx;            // 1 / ncx (k_fast_cells_v3: cell -> (row, column) through a float product)
    int32_t resize_hwin_ok;   // level >= 1: every tile of k_resize_linear_u8 (v4) finds its source rectangle inside the LDS tile and every
                              // column pair's four source bytes inside two aligned words (checked on the host against the tap tables)
    uint32_t ncx_magic;       // ceil(2^32 / ncx): cell / ncx == umulhi(cell, ncx_magic) for cell * ncx < 2^32 (scalar unit, no conversions)
    int64_t plane_off;        // byte offset of the plane inside one frame's pyramid block (levels >= 1)
    int64_t cand_off;         // entry offset of the candidate list inside one frame's candidate block
    int64_t node_off;         // entry offset of the node scratch inside one frame's node block
    int64_t xtab_off, ytab_off;   // ResizeTap offsets (level >= 1: taps from level-1 to level)
    double dx, dy;            // root patch size
    float scale;              // scale_factors_[level]
    float kp_size;            // (float)(unsigned)(31*scale)
};

// One FAST cell of the current geometry (k_fast_cells reads ITS cells' records with one load instead of deriving them from the level tables
// through a chain of dependent scalar loads): cell origin, clipped size, level. 8 bytes, table order = cell id.
struct CellDesc {
    uint16_t min_x, min_y;   // 19 + 64 * column / row
    uint8_t cw, ch;          // min(70, distance to the level's border): the tile is cw x ch, the testable area (cw - 6) x (ch - 6)
    uint8_t level, pad;
};
static_assert(sizeof(CellDesc) == 8, "CellDesc is read as one 64-bit word");

struct FrameGeo {
    int32_t num_levels;
    int32_t total_cells;
    int32_t total_kp_cap;
    int32_t ini_thr, min_thr;
    int32_t variant;   // ORACLE_SPEC rules 6, 7, 10 as run-time variants: bit 0 quad-tree switch factor 1 (default 3), bit 1 equal-count tie order
                       // earlier-created first (default later first), bit 2 blur taps 18,34,49,55 saturating (default 18,34,48,56), bit 3 steering by libm's cosf / sinf (default util::cos / util::sin)
    int32_t cell_base_tab[OVS_MAX_LEVELS];   // lv[l].cell_base for l < num_levels, INT32_MAX above: one scalar load finds a cell's level
    LevelGeo lv[OVS_MAX_LEVELS];
};

// Candidate entry (u64): [score:8 | y:13 | x:13 | node:16], score = FAST response (S-1), x/y = level-image coordinates.
__host__ __device__ inline uint64_t cand_pack(uint32_t x, uint32_t y, uint32_t score, uint32_t node) {
    return ((uint64_t)score << 42) | ((uint64_t)y << 29) | ((uint64_t)x << 16) | (uint64_t)node;
}
__host__ __device__ inline uint32_t cand_x(uint64_t c) { return (uint32_t)(c >> 16) & 0x1FFFu; }
__host__ __device__ inline uint32_t cand_y(uint64_t c) { return (uint32_t)(c >> 29) & 0x1FFFu; }
__host__ __device__ inline uint32_t cand_score(uint64_t c) { return (uint32_t)(c >> 42) & 0xFFu; }
__host__ __device__ inline uint32_t cand_node(uint64_t c) { return (uint32_t)c & 0xFFFFu; }
// Upstream's emission order of a candidate: cell row, cell column, then row-major inside the cell.
__host__ __device__ inline uint32_t cand_order(uint32_t x, uint32_t y, uint32_t ncx) {
    const uint32_t dx = x - kDetOrigin, dy = y - kDetOrigin;
    return (((dy >> 6) * ncx + (dx >> 6)) << 12) | ((dy & 63u) << 6) | (dx & 63u);
}

// Selected keypoint of a level before description (u64): [score:8 | y:13 | x:13].
struct DevBuffers {
    const FrameGeo* geo;          // device copy
    const ResizeTap* taps;        // device
    const CellDesc* cells;        // device, total_cells records
    uint8_t* pyr;                 // max_batch * pyr_frame_bytes (levels >= 1)
    size_t pyr_frame_bytes;
    uint64_t* cand;               // max_batch * cand_frame_entries
    size_t cand_frame_entries;
    uint32_t* cand_count;         // max_batch * num_levels
    uint32_t* nodes;              // max_batch * node_frame_entries * 4 u32 (two ping-pong lists inside)
    size_t node_frame_entries;
    uint64_t* lvl_kps;            // max_batch * total_kp_cap
    uint32_t* lvl_count;          // max_batch * num_levels
};

// k_pyramid_chain (orb_pyramid.hip): what one tile column (or row) of the TX x TY tile grid covers of one level: the pixels it COMPUTES
// [c0, c1) and, inside them, the pixels it OWNS, i.e. stores to the level's plane, [o0, o1). Level 0: the source rectangle to stage (c0 of
// a column span is a multiple of 4), nothing owned.
struct ChainSpan {
    int16_t c0, c1, o0, o1;
};
// k_resize_pair_u8 (orb_pyramid.hip): what one tile column (or tile row) of the UPPER level's 128 x 32 tile grid needs of the two levels below it.
// Column spans: s_lo = origin of stage B's LDS tile (first tap & ~15), b0 = first middle column the tile computes (first tap & ~3), n = 4-pixel
// groups of the middle region, own = groups it stores (up to the next tile's b0; all of them for the last tile), a0 = first lower column staged
// (& ~15), an = 32-bit words per staged row. Row spans: s_lo = b0 = first middle row, n = rows of the region, own = rows stored, a0 / an = first
// lower row / rows staged.
struct PairSpan {
    int16_t s_lo, b0, n, own, a0, an, pad0, pad1;
};
static_assert(sizeof(PairSpan) == 16, "PairSpan is read as one 128-bit word");
// k_resize_pair_u8's horizontal pass for the column pair (2 j, min(2 j + 1, cols - 1)) of a level, built on the host from the level's x taps
// (orb_api.hip build_htaps; the kernel used to derive it per thread and tile: ~25 vector instructions three times over): the two v_perm_b32
// selectors that pair (S[o0], S[o1]) out of the two aligned source words starting at word wa, and the two coefficient words (16 a0 | 16 a1 << 16).
struct HTapRec {
    uint32_t sel_a, sel_b, ca, cb;
    uint32_t wa;   // o0(2 j) >> 2, in words from the start of the source row
    uint32_t pad0, pad1, pad2;
};
static_assert(sizeof(HTapRec) == 32, "HTapRec: one 128-bit load + one word");
constexpr int kChainMaxW = 192;        // widest computed region of a level >= 1 (three 64-lane column chunks)
constexpr int kChainMaxW0 = 256;       // widest staged level-0 rectangle (64 words)
constexpr int kChainMaxH0 = 128;       // its height (16 waves x 8 rows)
constexpr int kChainMaxLds = 64 * 1024;
size_t chain_lds_bytes(int bufA_bytes, int bufB_bytes, int tap_cap);   // dynamic LDS of one k_pyramid_chain workgroup (bufA, bufB multiples of 16)

// ---- kernel launchers (each defined next to its kernel) ----
hipError_t launch_pyramid_chain(const uint8_t* img0, size_t frame_stride0, int pitch0, uint8_t* pyr, size_t pyr_frame_bytes, const FrameGeo* d_geo,
                                const ResizeTap* d_taps, const ChainSpan* d_plan, int TX, int TY, int bufA_bytes, int bufB_bytes, int tap_cap,
                                int batch, hipStream_t s);
hipError_t launch_resize_pair(const uint8_t* src, size_t src_frame_stride, int src_pitch, uint8_t* dst1, int pitch1, int rows1, int cols1,
                              uint8_t* dst2, int pitch2, int rows2, int cols2, size_t dst_frame_stride, const ResizeTap* yt1, const ResizeTap* yt2,
                              const HTapRec* ht1, const HTapRec* ht2, const PairSpan* plan, int batch, hipStream_t s);
bool resize_pair_launchable(const uint8_t* src, size_t src_frame_stride, int src_pitch, int rows2, int cols2, int batch);
hipError_t launch_resize(const uint8_t* src, size_t src_frame_stride, int src_pitch, int srows, int scols, uint8_t* dst,
                         size_t dst_frame_stride, int dst_pitch, int drows, int dcols, const ResizeTap* xt, const ResizeTap* yt,
                         int batch, hipStream_t s, int hwin_ok);
hipError_t launch_fast(const FrameGeo& hgeo, const DevBuffers& d, const uint8_t* img0, size_t stride0, size_t frame_stride0,
                       const uint8_t* mask, int mask_rows, int batch, hipStream_t s, int cell_lo = 0, int n_cells = -1);
hipError_t launch_tree(const FrameGeo& hgeo, const DevBuffers& d, int batch, hipStream_t s, int level_lo = 0, int n_levels = -1);
size_t tree_lds_bytes_for(const FrameGeo& hgeo);
constexpr size_t kMaxLdsPerWorkgroup = 160 * 1024;
hipError_t launch_describe(const FrameGeo& hgeo, const DevBuffers& d, const uint8_t* img0, size_t stride0, size_t frame_stride0,
                           ovs_keypoint* kps, uint8_t* desc, int32_t* counts, int cap, int batch, hipStream_t s, ovs_keypoint* kps_m = nullptr, uint8_t* desc_m = nullptr,
                           int32_t* counts_m = nullptr);

// Tuning / A-B switches of the launchers. Read from the environment ONCE per process (first use; thread-safe static initialisation) -- the
// launch paths never call getenv, which is not safe against a host application's concurrent setenv. All default to the production setting.
constexpr int kMaxTuningDevices = 16;
struct Tuning {
    int fast_impl;         // OVS_FAST_IMPL: 1 / unset = k_fast_cells (one workgroup per cell), 2 = k_fast_wave (round 6 experiment: one wavefront per cell, no barriers; same outputs, same batch time, 3x the single-frame time)
    int fast_map;          // OVS_FAST_MAP: 1 / unset = groups fastest inside an XCD's share (round 6), 0 = frames fastest (rounds 3-5)
    int fast_pf;           // OVS_FAST_PF: k_fast_wave's L2 prefetch distance in groups (0 = none)
    int fast_cells;        // OVS_FAST_CELLS: consecutive cells per FAST workgroup (0 = by launch size)
    int fast_pad_lds;      // OVS_FAST_PAD_LDS: extra dynamic LDS per k_fast_cells workgroup (occupancy probe)
    bool fast_timing;      // OVS_FAST_TIMING: per-phase cycle counts of k_fast_cells, printed per launch
    bool describe_xcd;     // OVS_DESCRIBE_XCD=0: plain frame-major order in k_describe
    int resolve_wide_from; // OVS_RESOLVE_WIDE_FROM: queries from which a resolver round takes 512 of them
    int pose_threads;      // OVS_POSE_THREADS: 256 / 512 (0 = by problem size)
    int pose_groups;       // OVS_POSE_GROUPS: workgroups a single frame's pose optimisation is spread over (0 = by observation count, 1 = one)
    bool ba_trace;         // OVS_BA_TRACE: per-iteration trace of the LM loops on stderr
    bool ba_backsub_edges; // OVS_BA_BACKSUB_EDGES=0: k_trial_update back-substitutes one lane per LANDMARK walking its edges (rounds 4-5) instead of one lane per edge (round 6; same bits)
    bool ba_dev_outliers;  // OVS_BA_DEV_OUTLIERS=0: ovs_local_ba_optimize downloads the per-edge chi2 / depth arrays and judges the edges on the host (rounds 1-5) instead of on the device (round 6; same flags)
    bool chol_resident;    // OVS_CHOL_RESIDENT=0: systems up to 288 unknowns take k_chol_solve (tiles through memory) instead of k_chol_resident
    bool pyr_pair;         // OVS_PYR_PAIR=0: batches build the pyramid level by level (k_resize_linear_u8 x7) instead of two levels per launch (k_resize_pair_u8, round 6)
    int pyr_chain;         // OVS_PYR_CHAIN: frames per launch up to which the pyramid is ONE k_pyramid_chain launch (default 2: measured 24 vs 37 us for one frame, 33.5 vs 35.7 for two, 73 vs 46 for eight; 0 = never)
};
const Tuning& tuning();

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: remember the largest size set for one kernel on every device
struct LdsAttrCache {
    std::atomic<size_t> set[kMaxTuningDevices] = {};
    std::mutex mu;   // hipFuncSetAttribute is last-writer-wins: two threads' first launches with different sizes must not leave the smaller one set
};
inline hipError_t ensure_dynamic_lds(const void* fn, size_t bytes, LdsAttrCache& c) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool cached = dev >= 0 && dev < kMaxTuningDevices;
    if (cached && bytes <= c.set[dev].load(std::memory_order_acquire)) return hipSuccess;   // fast path: no lock once the size is covered
    std::lock_guard<std::mutex> lock(c.mu);
    if (cached && bytes <= c.set[dev].load(std::memory_order_acquire)) return hipSuccess;   // another thread set at least this much meanwhile
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && cached) c.set[dev].store(bytes, std::memory_order_release);       // under the lock the attribute only ever grows
    return e;
}

void set_last_error(const char* what, hipError_t e);
// ovs_debug_inject_hip_failures(skip, n): after `skip` further OVS_HIP_TRY-checked calls, the next n of them (the calls themselves
// still run) report hipErrorLaunchFailure. One relaxed atomic load per checked HIP call when nothing is armed.
extern std::atomic<int32_t> g_injected_hip_failures, g_injected_hip_skip;
inline hipError_t fault_filter(hipError_t e) {
    if (g_injected_hip_failures.load(std::memory_order_relaxed) > 0) {
        if (g_injected_hip_skip.load(std::memory_order_relaxed) > 0 && g_injected_hip_skip.fetch_sub(1, std::memory_order_relaxed) > 0) return e;
        if (g_injected_hip_failures.fetch_sub(1, std::memory_order_relaxed) > 0) return hipErrorLaunchFailure;
    }
    return e;
}
void set_last_error_text(const std::string& what);   // failures that are not HIP errors (a refused file, an exception stopped at the ABI)

// Device view of one frame's image pyramid (feature::orb_extractor::image_pyramid_) of an extractor's LAST extract.
struct PyrView {
    const uint8_t* base[OVS_MAX_LEVELS];
    int32_t pitch[OVS_MAX_LEVELS];
    int32_t rows[OVS_MAX_LEVELS], cols[OVS_MAX_LEVELS];
    float scale[OVS_MAX_LEVELS], inv_scale[OVS_MAX_LEVELS];
    int32_t num_levels;
};
// false if the handle has not extracted yet or `frame` is outside its last batch
bool orb_pyramid_view(const ovs_orb* h, int frame, PyrView* out);
int orb_device(const ovs_orb* h);
hipStream_t orb_last_stream(const ovs_orb* h);   // the stream the handle's last extract was enqueued on
// true iff (kps, desc, n) equal what the handle's last HOST-form extract returned; then *d_kps / *d_desc are the device copies still resident
bool orb_host_outputs_equal(const ovs_orb* h, const ovs_keypoint* kps, const uint8_t* desc, int32_t n, const ovs_keypoint** d_kps, const uint8_t** d_desc);

// Stage timer for bench.py: HIP events recorded on the launch stream at stage boundaries; a small ring of call slots.
template <int NSTAGE>
struct StageProfiler {
    static constexpr int kSlots = 64;
    bool enabled = false;
    hipEvent_t ev[kSlots][NSTAGE + 1] = {};
    bool created = false;
    int head = 0, pending = 0;      // slots [head-pending, head) hold un-read calls
    double acc_ms[NSTAGE] = {};
    int acc_calls = 0;
    int cur = -1;

    hipError_t ensure() {
        if (created) return hipSuccess;
        for (int i = 0; i < kSlots; ++i)
            for (int j = 0; j <= NSTAGE; ++j) {
                hipError_t e = hipEventCreate(&ev[i][j]);
                if (e != hipSuccess) return e;
            }
        created = true;
        return hipSuccess;
    }
    hipError_t drain_one() {
        const int slot = ((head - pending) % kSlots + kSlots) % kSlots;
        hipError_t e = hipEventSynchronize(ev[slot][NSTAGE]);
        if (e != hipSuccess) return e;
        for (int j = 0; j < NSTAGE; ++j) {
            float ms = 0;
            e = hipEventElapsedTime(&ms, ev[slot][j], ev[slot][j + 1]);
            if (e != hipSuccess) return e;
            acc_ms[j] += ms;
        }
        ++acc_calls;
        --pending;
        return hipSuccess;
    }
    // begin a call: returns the slot to record into (-1 when disabled)
    hipError_t begin(hipStream_t s) {
        cur = -1;
        if (!enabled) return hipSuccess;
        hipError_t e = ensure();
        if (e != hipSuccess) return e;
        if (pending == kSlots) {
            e = drain_one();
            if (e != hipSuccess) return e;
        }
        cur = head % kSlots;
        head = (head + 1) % kSlots;
        ++pending;
        return hipEventRecord(ev[cur][0], s);
    }
    hipError_t mark(int stage_done, hipStream_t s) {   // stage_done in 1..NSTAGE
        if (cur < 0) return hipSuccess;
        return hipEventRecord(ev[cur][stage_done], s);
    }
    hipError_t read(float* ms, int32_t* ncalls) {
        while (pending > 0) {
            hipError_t e = drain_one();
            if (e != hipSuccess) return e;
        }
        for (int j = 0; j < NSTAGE; ++j) {
            ms[j] = (float)acc_ms[j];
            acc_ms[j] = 0;
        }
        *ncalls = acc_calls;
        acc_calls = 0;
        return hipSuccess;
    }
    void destroy() {
        if (!created) return;
        for (int i = 0; i < kSlots; ++i)
            for (int j = 0; j <= NSTAGE; ++j) (void)hipEventDestroy(ev[i][j]);
        created = false;
    }
};

// 1 / sqrt(p) by v_rsq_f64 + three Newton steps (enough from an 11-bit seed); sqrt(p) = p * that. A Cholesky pivot then costs ~15 dependent
// operations instead of the ~45 of an IEEE sqrt followed by an IEEE division (results differ from those by an ulp or two): the pivots are
// the one chain of a factorisation nothing can overlap (ba_solve.hip: 6 n_free of them; pose_opt.hip: 6 per LM trial on one lane).
#ifdef __HIPCC__
__device__ __forceinline__ double rsqrt_newton(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
#pragma unroll
    for (int i = 0; i < 3; ++i) y = y * __builtin_fma(-h * y, y, 1.5);
    return y;
}
#endif

}   // namespace ovs

#define OVS_HIP_TRY(expr)                                  \
    do {                                                   \
        hipError_t _e = ovs::fault_filter(expr);           \
        if (_e != hipSuccess) {                            \
            ovs::set_last_error(#expr, _e);                \
            return OVS_ERR_HIP;                            \
        }                                                  \
    } while (0)
