// Shared helpers of the windowed-matcher shims: the per-thread device context (upstream stack-constructs a matcher per call; the
// device context is kept per thread so a call does not allocate) and camera::base -> ovs_grid_params.
#pragma once
#include <ovslam_hip.h>

#include <atomic>
#include <initializer_list>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../data/frame_stub.h"
#include "../util/device_policy.h"

namespace openvslam {
namespace match {
namespace detail {

struct window_holder {
    ovs_wmatcher* w = nullptr;
    int device = 0;
    int cap_t = 0, cap_q = 0;
    int entries = 1 << 22;   // candidate-list budget; doubled by grow() after OVS_ERR_CAPACITY, up to 2^26 (256 MB of keys)
    ~window_holder() {
        if (w) ovs_wmatcher_destroy(w);
    }
    ovs_wmatcher* get(int n_t, int n_q) {
        if (w && n_t <= cap_t && n_q <= cap_q) return w;
        if (w) ovs_wmatcher_destroy(w);
        w = nullptr;
        cap_t = n_t < 8192 ? 8192 : n_t;
        cap_q = n_q < 16384 ? 16384 : n_q;
        const int st = ovs_wmatcher_create(cap_t, cap_q, entries, device, &w);
        if (st != OVS_OK) {
            w = nullptr;
            cap_t = cap_q = 0;
            throw util::device_error(st, std::string("ovs_wmatcher_create: ") + ovs_last_error());   // caught by guarded()
        }
        return w;
    }
    void reset() {
        if (w) ovs_wmatcher_destroy(w);
        w = nullptr;
        cap_t = cap_q = 0;
    }
    bool grow() {
        if (entries >= (1 << 26)) return false;
        entries <<= 1;
        reset();   // rebuilt with the larger budget by the next get()
        return true;
    }
};
// one matcher context per (thread, device): a frame is matched on the device its extractor ran on (frame i -> GPU i mod G)
inline window_holder& window_ctx(int device = 0) {
    thread_local std::vector<std::unique_ptr<window_holder>> per_device;
    if (device < 0) device = 0;
    if ((size_t)device >= per_device.size()) per_device.resize((size_t)device + 1);
    if (!per_device[(size_t)device]) {
        per_device[(size_t)device] = std::make_unique<window_holder>();
        per_device[(size_t)device]->device = device;
    }
    return *per_device[(size_t)device];
}

// Per-thread staging vectors of the shims, reused from call to call: a tracked frame flattens ~2000 landmarks into half a dozen arrays twice
// (match_current_and_last_frames, match_frame_and_landmarks); fresh std::vectors cost allocation, zero-fill and first-touch page faults of
// ~110 KB per call (~10 us of a 0.08 - 0.14 ms call). resize() keeps the old contents: entries a call does not write are masked by its `valid`
// array, which every call writes completely; the fill loops also give masked entries defined values (zeros) for the numeric fields.
template <class T>
inline std::vector<T>& scratch_vec(int slot, size_t n) {
    thread_local std::vector<T> v[8];
    v[slot].resize(n);
    return v[slot];
}

inline ovs_grid_params grid_of(const camera::base* cam) {
    ovs_grid_params gp;
    gp.min_x = cam->img_bounds_.min_x_;
    gp.min_y = cam->img_bounds_.min_y_;
    gp.max_x = cam->img_bounds_.max_x_;
    gp.max_y = cam->img_bounds_.max_y_;
    gp.cols = (int32_t)cam->num_grid_cols_;
    gp.rows = (int32_t)cam->num_grid_rows_;
    return gp;
}

inline ovs_camera camera_of(const camera::base* cam) {
    // camera::fisheye and camera::radial_division: upstream undistorts a frame's keypoints in its constructor and both models'
    // reproject_to_image are the pinhole projection with the model's fx, fy, cx, cy on those undistorted coordinates (the distortion lives in
    // undistort_keypoints / convert_keypoints_to_bearings only, outside this path) -- oracle/ORACLE_SPEC.md rule 31. So on the device they ARE
    // the perspective model; only camera::equirectangular projects differently.
    ovs_camera c;
    c.model = cam->model_type_ == camera::model_type_t::Equirectangular ? 1 : 0;
    c.setup = (int32_t)cam->setup_type_;
    c.fx = cam->fx_;
    c.fy = cam->fy_;
    c.cx = cam->cx_;
    c.cy = cam->cy_;
    c.focal_x_baseline = cam->focal_x_baseline_;
    c.true_baseline = cam->true_baseline_;
    c.cols = (int32_t)cam->cols_;
    c.rows = (int32_t)cam->rows_;
    return c;
}

// the device-side cache of a frame or keyframe: uploaded + indexed on first use (on the cache's device), then shared by every later matcher
// call on the object, its copies and the keyframe made from it. The caller keeps the returned reference for the duration of its ABI call.
template <class F>   // data::frame or data::keyframe: the same members
inline std::shared_ptr<void> device_handle_of(const F& frm, bool want_bearings = false) {
    data::frame_device_cache& cache = *frm.device_cache_;
    return cache.get(
        [&]() -> std::shared_ptr<void> {
            const ovs_grid_params gp = grid_of(frm.camera_);
            ovs_frame_dev* f = nullptr;
            const bool stereo = !frm.stereo_x_right_.empty();
            const int st = ovs_frame_dev_create(cache.device, &gp, reinterpret_cast<const ovs_keypoint*>(frm.undist_keypts_.data()), frm.descriptors_.data,
                                                stereo ? frm.stereo_x_right_.data() : nullptr, (int32_t)frm.undist_keypts_.size(), &f);
            if (st != OVS_OK) throw util::device_error(st, std::string("ovs_frame_dev_create: ") + ovs_last_error());   // caught by guarded()
            return std::shared_ptr<void>(f, [](void* h) { ovs_frame_dev_destroy(static_cast<ovs_frame_dev*>(h)); });
        },
        want_bearings,
        [&](void* h) {
            std::vector<double> b(3 * frm.bearings_.size());
            for (size_t i = 0; i < frm.bearings_.size(); ++i)
                for (int a = 0; a < 3; ++a) b[3 * i + (size_t)a] = frm.bearings_[i](a);
            const int st = ovs_frame_dev_attach_bearings(static_cast<ovs_frame_dev*>(h), b.data());
            if (st != OVS_OK) throw util::device_error(st, std::string("ovs_frame_dev_attach_bearings: ") + ovs_last_error());
        });
}
// The handle of `frm` for a call that runs on `device`. Two-object matchers (bow_tree, area, match_for_triangulation,
// match_keyframes_mutually) run on the first object's device; with frames spread over several GPUs (frame i -> GPU i mod G) the second
// object may be resident elsewhere, and the ABI refuses a handle of another device (OVS_ERR_INVALID). Then a temporary handle is built on
// the call's device from the host members (one upload, freed when the call returns); the object's own cache stays where it is.
inline std::atomic<int>& foreign_handles_built() {   // counts the temporary handles; force_foreign_handles() makes EVERY call build one (test hook:
    static std::atomic<int> n{0};                     // the foreign-device path on a one-GPU box)
    return n;
}
inline std::atomic<bool>& force_foreign_handles() {
    static std::atomic<bool> on{false};
    return on;
}
template <class F>
inline std::shared_ptr<void> device_handle_on(const F& frm, int device, bool want_bearings = false) {
    if (frm.device_cache_->device == device && !force_foreign_handles().load()) return device_handle_of(frm, want_bearings);
    ++foreign_handles_built();
    const ovs_grid_params gp = grid_of(frm.camera_);
    ovs_frame_dev* f = nullptr;
    const bool stereo = !frm.stereo_x_right_.empty();
    int st = ovs_frame_dev_create(device, &gp, reinterpret_cast<const ovs_keypoint*>(frm.undist_keypts_.data()), frm.descriptors_.data,
                                  stereo ? frm.stereo_x_right_.data() : nullptr, (int32_t)frm.undist_keypts_.size(), &f);
    if (st != OVS_OK) throw util::device_error(st, std::string("ovs_frame_dev_create (foreign device): ") + ovs_last_error());
    std::shared_ptr<void> h(f, [](void* p) { ovs_frame_dev_destroy(static_cast<ovs_frame_dev*>(p)); });
    if (want_bearings) {
        std::vector<double> b(3 * frm.bearings_.size());
        for (size_t i = 0; i < frm.bearings_.size(); ++i)
            for (int a = 0; a < 3; ++a) b[3 * i + (size_t)a] = frm.bearings_[i](a);
        st = ovs_frame_dev_attach_bearings(f, b.data());
        if (st != OVS_OK) throw util::device_error(st, std::string("ovs_frame_dev_attach_bearings (foreign device): ") + ovs_last_error());
    }
    return h;
}
inline const ovs_frame_dev* dev(const std::shared_ptr<void>& h) { return static_cast<const ovs_frame_dev*>(h.get()); }
template <class F>
inline int device_of(const F& frm) { return frm.device_cache_->device; }

// rows 0..2 of a 4x4 [R|t] -> 12 doubles: rotation row-major, then translation
inline void pose12(const Mat44_t& T, double* out) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out[3 * i + j] = T(i, j);
        out[9 + i] = T(i, 3);
    }
}

inline void flatten_bow(const data::bow_feature_vector& fv, std::vector<int32_t>& ids, std::vector<int32_t>& start, std::vector<int32_t>& items) {
    ids.clear();
    items.clear();
    start.assign(1, 0);
    for (const auto& node : fv) {   // std::map iterates in ascending node id
        ids.push_back((int32_t)node.first);
        for (const auto idx : node.second) items.push_back((int32_t)idx);
        start.push_back((int32_t)items.size());
    }
}

// One ABI call of a windowed matcher under the failure policy of util/device_policy.h: `call` re-evaluates window_ctx().get(...) and
// device_frame_of(...) each time it runs; before the retry the thread's matcher context and the device caches of the frames the call
// uses are dropped. false -> the caller returns zero matches (its outputs may be partly written: it must not read them).
template <class Call>
inline bool guarded(const char* what, Call&& call, std::initializer_list<data::frame_device_cache*> caches = {}, int device = 0) {
    return util::run_guarded(
        what, call,
        [&] {
            window_ctx(device).reset();
            for (data::frame_device_cache* c : caches)
                if (c) c->drop();
        },
        [device] { return window_ctx(device).grow(); });
}

static_assert(sizeof(cv::KeyPoint) == sizeof(ovs_keypoint), "cv::KeyPoint crosses the ABI as ovs_keypoint");

}   // namespace detail
}   // namespace match
}   // namespace openvslam
