// Host-only checks of the shims' failure policy (openvslam/util/device_policy.h). No device is needed:
//   1. run_guarded() itself, driven by scripted statuses: OVS_OK passes through; OVS_ERR_HIP drops the contexts and issues the call once
//      more; a second failure, or any other status, or a device_error thrown by a context holder, ends in `false` (the empty result);
//      the counters say what happened; nothing ever propagates.
//   2. On a box WITHOUT a HIP device (ovs_device_count() == 0; skipped otherwise) the classes with upstream's signatures answer with the
//      empty result of every function instead of throwing: that is what a tracker sees when the GPU is gone.
// usage: test_policy_host      (prints one line per check; exit code 0 = all held)
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "openvslam/feature/orb_extractor.h"
#include "openvslam/match/area.h"
#include "openvslam/match/robust.h"
#include "openvslam/optimize/pose_optimizer.h"
#include "openvslam/util/device_policy.h"

using namespace openvslam;

static int g_bad = 0;
static void expect(bool ok, const char* what) {
    std::printf("%s %s\n", ok ? "ok  " : "FAIL", what);
    if (!ok) ++g_bad;
}

struct Snapshot {
    unsigned long failed, retried, recovered, degraded;
    static Snapshot take() {
        auto& c = util::device_failures();
        return {c.failed_calls.load(), c.retried.load(), c.recovered.load(), c.degraded.load()};
    }
    bool moved_by(const Snapshot& before, unsigned long f, unsigned long r, unsigned long rec, unsigned long d) const {
        return failed - before.failed == f && retried - before.retried == r && recovered - before.recovered == rec && degraded - before.degraded == d;
    }
};

static void check_run_guarded() {
    int calls = 0, resets = 0;
    auto scripted = [&](std::vector<int> statuses) {
        calls = resets = 0;
        return util::run_guarded(
            "scripted", [&, statuses] { return statuses[(size_t)calls++ < statuses.size() ? (size_t)calls - 1 : statuses.size() - 1]; }, [&] { ++resets; });
    };
    Snapshot s0 = Snapshot::take();
    expect(scripted({OVS_OK}) && calls == 1 && resets == 0 && Snapshot::take().moved_by(s0, 0, 0, 0, 0), "OVS_OK: one call, no reset, counters untouched");
    s0 = Snapshot::take();
    expect(scripted({OVS_ERR_HIP, OVS_OK}) && calls == 2 && resets == 1 && Snapshot::take().moved_by(s0, 1, 1, 1, 0),
           "OVS_ERR_HIP then OVS_OK: contexts dropped once, the retry's result is the caller's");
    s0 = Snapshot::take();
    expect(!scripted({OVS_ERR_HIP, OVS_ERR_HIP, OVS_OK}) && calls == 2 && resets == 1 && Snapshot::take().moved_by(s0, 1, 1, 0, 1),
           "OVS_ERR_HIP twice: exactly one retry, then the empty result");
    s0 = Snapshot::take();
    expect(!scripted({OVS_ERR_NO_DEVICE, OVS_OK}) && calls == 1 && resets == 0 && Snapshot::take().moved_by(s0, 1, 0, 0, 1),
           "OVS_ERR_NO_DEVICE: no retry, the empty result");
    // deterministic caller-side statuses surface as exceptions instead of a tracker that silently never matches
    s0 = Snapshot::take();
    bool threw = false;
    try {
        scripted({OVS_ERR_CAPACITY});
    } catch (const std::length_error&) {
        threw = true;
    }
    expect(threw && calls == 1 && resets == 0 && Snapshot::take().moved_by(s0, 1, 0, 0, 0), "OVS_ERR_CAPACITY, nothing to grow: std::length_error");
    {
        int n = 0, grown = 0;
        s0 = Snapshot::take();
        const bool okg = util::run_guarded(
            "grow", [&] { return n++ == 0 ? OVS_ERR_CAPACITY : OVS_OK; }, [&] {}, [&] { ++grown; return true; });
        expect(okg && n == 2 && grown == 1 && Snapshot::take().moved_by(s0, 1, 1, 1, 0), "OVS_ERR_CAPACITY with a context that can grow: enlarged once, retried, ok");
    }
    threw = false;
    s0 = Snapshot::take();
    try {
        scripted({OVS_ERR_INVALID});
    } catch (const std::invalid_argument&) {
        threw = true;
    }
    expect(threw && calls == 1 && Snapshot::take().moved_by(s0, 1, 0, 0, 0) && util::device_failures().surfaced.load() >= 2 &&
               util::device_failures().by_status[1].load() >= 1,
           "OVS_ERR_INVALID: std::invalid_argument, counted per status");
    // a context holder that cannot build its handle throws device_error inside the guarded call: same policy, by its status
    int thrown = 0;
    s0 = Snapshot::take();
    bool ok = util::run_guarded(
        "throwing", [&]() -> int { if (thrown++ == 0) throw util::device_error(OVS_ERR_HIP, "handle lost"); return OVS_OK; }, [&] {});
    expect(ok && thrown == 2 && Snapshot::take().moved_by(s0, 1, 1, 1, 0), "device_error(OVS_ERR_HIP) from a context holder: retried like the status");
    s0 = Snapshot::take();
    bool escaped = false;
    try {
        ok = util::run_guarded("throwing", [&]() -> int { throw util::device_error(OVS_ERR_NO_DEVICE, "no device"); }, [&] {});
    } catch (...) {
        escaped = true;
    }
    expect(!ok && !escaped && Snapshot::take().moved_by(s0, 1, 0, 0, 1), "device_error(OVS_ERR_NO_DEVICE): caught, the empty result");
}

static void check_classes_without_a_device() {
    if (ovs_device_count() > 0) {
        std::printf("skip a HIP device is present: the no-device behaviour of the classes is not exercised (test_fault_shim injects failures instead)\n");
        return;
    }
    bool escaped = false;
    try {
        const int rows = 120, cols = 160;
        cv::Mat img(rows, cols, cv::CV_8UC1);
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x) img.ptr(y)[x] = (uint8_t)((x * 7 + y * 13) ^ (x * y));
        feature::orb_extractor ex(feature::orb_params(500, 1.2f, 8, 20, 7));
        std::vector<cv::KeyPoint> kps(3);
        cv::Mat desc;
        const Snapshot s0 = Snapshot::take();
        ex.extract(img, cv::Mat(), kps, desc);
        expect(kps.empty() && desc.rows == 0 && Snapshot::take().degraded > s0.degraded, "orb_extractor::extract: no keypoints, no exception");
        expect(ex.get_scale_factors().size() == 8 && ex.get_scale_factors()[1] == 1.2f, "orb_extractor tables need no device");
        // two hand-made frames with keypoints and descriptors: the matchers must answer 0
        data::frame fa, fb;
        camera::base cam;
        cam.cols_ = cols;
        cam.rows_ = rows;
        cam.fx_ = cam.fy_ = 100.0;
        cam.cx_ = cols / 2.0;
        cam.cy_ = rows / 2.0;
        cam.img_bounds_.max_x_ = (float)cols;
        cam.img_bounds_.max_y_ = (float)rows;
        const unsigned n = 64;
        for (data::frame* f : {&fa, &fb}) {
            f->keypts_.resize(n);
            f->descriptors_.create((int)n, 32, cv::CV_8U);
            for (unsigned i = 0; i < n; ++i) {
                f->keypts_[i] = cv::KeyPoint();
                f->keypts_[i].pt.x = (float)(10 + (i * 37) % 140);
                f->keypts_[i].pt.y = (float)(10 + (i * 53) % 100);
                f->keypts_[i].octave = 0;
                for (int b = 0; b < 32; ++b) f->descriptors_.ptr((int)i)[b] = (uint8_t)(i * 31 + b * 17);
            }
            f->num_keypts_ = n;
            f->undist_keypts_ = f->keypts_;
            f->camera_ = &cam;
            f->scale_factors_ = ex.get_scale_factors();
            f->inv_level_sigma_sq_ = ex.get_inv_level_sigma_sq();
        }
        data::keyframe kf;
        kf.keypts_ = fb.keypts_;
        kf.descriptors_ = fb.descriptors_;
        kf.num_keypts_ = n;
        std::vector<std::unique_ptr<data::landmark>> own;
        kf.landmarks_.assign(n, nullptr);
        for (unsigned i = 0; i < n; ++i) {
            own.emplace_back(new data::landmark());
            kf.landmarks_[i] = own.back().get();
        }
        std::vector<std::pair<int, int>> matches(5);
        expect(match::robust(0.9f, false).brute_force_match(fa, &kf, matches) == 0 && matches.empty(), "robust::brute_force_match: 0 matches, no exception");
        std::vector<cv::Point2f> prev(n);
        for (unsigned i = 0; i < n; ++i) prev[i] = fa.undist_keypts_[i].pt;
        const std::vector<cv::Point2f> prev_in = prev;
        std::vector<int> idx;
        const unsigned n_area = match::area(0.9f, true).match_in_consistent_area(fa, fb, prev, idx, 50);
        bool untouched = idx.size() == n && std::memcmp(prev.data(), prev_in.data(), n * sizeof(cv::Point2f)) == 0;
        for (int v : idx) untouched = untouched && v == -1;
        expect(n_area == 0 && untouched, "area::match_in_consistent_area: 0 matches, every index -1, prev_matched_pts untouched");
        // pose optimiser: zero inliers, the pose stays what it was
        fa.landmarks_.assign(n, nullptr);
        for (unsigned i = 0; i < n; ++i) {
            Vec3_t p;
            p(0) = (fa.keypts_[i].pt.x - cam.cx_) / cam.fx_ * 4.0;
            p(1) = (fa.keypts_[i].pt.y - cam.cy_) / cam.fy_ * 4.0;
            p(2) = 4.0;
            own[i]->set_pos_in_world(p);
            fa.landmarks_[i] = own[i].get();
        }
        Mat44_t T;
        T(0, 3) = 0.02;
        fa.set_cam_pose(T);
        const unsigned n_pose = optimize::pose_optimizer().optimize(fa);
        expect(n_pose == 0 && fa.cam_pose_cw_(0, 3) == 0.02, "pose_optimizer::optimize: 0 inliers, pose untouched");
    } catch (const std::exception& e) {
        std::printf("     exception: %s\n", e.what());
        escaped = true;
    } catch (...) {
        escaped = true;
    }
    expect(!escaped, "no exception left a class");
}

int main() {
    check_run_guarded();
    check_classes_without_a_device();
    std::printf("%s\n", g_bad ? "FAILED" : "ALL OK");
    return g_bad ? 1 : 0;
}
