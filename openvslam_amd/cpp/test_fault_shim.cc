// The failure policy of the class shims (openvslam/util/device_policy.h) under injected HIP failures (ovs_debug_inject_hip_failures):
// a single failed call is retried on rebuilt contexts and the caller sees the normal result; a device that keeps failing yields the empty
// result of every function -- no exception reaches the caller --, and once the device answers again so do the classes.
// usage: test_fault_shim rows cols nfeat frame_a.raw frame_b.raw      (prints one line per check; exit code 0 = all held)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "openvslam/feature/orb_extractor.h"
#include "openvslam/match/area.h"
#include "openvslam/match/projection.h"
#include "openvslam/match/robust.h"
#include "openvslam/optimize/pose_optimizer.h"
#include "openvslam/util/device_policy.h"

using namespace openvslam;

static cv::Mat read_raw(const char* path, int rows, int cols) {
    cv::Mat m(rows, cols, cv::CV_8UC1);
    FILE* f = std::fopen(path, "rb");
    if (!f || std::fread(m.data, 1, (size_t)rows * cols, f) != (size_t)rows * cols) {
        std::fprintf(stderr, "cannot read %s\n", path);
        std::exit(2);
    }
    std::fclose(f);
    return m;
}

static int g_bad = 0;
static void expect(bool ok, const char* what) {
    std::printf("%s %s\n", ok ? "ok  " : "FAIL", what);
    if (!ok) ++g_bad;
}

struct Scene {
    feature::orb_extractor extractor;
    camera::base cam;
    data::frame frm_a, frm_b;
    data::keyframe keyfrm;
    std::vector<std::unique_ptr<data::landmark>> own;
    std::vector<data::landmark*> local_lms;
    explicit Scene(int nfeat) : extractor(feature::orb_params(nfeat, 1.2f, 8, 20, 7)) {}
};

struct Results {
    std::vector<cv::KeyPoint> kps;
    std::vector<uint8_t> desc;
    unsigned n_bf = 0, n_area = 0, n_proj = 0, n_pose = 0;
    std::vector<int> area_idx;
    bool operator==(const Results& o) const {
        return kps.size() == o.kps.size() && (kps.empty() || std::memcmp(kps.data(), o.kps.data(), kps.size() * sizeof(cv::KeyPoint)) == 0) &&
               desc == o.desc && n_bf == o.n_bf && n_area == o.n_area && n_proj == o.n_proj && n_pose == o.n_pose && area_idx == o.area_idx;
    }
};

// one tracked frame through the classes, as test_shim.cc drives them; every call may be hit by an injected failure
static Results run_all(Scene& s, const cv::Mat& a, const cv::Mat& b, int cols, int rows) {
    Results r;
    data::frame& fa = s.frm_a;
    data::frame& fb = s.frm_b;
    fa = data::frame();
    fb = data::frame();
    s.extractor.extract(a, cv::Mat(), fa.keypts_, fa.descriptors_);
    fa.num_keypts_ = fa.keypts_.size();
    s.extractor.extract(b, cv::Mat(), fb.keypts_, fb.descriptors_);
    fb.num_keypts_ = fb.keypts_.size();
    r.kps = fa.keypts_;
    r.desc.assign(fa.descriptors_.data, fa.descriptors_.data + (size_t)32 * fa.num_keypts_);
    s.cam.cols_ = cols;
    s.cam.rows_ = rows;
    s.cam.fx_ = s.cam.fy_ = 500.0;
    s.cam.cx_ = cols / 2.0;
    s.cam.cy_ = rows / 2.0;
    s.cam.img_bounds_.max_x_ = (float)cols;
    s.cam.img_bounds_.max_y_ = (float)rows;
    fa.undist_keypts_ = fa.keypts_;
    fb.undist_keypts_ = fb.keypts_;
    fa.camera_ = fb.camera_ = &s.cam;
    fa.scale_factors_ = fb.scale_factors_ = s.extractor.get_scale_factors();
    fa.inv_level_sigma_sq_ = fb.inv_level_sigma_sq_ = s.extractor.get_inv_level_sigma_sq();
    // robust::brute_force_match against a keyframe holding frame b
    s.keyfrm = data::keyframe();
    s.keyfrm.keypts_ = fb.keypts_;
    s.keyfrm.descriptors_ = fb.descriptors_;
    s.keyfrm.num_keypts_ = fb.num_keypts_;
    s.own.clear();
    s.keyfrm.landmarks_.assign(s.keyfrm.num_keypts_, nullptr);
    for (unsigned i = 0; i < s.keyfrm.num_keypts_; ++i) {
        s.own.emplace_back(new data::landmark());
        s.keyfrm.landmarks_[i] = s.own.back().get();
    }
    std::vector<std::pair<int, int>> matches;
    r.n_bf = match::robust(0.9f, false).brute_force_match(fa, &s.keyfrm, matches);
    // area
    std::vector<cv::Point2f> prev(fa.num_keypts_);
    for (unsigned i = 0; i < fa.num_keypts_; ++i) prev[i] = fa.undist_keypts_[i].pt;
    r.n_area = match::area(0.9f, true).match_in_consistent_area(fa, fb, prev, r.area_idx, 100);
    // projection::match_frame_and_landmarks: frame a's keypoints as landmarks reprojected at their own position
    s.local_lms.clear();
    const size_t first_local = s.own.size();
    for (unsigned i = 0; i < fa.num_keypts_; ++i) {
        s.own.emplace_back(new data::landmark());
        auto* lm = s.own.back().get();
        lm->descriptor_ = fa.descriptors_.row((int)i);
        lm->reproj_in_tracking_(0) = fa.keypts_[i].pt.x;
        lm->reproj_in_tracking_(1) = fa.keypts_[i].pt.y;
        lm->is_observable_in_tracking_ = true;
        lm->scale_level_in_tracking_ = fa.keypts_[i].octave;
        // a world position that projects onto the keypoint at depth 4 (identity pose): the pose optimiser's input
        Vec3_t p;
        p(0) = (fa.keypts_[i].pt.x - s.cam.cx_) / s.cam.fx_ * 4.0;
        p(1) = (fa.keypts_[i].pt.y - s.cam.cy_) / s.cam.fy_ * 4.0;
        p(2) = 4.0;
        lm->set_pos_in_world(p);
        s.local_lms.push_back(lm);
    }
    (void)first_local;
    fb.landmarks_.assign(fb.num_keypts_, nullptr);
    r.n_proj = match::projection(0.8f, true).match_frame_and_landmarks(fb, s.local_lms, 8.0f);
    // pose_optimizer on frame a with its own landmarks, from a slightly wrong pose
    fa.landmarks_.assign(fa.num_keypts_, nullptr);
    for (unsigned i = 0; i < fa.num_keypts_; ++i) fa.landmarks_[i] = s.local_lms[i];
    Mat44_t T;   // (the stand-in default-constructs to the identity)
    T(0, 3) = 0.02;
    T(1, 3) = -0.01;
    fa.set_cam_pose(T);
    r.n_pose = optimize::pose_optimizer().optimize(fa);
    return r;
}

int main(int argc, char** argv) {
    if (argc != 6) return 2;
    const int rows = std::atoi(argv[1]), cols = std::atoi(argv[2]), nfeat = std::atoi(argv[3]);
    const cv::Mat a = read_raw(argv[4], rows, cols), b = read_raw(argv[5], rows, cols);
    Scene s(nfeat);
    auto& c = util::device_failures();
    try {
        const Results base = run_all(s, a, b, cols, rows);
        std::printf("base: %zu keypoints, brute force %u, area %u, projection %u, pose inliers %u\n", base.kps.size(), base.n_bf, base.n_area, base.n_proj,
                    base.n_pose);
        expect(base.kps.size() > 100 && base.n_bf > 20 && base.n_area > 20 && base.n_proj > 20 && base.n_pose > 20, "the scene exercises every class");
        expect(c.failed_calls == 0 && c.degraded == 0, "no failure without injection");
        // ---- a single failed HIP call, at different depths into the frame: retried, the caller sees the normal result
        for (int skip : {0, 3, 9, 20, 45, 80, 110, 140, 170, 200}) {
            const unsigned long f0 = c.failed_calls, r0 = c.recovered, d0 = c.degraded;
            ovs_debug_inject_hip_failures(skip, 1);   // `skip` checked HIP calls pass, the next one reports a failure
            const Results again = run_all(s, a, b, cols, rows);
            ovs_debug_inject_hip_failures(0, 0);
            expect(again == base, "one injected failure: results identical to the undisturbed run");
            if (c.failed_calls == f0) {   // the frame makes fewer checked HIP calls than `skip`: nothing was hit
                std::printf("note skip %d is beyond the frame's HIP calls\n", skip);
                continue;
            }
            expect(c.failed_calls == f0 + 1 && c.recovered == r0 + 1 && c.degraded == d0, "one failure counted, one recovery, nothing degraded");
        }
        // ---- a device that keeps failing: every class returns its empty result, nothing throws
        const unsigned long d0 = c.degraded;
        ovs_debug_inject_hip_failures(0, 1 << 30);
        const Results dead = run_all(s, a, b, cols, rows);
        ovs_debug_inject_hip_failures(0, 0);
        expect(dead.kps.empty() && dead.desc.empty(), "dead device: extract() returns no keypoints");
        expect(dead.n_bf == 0 && dead.n_area == 0 && dead.n_proj == 0 && dead.n_pose == 0, "dead device: every matcher and the pose optimiser return 0");
        expect(c.degraded >= d0 + 2, "the degraded calls were counted");
        // with keypoints in hand but the device failing inside the matchers: zero matches, outputs in their initial state
        {
            Results probe = run_all(s, a, b, cols, rows);   // healthy: fills s.frm_a / s.frm_b
            expect(probe == base, "the device answers again: results identical to the first run");
            ovs_debug_inject_hip_failures(0, 1 << 30);
            std::vector<cv::Point2f> prev(s.frm_a.num_keypts_);
            for (unsigned i = 0; i < s.frm_a.num_keypts_; ++i) prev[i] = s.frm_a.undist_keypts_[i].pt;
            const std::vector<cv::Point2f> prev0 = prev;
            std::vector<int> idx;
            const unsigned n_area = match::area(0.9f, true).match_in_consistent_area(s.frm_a, s.frm_b, prev, idx, 100);
            bool untouched = n_area == 0 && idx.size() == s.frm_a.num_keypts_;
            for (int v : idx) untouched = untouched && v == -1;
            for (size_t i = 0; i < prev.size(); ++i) untouched = untouched && prev[i].x == prev0[i].x && prev[i].y == prev0[i].y;
            expect(untouched, "failing area matcher: 0 matches, every index -1, prev_matched_pts unchanged");
            s.frm_b.landmarks_.assign(s.frm_b.num_keypts_, nullptr);
            const unsigned n_proj = match::projection(0.8f, true).match_frame_and_landmarks(s.frm_b, s.local_lms, 8.0f);
            bool none = n_proj == 0;
            for (auto* lm : s.frm_b.landmarks_) none = none && lm == nullptr;
            expect(none, "failing projection matcher: 0 matches, no landmark written into the frame");
            const Mat44_t before = s.frm_a.cam_pose_cw_;
            const unsigned n_pose = optimize::pose_optimizer().optimize(s.frm_a);
            bool same = n_pose == 0;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) same = same && s.frm_a.cam_pose_cw_(i, j) == before(i, j);
            expect(same, "failing pose optimiser: 0 inliers, pose untouched");
            ovs_debug_inject_hip_failures(0, 0);
        }
        const Results healed = run_all(s, a, b, cols, rows);
        expect(healed == base, "after the failures: results identical to the first run");
    } catch (const std::exception& e) {
        ovs_debug_inject_hip_failures(0, 0);
        std::printf("FAIL an exception reached the caller: %s\n", e.what());
        return 1;
    }
    std::printf("failed_calls %lu retried %lu recovered %lu degraded %lu\n", (unsigned long)c.failed_calls, (unsigned long)c.retried,
                (unsigned long)c.recovered, (unsigned long)c.degraded);
    return g_bad ? 1 : 0;
}
