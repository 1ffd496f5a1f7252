// Two threads on the same frame / keyframe objects, as upstream's tracking and mapping threads are:
//   * a keyframe constructed from a frame SHARES the frame's device cache (data/frame_stub.h): nothing is uploaded for it;
//   * thread T (tracking style: projection::match_frame_and_landmarks on a copy of the frame, bow_tree::match_frame_and_keyframe against
//     the keyframe) and thread M (mapping style: fuse::detect_duplication on the keyframe, robust::match_for_triangulation between two
//     keyframes) start together on objects whose caches are still EMPTY, so the first-use creation of the resident handles races every
//     iteration; every iteration's results must equal the single-threaded ones;
//   * with two HIP devices (skipped otherwise) the same calls run with the frames' caches on device 1 and give the same results.
// Built twice: plain (test_threads_shim) and with -fsanitize=thread on the host side (test_threads_shim_tsan).
// usage: test_threads_shim rows cols nfeat frame_a.raw frame_b.raw iterations
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "openvslam/feature/orb_extractor.h"
#include "openvslam/match/bow_tree.h"
#include "openvslam/match/fuse.h"
#include "openvslam/match/projection.h"
#include "openvslam/match/robust.h"
#include "openvslam/match/window_ctx.h"
#include "openvslam/util/device_policy.h"

#include <ovslam_hip.h>

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

struct Scene {
    camera::base cam;
    data::frame fa, fb;                                    // the two extracted frames (caches empty until a matcher asks)
    std::vector<std::unique_ptr<data::landmark>> own;      // every landmark object of the scene
    std::vector<data::landmark*> local_lms;                // one per keypoint of frame a, reprojecting onto it (tracking's local map)
    std::vector<data::landmark*> to_check;                 // copies of the keyframe's landmarks (fuse::detect_duplication's input)
    std::vector<data::landmark*> kf_lms;                   // the keyframe's own landmarks (every keypoint has one)
};

static void fill_frame(data::frame& f, const std::vector<cv::KeyPoint>& kps, const cv::Mat& desc, camera::base* cam, const std::vector<float>& sfs) {
    f.keypts_ = f.undist_keypts_ = kps;
    f.descriptors_ = desc;
    f.num_keypts_ = (unsigned)kps.size();
    f.camera_ = cam;
    f.scale_factors_ = sfs;
    f.inv_level_sigma_sq_.resize(sfs.size());
    for (size_t l = 0; l < sfs.size(); ++l) f.inv_level_sigma_sq_[l] = 1.0f / (sfs[l] * sfs[l]);
    f.log_scale_factor_ = std::log(1.2f);
    f.landmarks_.assign(f.num_keypts_, nullptr);
    f.bearings_.resize(f.num_keypts_);
    for (unsigned i = 0; i < f.num_keypts_; ++i) {
        const double vx = ((double)kps[i].pt.x - cam->cx_) / cam->fx_, vy = ((double)kps[i].pt.y - cam->cy_) / cam->fy_;
        const double nrm = std::sqrt((vx * vx + vy * vy) + 1.0);
        f.bearings_[i](0) = vx / nrm;
        f.bearings_[i](1) = vy / nrm;
        f.bearings_[i](2) = 1.0 / nrm;
        f.bow_feat_vec_[desc.ptr((int)i)[0] & 127u].push_back(i);
    }
}

static data::landmark* make_landmark(Scene& s, const data::frame& f, unsigned i, double z) {
    s.own.emplace_back(new data::landmark());
    data::landmark* lm = s.own.back().get();
    const cv::KeyPoint& kp = f.undist_keypts_[i];
    lm->pos_w_(0) = ((double)kp.pt.x - s.cam.cx_) / s.cam.fx_ * z;
    lm->pos_w_(1) = ((double)kp.pt.y - s.cam.cy_) / s.cam.fy_ * z;
    lm->pos_w_(2) = z;
    const double nrm = std::sqrt((lm->pos_w_(0) * lm->pos_w_(0) + lm->pos_w_(1) * lm->pos_w_(1)) + z * z);
    for (int a = 0; a < 3; ++a) lm->mean_normal_(a) = lm->pos_w_(a) / nrm;
    lm->max_valid_dist_ = (float)(nrm * f.scale_factors_[(size_t)kp.octave] * 0.93);
    lm->min_valid_dist_ = lm->max_valid_dist_ / f.scale_factors_.back();
    lm->descriptor_.create(1, 32, cv::CV_8U);
    std::memcpy(lm->descriptor_.data, f.descriptors_.ptr((int)i), 32);
    lm->reproj_in_tracking_(0) = kp.pt.x;
    lm->reproj_in_tracking_(1) = kp.pt.y;
    lm->is_observable_in_tracking_ = true;
    lm->scale_level_in_tracking_ = kp.octave;
    return lm;
}

struct Counts {
    unsigned proj = 0, bow = 0, dup = 0, tri = 0;
    bool operator==(const Counts& o) const { return proj == o.proj && bow == o.bow && dup == o.dup && tri == o.tri; }
};

// tracking-style calls on a COPY of frame a (shares a's cache) and on the keyframe
static void tracking_calls(const Scene& s, const data::frame& fa_obj, data::keyframe& kf, Counts& c) {
    data::frame f = fa_obj;
    f.landmarks_.assign(f.num_keypts_, nullptr);
    c.proj = match::projection(0.8f, true).match_frame_and_landmarks(f, const_cast<std::vector<data::landmark*>&>(s.local_lms), 5.0f);
    std::vector<data::landmark*> matched;
    data::frame g = fa_obj;
    c.bow = match::bow_tree(0.75f, true).match_frame_and_keyframe(&kf, g, matched);
}
// mapping-style calls on the keyframe(s)
static void mapping_calls(const Scene& s, data::keyframe& kf, data::keyframe& kt1, data::keyframe& kt2, Counts& c) {
    std::vector<data::landmark*> dups;
    Mat44_t I;
    c.dup = match::fuse(0.6f).detect_duplication(&kf, I, s.to_check, 4.0f, dups);
    Mat33_t E_12;   // [t_12]x with R_12 = I, t_12 = (0.2, 0.15, 0)
    E_12(0, 0) = 0, E_12(0, 1) = 0, E_12(0, 2) = 0.15;
    E_12(1, 0) = 0, E_12(1, 1) = 0, E_12(1, 2) = -0.2;
    E_12(2, 0) = -0.15, E_12(2, 1) = 0.2, E_12(2, 2) = 0;
    std::vector<std::pair<unsigned int, unsigned int>> pairs;
    c.tri = match::robust(0.6f, true).match_for_triangulation(&kt1, &kt2, E_12, pairs);
}

// fresh objects with EMPTY caches on `device`: keyframe kf made from frame b (all keypoints own a landmark), triangulation keyframes from a and b
struct Objects {
    data::frame fa, fb;
    std::unique_ptr<data::keyframe> kf, kt1, kt2;
    Objects(const Scene& s, int device) : fa(s.fa), fb(s.fb) {
        fa.device_cache_ = std::make_shared<data::frame_device_cache>();   // (a copy would share the scene's cache)
        fb.device_cache_ = std::make_shared<data::frame_device_cache>();
        fa.device_cache_->device = fb.device_cache_->device = device;
        kf.reset(new data::keyframe(fb));
        kf->landmarks_ = s.kf_lms;
        kt1.reset(new data::keyframe(fa));
        kt2.reset(new data::keyframe(fb));
        kt1->landmarks_.assign(kt1->num_keypts_, nullptr);
        kt2->landmarks_.assign(kt2->num_keypts_, nullptr);
        kt2->cam_pose_cw_(0, 3) = -0.2;
        kt2->cam_pose_cw_(1, 3) = -0.15;
    }
};

int main(int argc, char** argv) {
    if (argc != 7) return 2;
    const int rows = std::atoi(argv[1]), cols = std::atoi(argv[2]), nfeat = std::atoi(argv[3]), iters = std::atoi(argv[6]);
    const cv::Mat a = read_raw(argv[4], rows, cols), b = read_raw(argv[5], rows, cols);
    feature::orb_extractor extractor(feature::orb_params(nfeat, 1.2f, 8, 20, 7));
    Scene s;
    s.cam.cols_ = cols;
    s.cam.rows_ = rows;
    s.cam.fx_ = s.cam.fy_ = 0.6 * cols;
    s.cam.cx_ = cols / 2.0;
    s.cam.cy_ = rows / 2.0;
    s.cam.img_bounds_.max_x_ = (float)cols;
    s.cam.img_bounds_.max_y_ = (float)rows;
    std::vector<cv::KeyPoint> ka, kb;
    cv::Mat da, db;
    extractor.extract(a, cv::Mat(), ka, da);
    extractor.extract(b, cv::Mat(), kb, db);
    const std::vector<float> sfs = extractor.get_scale_factors();
    fill_frame(s.fa, ka, da, &s.cam, sfs);
    fill_frame(s.fb, kb, db, &s.cam, sfs);
    for (unsigned i = 0; i < s.fa.num_keypts_; ++i) s.local_lms.push_back(make_landmark(s, s.fa, i, 4.0 + (i % 7)));
    for (unsigned i = 0; i < s.fb.num_keypts_; ++i) {
        s.kf_lms.push_back(make_landmark(s, s.fb, i, 5.0 + (i % 5)));
        if (i % 3 != 0) s.to_check.push_back(make_landmark(s, s.fb, i, 5.0 + (i % 5)));   // same place, same descriptor: a duplicate
    }
    int bad = 0;
    auto expect = [&](bool ok, const char* what) {
        std::printf("%s %s\n", ok ? "ok  " : "FAIL", what);
        if (!ok) ++bad;
    };
    // ---- single-threaded reference, and the cache sharing between a frame, its copies and the keyframe made from it
    Counts ref;
    {
        Objects o(s, 0);
        expect(o.kf->device_cache_.get() == o.fb.device_cache_.get() && !o.fb.device_cache_->resident(), "a keyframe shares its frame's (still empty) device cache");
        tracking_calls(s, o.fa, *o.kf, ref);
        expect(o.fa.device_cache_->resident() && o.fb.device_cache_->resident(), "the first matcher call made frame and keyframe resident");
        mapping_calls(s, *o.kf, *o.kt1, *o.kt2, ref);
        expect(o.kt1->device_cache_->has_bearings && o.kt2->device_cache_->has_bearings, "match_for_triangulation attached the bearings to the shared handles");
        Counts again;
        tracking_calls(s, o.fa, *o.kf, again);
        mapping_calls(s, *o.kf, *o.kt1, *o.kt2, again);
        expect(again == ref, "resident handles give the same results on reuse");
        std::printf("reference: match_frame_and_landmarks %u, bow %u, detect_duplication %u, match_for_triangulation %u\n", ref.proj, ref.bow, ref.dup, ref.tri);
        expect(ref.proj > s.fa.num_keypts_ / 2 && ref.bow > 20 && ref.dup > s.to_check.size() / 2 && ref.tri > 20, "the scene exercises every matcher");
    }
    // ---- two threads, first use racing on empty caches every iteration
    std::atomic<int> mismatches{0};
    for (int it = 0; it < iters; ++it) {
        Objects o(s, 0);
        std::atomic<int> go{0};
        Counts ct, cm;
        std::thread T([&] {
            ++go;
            while (go.load() < 2) {}
            tracking_calls(s, o.fa, *o.kf, ct);
        });
        std::thread M([&] {
            ++go;
            while (go.load() < 2) {}
            mapping_calls(s, *o.kf, *o.kt1, *o.kt2, cm);
        });
        T.join();
        M.join();
        if (ct.proj != ref.proj || ct.bow != ref.bow || cm.dup != ref.dup || cm.tri != ref.tri) ++mismatches;
    }
    char line[160];
    std::snprintf(line, sizeof(line), "%d iterations of (tracking thread || mapping thread) on shared caches: every result equals the single-threaded one", iters);
    expect(mismatches.load() == 0, line);
    // ---- camera::fisheye / radial_division: pinhole reprojection on undistorted keypoints, i.e. the perspective results (ORACLE_SPEC rule 31)
    for (const auto model : {camera::model_type_t::Fisheye, camera::model_type_t::RadialDivision}) {
        s.cam.model_type_ = model;
        Objects o(s, 0);
        Counts c;
        tracking_calls(s, o.fa, *o.kf, c);
        mapping_calls(s, *o.kf, *o.kt1, *o.kt2, c);
        expect(c == ref, model == camera::model_type_t::Fisheye ? "a fisheye camera gives the perspective results (no refusal)" : "a radial-division camera gives the perspective results");
    }
    s.cam.model_type_ = camera::model_type_t::Perspective;
    // ---- two-object matchers whose second object is resident on ANOTHER device (frame i -> GPU i mod G): the shim builds a temporary handle on
    // the call's device instead of handing the ABI a foreign one (which it refuses with OVS_ERR_INVALID). On a one-GPU box the hook forces that
    // path for every two-object call; with two devices the frames really live on different ones.
    {
        Objects o(s, 0);
        Counts c;
        match::detail::force_foreign_handles() = true;
        const int before = match::detail::foreign_handles_built().load();
        tracking_calls(s, o.fa, *o.kf, c);
        mapping_calls(s, *o.kf, *o.kt1, *o.kt2, c);
        match::detail::force_foreign_handles() = false;
        expect(c == ref && match::detail::foreign_handles_built().load() >= before + 2, "temporary handles for the second object (forced): same results");
    }
    if (ovs_device_count() >= 2) {
        Objects o(s, 0);
        o.fb.device_cache_->device = 1;   // frame b, and the keyframes made from it (shared cache), live on device 1; frame a on device 0
        Counts c;
        const int before = match::detail::foreign_handles_built().load();
        tracking_calls(s, o.fa, *o.kf, c);
        mapping_calls(s, *o.kf, *o.kt1, *o.kt2, c);
        expect(c == ref && match::detail::foreign_handles_built().load() > before, "frame on device 0, keyframe on device 1: same results, no exception");
    } else {
        std::printf("skip mixed devices: ovs_device_count() = %d\n", ovs_device_count());
    }
    // ---- device 1
    if (ovs_device_count() >= 2) {
        Objects o(s, 1);
        Counts c1;
        tracking_calls(s, o.fa, *o.kf, c1);
        mapping_calls(s, *o.kf, *o.kt1, *o.kt2, c1);
        expect(c1 == ref, "the same calls with the frames' caches on device 1: same results");
    } else {
        std::printf("skip device 1: ovs_device_count() = %d\n", ovs_device_count());
    }
    const auto& fc = util::device_failures();
    expect(fc.failed_calls.load() == 0, "no ABI call failed");
    std::printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad ? 1 : 0;
}
