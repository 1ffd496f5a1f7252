// Per-call latency of feature::orb_extractor::extract THROUGH THE CLASS BOUNDARY: caller-owned pageable image in,
// std::vector<cv::KeyPoint> + cv::Mat out, H2D / D2H included. Prints one JSON object.
// usage: bench_shim rows cols nfeat frame_a.raw frame_b.raw iters
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <ovslam_hip.h>

#include "openvslam/feature/orb_extractor.h"
#include "openvslam/match/area.h"
#include "openvslam/match/fuse.h"
#include "openvslam/match/projection.h"
#include "openvslam/optimize/pose_optimizer.h"

using namespace openvslam;
using clk = std::chrono::steady_clock;

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

struct Stats {
    double mean = 0, median = 0, p95 = 0, min = 0;
};
static Stats stats(std::vector<double> v) {
    Stats s;
    if (v.empty()) return s;
    std::sort(v.begin(), v.end());
    for (double x : v) s.mean += x;
    s.mean /= (double)v.size();
    s.median = v[v.size() / 2];
    s.p95 = v[std::min(v.size() - 1, (size_t)(0.95 * (double)v.size()))];
    s.min = v[0];
    return s;
}

// `iters` timed extract() calls alternating between two images (fresh pixels every call), after 5 warm-up calls
static Stats run(feature::orb_extractor& ex, const cv::Mat& a, const cv::Mat& b, int iters, double* split_ms /*h2d, kernels, d2h*/, int* nkp) {
    std::vector<cv::KeyPoint> kps;
    cv::Mat desc;
    for (int i = 0; i < 5; ++i) ex.extract(i & 1 ? b : a, cv::Mat(), kps, desc);
    ovs_orb_profile_enable(const_cast<ovs_orb*>(ex.handle()), 1);
    std::vector<double> ms;
    double acc[3] = {0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        const auto t0 = clk::now();
        ex.extract(i & 1 ? b : a, cv::Mat(), kps, desc);
        ms.push_back(std::chrono::duration<double, std::milli>(clk::now() - t0).count());
        float s3[3];
        ovs_orb_host_profile_read(ex.handle(), s3);
        for (int k = 0; k < 3; ++k) acc[k] += s3[k];
    }
    ovs_orb_profile_enable(const_cast<ovs_orb*>(ex.handle()), 0);
    for (int k = 0; k < 3; ++k) split_ms[k] = acc[k] / iters;
    *nkp = (int)kps.size();
    return stats(ms);
}

// page-locked host memory straight from the HIP runtime (this program is plain g++: no HIP headers)
extern "C" int hipHostMalloc(void** ptr, size_t size, unsigned int flags);
extern "C" int hipHostFree(void* ptr);

int main(int argc, char** argv) {
    if (argc != 7) return 2;
    const int rows = std::atoi(argv[1]), cols = std::atoi(argv[2]), nfeat = std::atoi(argv[3]), iters = std::atoi(argv[6]);
    const cv::Mat a = read_raw(argv[4], rows, cols), b = read_raw(argv[5], rows, cols);
    std::printf("{\"rows\": %d, \"cols\": %d, \"max_num_keypts\": %d, \"iters\": %d", rows, cols, nfeat, iters);
    const char* names[4] = {"staged_no_pyramid", "staged_with_pyramid", "pageable_no_pyramid", "pageable_with_pyramid"};
    for (int cfg = 0; cfg < 4; ++cfg) {
        feature::orb_extractor ex(feature::orb_params(nfeat, 1.2f, 8, 20, 7));
        ex.set_image_pyramid_download((cfg & 1) != 0);
        std::vector<cv::KeyPoint> kps;
        cv::Mat desc;
        ex.extract(a, cv::Mat(), kps, desc);   // creates the handle
        ovs_orb_set_host_mode(const_cast<ovs_orb*>(ex.handle()), cfg < 2 ? 1 : 0);
        double split[3];
        int nkp = 0;
        const Stats s = run(ex, a, b, iters, split, &nkp);
        std::printf(", \"%s\": {\"mean_ms\": %.4f, \"median_ms\": %.4f, \"p95_ms\": %.4f, \"min_ms\": %.4f, \"h2d_ms\": %.4f, \"kernels_ms\": %.4f, \"d2h_ms\": %.4f, \"keypoints\": %d}",
                    names[cfg], s.mean, s.median, s.p95, s.min, split[0], split[1], split[2], nkp);
    }
    // two extractors on two threads (upstream: stereo left / right std::threads): neither may serialise the other
    {
        feature::orb_extractor ex_l(feature::orb_params(nfeat, 1.2f, 8, 20, 7)), ex_r(feature::orb_params(nfeat, 1.2f, 8, 20, 7));
        ex_l.set_image_pyramid_download(false);
        ex_r.set_image_pyramid_download(false);
        std::vector<cv::KeyPoint> k1, k2;
        cv::Mat d1, d2;
        for (int i = 0; i < 5; ++i) {
            ex_l.extract(a, cv::Mat(), k1, d1);
            ex_r.extract(b, cv::Mat(), k2, d2);
        }
        auto loop = [&](feature::orb_extractor& ex, const cv::Mat& img, std::vector<cv::KeyPoint>& k, cv::Mat& d, double* per_call) {
            const auto t0 = clk::now();
            for (int i = 0; i < iters; ++i) ex.extract(img, cv::Mat(), k, d);
            *per_call = std::chrono::duration<double, std::milli>(clk::now() - t0).count() / iters;
        };
        double solo = 0, l = 0, r = 0;
        loop(ex_l, a, k1, d1, &solo);
        std::thread tl([&] { loop(ex_l, a, k1, d1, &l); });
        std::thread tr([&] { loop(ex_r, b, k2, d2, &r); });
        tl.join();
        tr.join();
        std::printf(", \"two_threads\": {\"solo_ms\": %.4f, \"left_ms\": %.4f, \"right_ms\": %.4f", solo, l, r);
        // the same with both images in page-locked memory (hipHostMalloc; a capture buffer registered once with hipHostRegister behaves the
        // same): a copy from pageable memory pins its pages per call, and two threads doing that in one process queue on the process's
        // memory-map lock -- most of what the right / left extractors lose to each other above
        void *pa = nullptr, *pb = nullptr;
        const size_t bytes = (size_t)rows * cols;
        if (hipHostMalloc(&pa, bytes, 0) == 0 && hipHostMalloc(&pb, bytes, 0) == 0) {
            std::memcpy(pa, a.data, bytes);
            std::memcpy(pb, b.data, bytes);
            const cv::Mat ap(rows, cols, cv::CV_8U, pa, (size_t)cols), bp(rows, cols, cv::CV_8U, pb, (size_t)cols);
            double solo_p = 0, lp = 0, rp = 0;
            loop(ex_l, ap, k1, d1, &solo_p);
            std::thread t1([&] { loop(ex_l, ap, k1, d1, &lp); });
            std::thread t2([&] { loop(ex_r, bp, k2, d2, &rp); });
            t1.join();
            t2.join();
            std::printf(", \"pinned_input\": {\"solo_ms\": %.4f, \"left_ms\": %.4f, \"right_ms\": %.4f}", solo_p, lp, rp);
        }
        if (pa) hipHostFree(pa);
        if (pb) hipHostFree(pb);
        std::printf("}");
    }
    // ---- one tracked frame through the classes, as tracking_module::track() strings them together (what matters is that the
    // frame's keypoints / descriptors / grid go up once and every matcher after the first finds them in HBM):
    //   orb_extractor::extract -> projection::match_current_and_last_frames (first matcher on the frame: creates the device cache)
    //   -> pose_optimizer::optimize -> projection::match_frame_and_landmarks against the local map (frame already resident)
    {
        feature::orb_extractor ex(feature::orb_params(nfeat, 1.2f, 8, 20, 7));
        ex.set_image_pyramid_download(false);
        camera::base pcam;
        pcam.cols_ = cols;
        pcam.rows_ = rows;
        pcam.img_bounds_.max_x_ = (float)cols;
        pcam.img_bounds_.max_y_ = (float)rows;
        pcam.fx_ = pcam.fy_ = 500.0;
        pcam.cx_ = cols / 2.0;
        pcam.cy_ = rows / 2.0;
        data::frame last;
        ex.extract(a, cv::Mat(), last.keypts_, last.descriptors_);
        last.num_keypts_ = last.keypts_.size();
        last.undist_keypts_ = last.keypts_;
        last.camera_ = &pcam;
        last.scale_factors_ = ex.get_scale_factors();
        std::vector<std::unique_ptr<data::landmark>> own;
        last.landmarks_.assign(last.num_keypts_, nullptr);
        last.outlier_flags_.assign(last.num_keypts_, false);
        std::vector<data::landmark*> local_lms;
        for (unsigned i = 0; i < last.num_keypts_; ++i) {   // landmarks where the next frame (image b = a shifted by (5, 0)) sees them
            own.emplace_back(new data::landmark());
            auto* lm = own.back().get();
            const double z = 2.0 + (double)(i % 7);
            lm->pos_w_(0) = (((double)last.undist_keypts_[i].pt.x - 5.0) - pcam.cx_) / pcam.fx_ * z;
            lm->pos_w_(1) = ((double)last.undist_keypts_[i].pt.y - pcam.cy_) / pcam.fy_ * z;
            lm->pos_w_(2) = z;
            lm->descriptor_ = last.descriptors_.row((int)i);
            lm->reproj_in_tracking_(0) = last.undist_keypts_[i].pt.x - 5.0;
            lm->reproj_in_tracking_(1) = last.undist_keypts_[i].pt.y;
            lm->is_observable_in_tracking_ = true;
            lm->scale_level_in_tracking_ = last.undist_keypts_[i].octave;
            last.landmarks_[i] = lm;
            local_lms.push_back(lm);
        }
        std::vector<double> t_ext, t_cl, t_pose, t_lm, t_all;
        unsigned n_cl = 0, n_pose = 0, n_lm = 0;
        for (int i = 0; i < iters + 5; ++i) {
            const auto t0 = clk::now();
            data::frame curr;
            ex.extract(b, cv::Mat(), curr.keypts_, curr.descriptors_);
            curr.num_keypts_ = curr.keypts_.size();
            curr.undist_keypts_ = curr.keypts_;
            curr.camera_ = &pcam;
            curr.scale_factors_ = last.scale_factors_;
            curr.inv_level_sigma_sq_.resize(curr.scale_factors_.size());
            for (size_t l = 0; l < curr.scale_factors_.size(); ++l) curr.inv_level_sigma_sq_[l] = 1.0f / (curr.scale_factors_[l] * curr.scale_factors_[l]);
            curr.landmarks_.assign(curr.num_keypts_, nullptr);
            curr.cam_pose_cw_(0, 3) = 0.01;   // motion-model guess, slightly off
            const auto t1 = clk::now();
            n_cl = match::projection(0.9f, true).match_current_and_last_frames(curr, last, 15.0f);
            const auto t2 = clk::now();
            n_pose = optimize::pose_optimizer().optimize(curr);
            const auto t3 = clk::now();
            for (auto& l : curr.landmarks_) l = nullptr;   // (search_local_landmarks looks for the landmarks not yet tracked)
            n_lm = match::projection(0.8f, true).match_frame_and_landmarks(curr, local_lms, 5.0f);
            const auto t4 = clk::now();
            if (i < 5) continue;
            auto ms = [](clk::time_point x, clk::time_point y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
            t_ext.push_back(ms(t0, t1));
            t_cl.push_back(ms(t1, t2));
            t_pose.push_back(ms(t2, t3));
            t_lm.push_back(ms(t3, t4));
            t_all.push_back(ms(t0, t4));
        }
        // the initializer's matcher on two resident frames
        data::frame f1 = last, f2;
        ex.extract(b, cv::Mat(), f2.keypts_, f2.descriptors_);
        f2.num_keypts_ = f2.keypts_.size();
        f2.undist_keypts_ = f2.keypts_;
        f2.camera_ = &pcam;
        std::vector<double> t_area;
        unsigned n_area = 0;
        for (int i = 0; i < iters + 5; ++i) {
            std::vector<cv::Point2f> prev(f1.num_keypts_);
            for (unsigned k = 0; k < f1.num_keypts_; ++k) prev[k] = f1.undist_keypts_[k].pt;
            std::vector<int> m21;
            const auto t0 = clk::now();
            n_area = match::area(0.9f, true).match_in_consistent_area(f1, f2, prev, m21, 100);
            if (i >= 5) t_area.push_back(std::chrono::duration<double, std::milli>(clk::now() - t0).count());
        }
        std::printf(", \"tracking_per_frame\": {\"extract_median_ms\": %.4f, \"match_current_and_last_frames_median_ms\": %.4f, "
                    "\"pose_optimize_median_ms\": %.4f, \"match_frame_and_landmarks_median_ms\": %.4f, \"frame_total_median_ms\": %.4f, "
                    "\"frame_total_p95_ms\": %.4f, \"area_match_in_consistent_area_median_ms\": %.4f, \"keypoints\": %u, \"matches_cl\": %u, "
                    "\"pose_inliers\": %u, \"matches_local_map\": %u, \"matches_area\": %u, "
                    "\"note\": \"classes with upstream signatures; match_current_and_last_frames includes the once-per-frame upload of the frame "
                    "(ovs_frame_dev), match_frame_and_landmarks finds it resident\"}",
                    stats(t_ext).median, stats(t_cl).median, stats(t_pose).median, stats(t_lm).median, stats(t_all).median, stats(t_all).p95,
                    stats(t_area).median, last.num_keypts_, n_cl, n_pose, n_lm, n_area);
    }
    // ---- mapping side (round 4): fuse::replace_duplication of one new keyframe's landmarks into 20 covisible keyframes, as
    // mapping_module::fuse_landmark_duplication does per new keyframe. The keyframes are long-lived: their handles are resident after the
    // first call, so a call moves the landmarks to check up and the best indices down, nothing else.
    {
        feature::orb_extractor ex(feature::orb_params(nfeat, 1.2f, 8, 20, 7));
        ex.set_image_pyramid_download(false);
        camera::base pcam;
        pcam.cols_ = cols;
        pcam.rows_ = rows;
        pcam.img_bounds_.max_x_ = (float)cols;
        pcam.img_bounds_.max_y_ = (float)rows;
        pcam.fx_ = pcam.fy_ = 500.0;
        pcam.cx_ = cols / 2.0;
        pcam.cy_ = rows / 2.0;
        data::frame src;
        ex.extract(b, cv::Mat(), src.keypts_, src.descriptors_);
        src.num_keypts_ = src.keypts_.size();
        src.undist_keypts_ = src.keypts_;
        src.camera_ = &pcam;
        src.scale_factors_ = ex.get_scale_factors();
        src.inv_level_sigma_sq_.resize(src.scale_factors_.size());
        for (size_t l = 0; l < src.scale_factors_.size(); ++l) src.inv_level_sigma_sq_[l] = 1.0f / (src.scale_factors_[l] * src.scale_factors_[l]);
        src.log_scale_factor_ = std::log(1.2f);
        src.landmarks_.assign(src.num_keypts_, nullptr);
        const int n_kf = 20;
        std::vector<std::unique_ptr<data::keyframe>> kfs;
        for (int k = 0; k < n_kf; ++k) {
            data::frame f = src;
            f.device_cache_ = std::make_shared<data::frame_device_cache>();   // every keyframe its own data (here: equal content)
            kfs.emplace_back(new data::keyframe(f));
            kfs.back()->cam_pose_cw_(0, 3) = 0.002 * k;
        }
        std::vector<std::unique_ptr<data::landmark>> own;
        std::vector<data::landmark*> to_check;
        for (unsigned i = 0; i < src.num_keypts_; ++i) {
            own.emplace_back(new data::landmark());
            auto* lm = own.back().get();
            const double z = 3.0 + (double)(i % 5);
            lm->pos_w_(0) = ((double)src.undist_keypts_[i].pt.x - pcam.cx_) / pcam.fx_ * z;
            lm->pos_w_(1) = ((double)src.undist_keypts_[i].pt.y - pcam.cy_) / pcam.fy_ * z;
            lm->pos_w_(2) = z;
            const double nrm = std::sqrt((lm->pos_w_(0) * lm->pos_w_(0) + lm->pos_w_(1) * lm->pos_w_(1)) + z * z);
            for (int c = 0; c < 3; ++c) lm->mean_normal_(c) = lm->pos_w_(c) / nrm;
            lm->max_valid_dist_ = (float)(nrm * src.scale_factors_[(size_t)src.undist_keypts_[i].octave] * 0.93);
            lm->min_valid_dist_ = lm->max_valid_dist_ / src.scale_factors_.back();
            lm->descriptor_ = src.descriptors_.row((int)i);
            to_check.push_back(lm);
        }
        std::vector<double> t_first, t_res;
        unsigned n_fused = 0;
        for (int i = 0; i < iters / 4 + 3; ++i) {
            for (int k = 0; k < n_kf; ++k) {
                data::keyframe& kf = *kfs[(size_t)k];
                kf.landmarks_.assign(kf.num_keypts_, nullptr);   // undo the previous round's fusions: the same work every time
                for (auto* lm : to_check) {
                    lm->observations_.clear();
                    lm->num_observations_ = 1;
                }
                const auto t0 = clk::now();
                n_fused = match::fuse(0.6f).replace_duplication(&kf, to_check, 3.0f);
                const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
                (i == 0 ? t_first : t_res).push_back(ms);
            }
        }
        std::printf(", \"mapping_fuse\": {\"keyframes\": %d, \"keypoints\": %u, \"landmarks_to_check\": %zu, \"fused_per_call\": %u, "
                    "\"first_call_median_ms\": %.4f, \"resident_call_median_ms\": %.4f, \"resident_call_p95_ms\": %.4f, \"resident_20_keyframes_ms\": %.4f, "
                    "\"note\": \"fuse::replace_duplication through the class, host-side landmark flattening and write-back included; first call = "
                    "upload + grid of the keyframe (ovs_frame_dev), later calls find it resident\"}",
                    n_kf, src.num_keypts_, to_check.size(), n_fused, stats(t_first).median, stats(t_res).median, stats(t_res).p95,
                    stats(t_res).median * n_kf);
    }
    std::printf("}\n");
    return 0;
}
