// Smoke program of the C++ class shims: reads two raw u8 frames, runs orb_extractor::extract on both and
// robust::brute_force_match between them exactly as tracking code would, and dumps the results for tests/test_cpp_shim.py.
// usage: test_shim rows cols nfeat frame_a.raw frame_b.raw out.bin
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "openvslam/feature/orb_extractor.h"
#include "openvslam/match/area.h"
#include "openvslam/match/bow_tree.h"
#include "openvslam/match/projection.h"
#include "openvslam/match/robust.h"
#include "openvslam/match/stereo.h"

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

int main(int argc, char** argv) {
    if (argc != 7) return 2;
    const int rows = std::atoi(argv[1]), cols = std::atoi(argv[2]), nfeat = std::atoi(argv[3]);
    const cv::Mat a = read_raw(argv[4], rows, cols), b = read_raw(argv[5], rows, cols);
    feature::orb_extractor extractor(feature::orb_params(nfeat, 1.2f, 8, 20, 7));
    data::frame frm;
    data::keyframe keyfrm;
    extractor.extract(a, cv::Mat(), frm.keypts_, frm.descriptors_);
    frm.num_keypts_ = frm.keypts_.size();
    const int pyr7_rows = extractor.image_pyramid_.at(7).rows, pyr7_cols = extractor.image_pyramid_.at(7).cols;
    extractor.extract(b, cv::Mat(), keyfrm.keypts_, keyfrm.descriptors_);
    keyfrm.num_keypts_ = keyfrm.keypts_.size();
    std::vector<std::unique_ptr<data::landmark>> lms;
    keyfrm.landmarks_.assign(keyfrm.num_keypts_, nullptr);
    for (unsigned i = 0; i < keyfrm.num_keypts_; ++i)
        if (i % 10 != 3) {   // every 10th keypoint has no landmark
            lms.emplace_back(new data::landmark());
            lms.back()->will_be_erased_ = (i % 10 == 7);
            keyfrm.landmarks_[i] = lms.back().get();
        }
    std::vector<std::pair<int, int>> matches;
    const unsigned n = match::robust(0.9f, false).brute_force_match(frm, &keyfrm, matches);
    FILE* f = std::fopen(argv[6], "wb");
    const int32_t hdr[5] = {(int32_t)frm.num_keypts_, (int32_t)keyfrm.num_keypts_, (int32_t)n, pyr7_rows, pyr7_cols};
    std::fwrite(hdr, sizeof(hdr), 1, f);
    std::fwrite(frm.keypts_.data(), sizeof(cv::KeyPoint), frm.num_keypts_, f);
    std::fwrite(frm.descriptors_.data, 32, frm.num_keypts_, f);
    std::fwrite(keyfrm.keypts_.data(), sizeof(cv::KeyPoint), keyfrm.num_keypts_, f);
    std::fwrite(keyfrm.descriptors_.data, 32, keyfrm.num_keypts_, f);
    for (const auto& m : matches) {
        const int32_t p[2] = {m.first, m.second};
        std::fwrite(p, sizeof(p), 1, f);
    }
    // ---- windowed matchers, driven as module::initializer / tracking_module / frame's stereo ctor drive them
    camera::base cam;
    cam.cols_ = cols;
    cam.rows_ = rows;
    cam.img_bounds_.max_x_ = (float)cols;
    cam.img_bounds_.max_y_ = (float)rows;
    data::frame frm_b;
    frm_b.keypts_ = keyfrm.keypts_;
    frm_b.descriptors_ = keyfrm.descriptors_;
    frm_b.num_keypts_ = keyfrm.num_keypts_;
    frm.undist_keypts_ = frm.keypts_;
    frm_b.undist_keypts_ = frm_b.keypts_;
    frm.camera_ = frm_b.camera_ = &cam;
    frm.scale_factors_ = frm_b.scale_factors_ = extractor.get_scale_factors();
    // area: initial guess = own position, margin 100
    std::vector<cv::Point2f> prev_matched_pts(frm.num_keypts_);
    for (unsigned i = 0; i < frm.num_keypts_; ++i) prev_matched_pts[i] = frm.undist_keypts_[i].pt;
    std::vector<int> matched_2_in_1;
    const unsigned n_area = match::area(0.9f, true).match_in_consistent_area(frm, frm_b, prev_matched_pts, matched_2_in_1, 100);
    // projection: frame a's keypoints become landmarks reprojected at their position shifted by the frame offset
    std::vector<std::unique_ptr<data::landmark>> local_lms_own;
    std::vector<data::landmark*> local_lms;
    for (unsigned i = 0; i < frm.num_keypts_; ++i) {
        local_lms_own.emplace_back(new data::landmark());
        auto* lm = local_lms_own.back().get();
        lm->descriptor_ = frm.descriptors_.row((int)i);
        lm->reproj_in_tracking_(0) = frm.keypts_[i].pt.x - 4.0;
        lm->reproj_in_tracking_(1) = frm.keypts_[i].pt.y - 3.0;
        lm->is_observable_in_tracking_ = (i % 7 != 0);
        lm->scale_level_in_tracking_ = frm.keypts_[i].octave;
        local_lms.push_back(lm);
    }
    frm_b.landmarks_.assign(frm_b.num_keypts_, nullptr);
    const unsigned n_proj = match::projection(0.8f, true).match_frame_and_landmarks(frm_b, local_lms, 5.0f);
    std::vector<int32_t> proj_assigned(frm.num_keypts_, -1);
    for (unsigned j = 0; j < frm_b.num_keypts_; ++j)
        for (unsigned l = 0; frm_b.landmarks_[j] && l < local_lms.size(); ++l)
            if (local_lms[l] == frm_b.landmarks_[j]) proj_assigned[l] = (int32_t)j;
    // bow: node = low 7 bits of the first descriptor byte
    for (unsigned i = 0; i < keyfrm.num_keypts_; ++i) keyfrm.bow_feat_vec_[keyfrm.descriptors_.ptr((int)i)[0] & 127u].push_back(i);
    for (unsigned i = 0; i < frm.num_keypts_; ++i) frm.bow_feat_vec_[frm.descriptors_.ptr((int)i)[0] & 127u].push_back(i);
    std::vector<data::landmark*> matched_lms_in_frm;
    const unsigned n_bow = match::bow_tree(0.75f, true).match_frame_and_keyframe(&keyfrm, frm, matched_lms_in_frm);
    std::vector<int32_t> bow_kf_in_frm(frm.num_keypts_, -1);
    for (unsigned j = 0; j < frm.num_keypts_; ++j)
        for (unsigned i = 0; matched_lms_in_frm[j] && i < keyfrm.num_keypts_; ++i)
            if (keyfrm.landmarks_[i] == matched_lms_in_frm[j]) bow_kf_in_frm[j] = (int32_t)i;
    // stereo: a second extractor holds frame a's pyramid, the first one frame b's
    feature::orb_extractor extractor_a(feature::orb_params(nfeat, 1.2f, 8, 20, 7));
    std::vector<cv::KeyPoint> kps_a2;
    cv::Mat desc_a2;
    extractor_a.extract(a, cv::Mat(), kps_a2, desc_a2);
    std::vector<float> stereo_x_right, depths;
    match::stereo(&extractor_a, &extractor, kps_a2, keyfrm.keypts_, desc_a2, keyfrm.descriptors_, 386.1448f, 0.5372f).compute(stereo_x_right, depths);
    // match_current_and_last_frames: frame a is the last frame, its landmarks sit where frame b (the current frame, identity pose) sees
    // them: the keypoint shifted by the frame offset, back-projected at a depth that varies with the index
    camera::base pcam = cam;
    pcam.fx_ = pcam.fy_ = 500.0;
    pcam.cx_ = cols / 2.0;
    pcam.cy_ = rows / 2.0;
    data::frame last = frm, curr = frm_b;
    last.camera_ = curr.camera_ = &pcam;
    std::vector<std::unique_ptr<data::landmark>> last_lms_own;
    last.landmarks_.assign(last.num_keypts_, nullptr);
    last.outlier_flags_.assign(last.num_keypts_, false);
    for (unsigned i = 0; i < last.num_keypts_; ++i) {
        if (i % 13 == 5) continue;   // no landmark
        last_lms_own.emplace_back(new data::landmark());
        auto* lm = last_lms_own.back().get();
        const double z = 2.0 + (double)(i % 7);
        lm->pos_w_(0) = (((double)last.undist_keypts_[i].pt.x - 4.0) - pcam.cx_) / pcam.fx_ * z;
        lm->pos_w_(1) = (((double)last.undist_keypts_[i].pt.y - 3.0) - pcam.cy_) / pcam.fy_ * z;
        lm->pos_w_(2) = z;
        lm->descriptor_ = last.descriptors_.row((int)i);
        last.landmarks_[i] = lm;
        last.outlier_flags_[i] = (i % 11 == 0);
    }
    curr.landmarks_.assign(curr.num_keypts_, nullptr);
    const unsigned n_cl = match::projection(0.9f, true).match_current_and_last_frames(curr, last, 15.0f);
    std::vector<int32_t> cl_assigned(last.num_keypts_, -1);
    for (unsigned j = 0; j < curr.num_keypts_; ++j)
        for (unsigned i = 0; curr.landmarks_[j] && i < last.num_keypts_; ++i)
            if (last.landmarks_[i] == curr.landmarks_[j]) cl_assigned[i] = (int32_t)j;
    // match_for_triangulation: keyframe 1 = frame a at the origin, keyframe 2 = frame b displaced along the image shift direction
    data::keyframe kf1, kf2;
    auto fill_kf = [&](data::keyframe& kf, const std::vector<cv::KeyPoint>& kps, const cv::Mat& desc, unsigned lm_every,
                       std::vector<std::unique_ptr<data::landmark>>& own) {
        kf.keypts_ = kf.undist_keypts_ = kps;
        kf.descriptors_ = desc;
        kf.num_keypts_ = kps.size();
        kf.scale_factors_ = extractor.get_scale_factors();
        kf.camera_ = &pcam;
        kf.landmarks_.assign(kf.num_keypts_, nullptr);
        kf.bearings_.resize(kf.num_keypts_);
        for (unsigned i = 0; i < kf.num_keypts_; ++i) {
            if (i % lm_every == 0) {
                own.emplace_back(new data::landmark());
                kf.landmarks_[i] = own.back().get();
            }
            const double vx = ((double)kps[i].pt.x - pcam.cx_) / pcam.fx_, vy = ((double)kps[i].pt.y - pcam.cy_) / pcam.fy_;
            const double nrm = std::sqrt((vx * vx + vy * vy) + 1.0);
            kf.bearings_[i](0) = vx / nrm;
            kf.bearings_[i](1) = vy / nrm;
            kf.bearings_[i](2) = 1.0 / nrm;
            kf.bow_feat_vec_[desc.ptr((int)i)[0] & 127u].push_back(i);
        }
    };
    std::vector<std::unique_ptr<data::landmark>> kf_lms_own;
    fill_kf(kf1, frm.keypts_, frm.descriptors_, 3, kf_lms_own);
    fill_kf(kf2, keyfrm.keypts_, keyfrm.descriptors_, 4, kf_lms_own);
    kf2.cam_pose_cw_(0, 3) = -0.2;
    kf2.cam_pose_cw_(1, 3) = -0.15;
    Mat33_t E_12;   // [t_12]x R_12 with R_12 = I, t_12 = t_1 - R_12 t_2 = (0.2, 0.15, 0)
    const double t12[3] = {0.2, 0.15, 0.0};
    E_12(0, 0) = 0, E_12(0, 1) = -t12[2], E_12(0, 2) = t12[1];
    E_12(1, 0) = t12[2], E_12(1, 1) = 0, E_12(1, 2) = -t12[0];
    E_12(2, 0) = -t12[1], E_12(2, 1) = t12[0], E_12(2, 2) = 0;
    std::vector<std::pair<unsigned int, unsigned int>> tri_pairs;
    const unsigned n_tri = match::robust(0.6f, true).match_for_triangulation(&kf1, &kf2, E_12, tri_pairs);
    std::vector<int32_t> tri_2_in_1(kf1.num_keypts_, -1);
    for (const auto& pr : tri_pairs) tri_2_in_1[pr.first] = (int32_t)pr.second;
    const int32_t hdr2[6] = {(int32_t)n_area, (int32_t)n_proj, (int32_t)n_bow, (int32_t)stereo_x_right.size(), (int32_t)n_cl, (int32_t)n_tri};
    std::fwrite(hdr2, sizeof(hdr2), 1, f);
    std::fwrite(matched_2_in_1.data(), sizeof(int), matched_2_in_1.size(), f);
    std::fwrite(prev_matched_pts.data(), sizeof(cv::Point2f), prev_matched_pts.size(), f);
    std::fwrite(proj_assigned.data(), sizeof(int32_t), proj_assigned.size(), f);
    std::fwrite(bow_kf_in_frm.data(), sizeof(int32_t), bow_kf_in_frm.size(), f);
    std::fwrite(stereo_x_right.data(), sizeof(float), stereo_x_right.size(), f);
    std::fwrite(depths.data(), sizeof(float), depths.size(), f);
    std::fwrite(cl_assigned.data(), sizeof(int32_t), cl_assigned.size(), f);
    std::fwrite(tri_2_in_1.data(), sizeof(int32_t), tri_2_in_1.size(), f);
    std::fclose(f);
    std::printf("shim ok: %u + %u keypoints, %u matches, scale[7]=%f\n", frm.num_keypts_, keyfrm.num_keypts_, n, extractor.get_scale_factors().at(7));
    return 0;
}
