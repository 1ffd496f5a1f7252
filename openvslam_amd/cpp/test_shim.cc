// Smoke program of the C++ class shims: reads two raw u8 frames, runs orb_extractor::extract on both and
// robust::brute_force_match between them exactly as tracking code would, and dumps the results for tests/test_cpp_shim.py.
// usage: test_shim rows cols nfeat frame_a.raw frame_b.raw out.bin
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <set>
#include <vector>

#include "openvslam/feature/orb_extractor.h"
#include "openvslam/match/area.h"
#include "openvslam/match/bow_tree.h"
#include "openvslam/match/fuse.h"
#include "openvslam/optimize/pose_optimizer.h"
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
    keyfrm.undist_keypts_ = keyfrm.keypts_;   // (a real keyframe always carries both; the resident handle holds the undistorted ones)
    keyfrm.camera_ = &cam;
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
    const std::vector<float> st_sfs = extractor.get_scale_factors(), st_isfs = extractor.get_inv_scale_factors();
    match::stereo(extractor_a.image_pyramid_, extractor.image_pyramid_, kps_a2, keyfrm.keypts_, desc_a2, keyfrm.descriptors_, st_sfs, st_isfs, 386.1448f,
                  0.5372f)
        .compute(stereo_x_right, depths);
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
    // pose_optimizer: the current frame with the landmarks match_current_and_last_frames gave it, from a perturbed pose
    data::frame pfrm = curr;
    pfrm.inv_level_sigma_sq_.resize(pfrm.scale_factors_.size());
    for (size_t l = 0; l < pfrm.scale_factors_.size(); ++l) pfrm.inv_level_sigma_sq_[l] = 1.0f / (pfrm.scale_factors_[l] * pfrm.scale_factors_[l]);
    pfrm.cam_pose_cw_(0, 3) = 0.02;
    pfrm.cam_pose_cw_(1, 3) = -0.015;
    pfrm.cam_pose_cw_(2, 3) = 0.01;
    const unsigned n_pose_valid = optimize::pose_optimizer().optimize(pfrm);
    // fuse::replace_duplication: keyframe = frame b (identity pose); landmarks_to_check = frame a's landmarks (positions as above); every
    // fifth keypoint of the keyframe already owns a landmark with three observations
    data::keyframe fkf;
    fkf.keypts_ = fkf.undist_keypts_ = keyfrm.keypts_;
    fkf.descriptors_ = keyfrm.descriptors_;
    fkf.num_keypts_ = keyfrm.num_keypts_;
    fkf.scale_factors_ = extractor.get_scale_factors();
    fkf.inv_level_sigma_sq_ = pfrm.inv_level_sigma_sq_;
    fkf.log_scale_factor_ = std::log(1.2f);
    fkf.camera_ = &pcam;
    fkf.landmarks_.assign(fkf.num_keypts_, nullptr);
    std::vector<std::unique_ptr<data::landmark>> fuse_own;
    for (unsigned j = 0; j < fkf.num_keypts_; j += 5) {
        fuse_own.emplace_back(new data::landmark());
        auto* lm = fuse_own.back().get();
        lm->num_observations_ = 2;          // + the observation added below = 3
        lm->add_observation(&fkf, j);
        fkf.landmarks_[j] = lm;
    }
    const size_t n_kf_lms = fuse_own.size();
    std::vector<data::landmark*> to_check;
    const std::vector<float> sfs = extractor.get_scale_factors();
    for (unsigned i = 0; i < last.num_keypts_; ++i) {
        fuse_own.emplace_back(new data::landmark());
        auto* lm = fuse_own.back().get();
        if (last.landmarks_[i]) lm->pos_w_ = last.landmarks_[i]->pos_w_;
        else lm->pos_w_(2) = -1.0;          // behind the camera
        const double dist = std::sqrt((lm->pos_w_(0) * lm->pos_w_(0) + lm->pos_w_(1) * lm->pos_w_(1)) + lm->pos_w_(2) * lm->pos_w_(2));
        for (int a = 0; a < 3; ++a) lm->mean_normal_(a) = lm->pos_w_(a) / dist;
        lm->max_valid_dist_ = (float)(dist * sfs[(size_t)last.undist_keypts_[i].octave] * 0.93);   // 0.93: keeps ceil(log ratio) off the knife edge
        lm->min_valid_dist_ = lm->max_valid_dist_ / sfs.back() * 0.8f;
        lm->descriptor_ = last.descriptors_.row((int)i);
        lm->num_observations_ = 1 + i % 4;
        lm->will_be_erased_ = (i % 17 == 3);
        to_check.push_back(lm);
    }
    const unsigned n_fused = match::fuse(0.6f).replace_duplication(&fkf, to_check, 3.0f);
    std::vector<int32_t> fuse_slots(fkf.num_keypts_, -1);   // who sits on keypoint j afterwards: index in to_check, or 100000 + original owner
    for (unsigned j = 0; j < fkf.num_keypts_; ++j) {
        if (!fkf.landmarks_[j]) continue;
        for (size_t q = 0; q < fuse_own.size(); ++q)
            if (fuse_own[q].get() == fkf.landmarks_[j]) fuse_slots[j] = q < n_kf_lms ? (int32_t)(100000 + q) : (int32_t)(q - n_kf_lms);
    }
    std::vector<uint8_t> fuse_erased(fuse_own.size());
    for (size_t q = 0; q < fuse_own.size(); ++q) fuse_erased[q] = fuse_own[q]->will_be_erased();
    // ---- the remaining projection overloads and bow_tree::match_keyframes on one scene: keyframes A (frame a) and B (frame b), identity
    //      poses, every keypoint owns a landmark at depth 5 back-projected from its own position
    auto make_kf = [&](data::keyframe& kf, const std::vector<cv::KeyPoint>& kps, const cv::Mat& desc, std::vector<std::unique_ptr<data::landmark>>& own) {
        kf.keypts_ = kf.undist_keypts_ = kps;
        kf.descriptors_ = desc;
        kf.num_keypts_ = kps.size();
        kf.scale_factors_ = sfs;
        kf.inv_level_sigma_sq_ = pfrm.inv_level_sigma_sq_;
        kf.log_scale_factor_ = std::log(1.2f);
        kf.camera_ = &pcam;
        kf.landmarks_.assign(kf.num_keypts_, nullptr);
        for (unsigned i = 0; i < kf.num_keypts_; ++i) {
            kf.bow_feat_vec_[desc.ptr((int)i)[0] & 127u].push_back(i);
            if (i % 9 == 4) continue;
            own.emplace_back(new data::landmark());
            auto* lm = own.back().get();
            lm->pos_w_(0) = ((double)kps[i].pt.x - pcam.cx_) / pcam.fx_ * 5.0;
            lm->pos_w_(1) = ((double)kps[i].pt.y - pcam.cy_) / pcam.fy_ * 5.0;
            lm->pos_w_(2) = 5.0;
            const double dist = std::sqrt((lm->pos_w_(0) * lm->pos_w_(0) + lm->pos_w_(1) * lm->pos_w_(1)) + 25.0);
            for (int a = 0; a < 3; ++a) lm->mean_normal_(a) = lm->pos_w_(a) / dist;
            lm->max_valid_dist_ = (float)(dist * sfs[(size_t)kps[i].octave] * 0.93);
            lm->min_valid_dist_ = lm->max_valid_dist_ / sfs.back() * 0.8f;
            lm->descriptor_ = desc.row((int)i);
            lm->will_be_erased_ = (i % 19 == 6);
            lm->add_observation(&kf, i);
            kf.landmarks_[i] = lm;
        }
    };
    std::vector<std::unique_ptr<data::landmark>> own_a, own_b;
    data::keyframe kfa, kfb;
    make_kf(kfa, frm.keypts_, frm.descriptors_, own_a);
    make_kf(kfb, keyfrm.keypts_, keyfrm.descriptors_, own_b);
    auto index_in = [](const data::keyframe& kf, const data::landmark* lm) -> int32_t {
        if (!lm) return -1;
        for (unsigned i = 0; i < kf.num_keypts_; ++i)
            if (kf.landmarks_[i] == lm) return (int32_t)i;
        return -2;
    };
    // match_frame_and_keyframe: frame b looks at keyframe A's landmarks from a camera displaced by the image shift at depth 5
    data::frame fb2 = frm_b;
    fb2.camera_ = &pcam;
    fb2.log_scale_factor_ = std::log(1.2f);
    fb2.landmarks_.assign(fb2.num_keypts_, nullptr);
    fb2.cam_pose_cw_(0, 3) = -4.0 / pcam.fx_ * 5.0;
    fb2.cam_pose_cw_(1, 3) = -3.0 / pcam.fy_ * 5.0;
    std::set<data::landmark*> already;
    for (unsigned i = 0; i < kfa.num_keypts_; i += 23)
        if (kfa.landmarks_[i]) already.insert(kfa.landmarks_[i]);
    const unsigned n_fk = match::projection(0.9f, true).match_frame_and_keyframe(fb2, &kfa, already, 10.0f, 100);
    std::vector<int32_t> fk_owner(fb2.num_keypts_);
    for (unsigned j = 0; j < fb2.num_keypts_; ++j) fk_owner[j] = index_in(kfa, fb2.landmarks_[j]);
    // match_by_Sim3_transform: keyframe B under the Sim3 [1.5 R | 1.5 t] of that same displaced pose, keyframe A's landmarks
    Mat44_t S_cw = fb2.cam_pose_cw_;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) S_cw(i, j) *= 1.5;
    std::vector<data::landmark*> lms_a = kfa.get_landmarks();
    std::vector<data::landmark*> matched_in_b(kfb.num_keypts_, nullptr);
    for (unsigned j = 0; j < kfb.num_keypts_; j += 29) matched_in_b[j] = lms_a[(j * 7) % kfa.num_keypts_];   // may be null
    const std::vector<data::landmark*> matched_in_b_before = matched_in_b;
    const unsigned n_s3 = match::projection(0.9f, false).match_by_Sim3_transform(&kfb, S_cw, lms_a, matched_in_b, 8.0f);
    std::vector<int32_t> s3_owner(kfb.num_keypts_);
    for (unsigned j = 0; j < kfb.num_keypts_; ++j) s3_owner[j] = index_in(kfa, matched_in_b[j]);
    // match_keyframes_mutually: Sim3_12 = (1, I, t_12) with t_12 the displacement that takes B's camera coordinates to A's
    std::vector<data::landmark*> matched_1(kfa.num_keypts_, nullptr);
    for (unsigned i = 0; i < kfa.num_keypts_; i += 31) matched_1[i] = kfb.landmarks_[(i * 3) % kfb.num_keypts_];   // may be null
    Mat33_t R_12;
    Vec3_t t_12;
    t_12(0) = 4.0 / pcam.fx_ * 5.0;
    t_12(1) = 3.0 / pcam.fy_ * 5.0;
    t_12(2) = 0.0;
    const std::vector<data::landmark*> matched_1_before = matched_1;
    const unsigned n_mut = match::projection(0.9f, false).match_keyframes_mutually(&kfa, &kfb, matched_1, 1.0f, R_12, t_12, 7.5f);
    std::vector<int32_t> mut_2_in_1(kfa.num_keypts_);
    for (unsigned i = 0; i < kfa.num_keypts_; ++i) mut_2_in_1[i] = matched_1[i] == matched_1_before[i] && matched_1[i] ? -3 : index_in(kfb, matched_1[i]);
    // bow_tree::match_keyframes
    std::vector<data::landmark*> bow_lms_1;
    const unsigned n_bk = match::bow_tree(0.75f, true).match_keyframes(&kfa, &kfb, bow_lms_1);
    std::vector<int32_t> bk_2_in_1(kfa.num_keypts_);
    for (unsigned i = 0; i < kfa.num_keypts_; ++i) bk_2_in_1[i] = index_in(kfb, bow_lms_1[i]);
    const int32_t hdr4[4] = {(int32_t)n_fk, (int32_t)n_s3, (int32_t)n_mut, (int32_t)n_bk};
    (void)matched_in_b_before;
    const int32_t hdr3[4] = {(int32_t)n_pose_valid, (int32_t)n_fused, (int32_t)n_kf_lms, (int32_t)to_check.size()};
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
    std::fwrite(hdr3, sizeof(hdr3), 1, f);
    std::fwrite(pfrm.cam_pose_cw_.m, sizeof(double), 12, f);
    std::vector<uint8_t> pose_outliers(pfrm.num_keypts_);
    for (unsigned j = 0; j < pfrm.num_keypts_; ++j) pose_outliers[j] = pfrm.outlier_flags_[j];
    std::fwrite(pose_outliers.data(), 1, pose_outliers.size(), f);
    std::fwrite(fuse_slots.data(), sizeof(int32_t), fuse_slots.size(), f);
    std::fwrite(fuse_erased.data(), 1, fuse_erased.size(), f);
    std::fwrite(hdr4, sizeof(hdr4), 1, f);
    std::fwrite(fk_owner.data(), sizeof(int32_t), fk_owner.size(), f);
    std::fwrite(s3_owner.data(), sizeof(int32_t), s3_owner.size(), f);
    std::fwrite(mut_2_in_1.data(), sizeof(int32_t), mut_2_in_1.size(), f);
    std::fwrite(bk_2_in_1.data(), sizeof(int32_t), bk_2_in_1.size(), f);
    std::fclose(f);
    std::printf("shim ok: %u + %u keypoints, %u matches, scale[7]=%f (area %u, frame_and_landmarks %u, bow %u, current_and_last %u, triangulation %u)\n",
                frm.num_keypts_, keyfrm.num_keypts_, n, extractor.get_scale_factors().at(7), n_area, n_proj, n_bow, n_cl, n_tri);
    return 0;
}
