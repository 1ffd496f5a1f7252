// Drives optimize::local_bundle_adjuster::optimize(curr_keyfrm, force_stop_flag) through the class, on a scene tests/test_cpp_shim.py
// writes: keyframes with poses / keypoints / covisibility, landmarks with observations. Dumps poses, positions and which observations
// were erased. Also drives robust::match_frame_and_keyframe (usage: test_lba_shim lba scene.bin out.bin | mfk scene.bin out.bin).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "openvslam/data/bow_vocabulary.h"
#include "openvslam/io/map_database_io.h"
#include "openvslam/match/robust.h"
#include "openvslam/optimize/local_bundle_adjuster.h"
#include "openvslam/optimize/pose_optimizer.h"

using namespace openvslam;

namespace {
std::vector<unsigned char> read_all(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) {
        std::fprintf(stderr, "cannot read %s\n", path);
        std::exit(2);
    }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b((size_t)n);
    if (std::fread(b.data(), 1, (size_t)n, f) != (size_t)n) std::exit(2);
    std::fclose(f);
    return b;
}
struct Reader {
    const unsigned char* p;
    template <typename T>
    T get() {
        T v;
        std::memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
};

int run_lba(const char* in, const char* out) {
    const auto buf = read_all(in);
    Reader r{buf.data()};
    const int n_kf = r.get<int32_t>(), n_lm = r.get<int32_t>(), n_obs = r.get<int32_t>(), curr = r.get<int32_t>(), setup_model = r.get<int32_t>();
    const int stop_before = r.get<int32_t>();
    const int setup = setup_model & 0xff, model = setup_model >> 8;   // model: camera::model_type_t (0 perspective, 2 equirectangular)
    camera::base cam;
    cam.fx_ = r.get<double>();
    cam.fy_ = r.get<double>();
    cam.cx_ = r.get<double>();
    cam.cy_ = r.get<double>();
    cam.focal_x_baseline_ = (float)r.get<double>();
    cam.setup_type_ = (camera::setup_type_t)setup;
    cam.model_type_ = (camera::model_type_t)model;
    if (cam.model_type_ == camera::model_type_t::Equirectangular) {   // the scene file carries (cols, rows) in the fx / fy slots
        cam.cols_ = (unsigned)cam.fx_;
        cam.rows_ = (unsigned)cam.fy_;
    }
    std::vector<float> ils(8);
    for (auto& v : ils) v = r.get<float>();
    std::vector<std::unique_ptr<data::keyframe>> kfs;
    std::vector<int> covisible((size_t)n_kf);
    for (int k = 0; k < n_kf; ++k) {
        kfs.emplace_back(new data::keyframe());
        kfs.back()->id_ = (unsigned)r.get<int32_t>();
        covisible[k] = r.get<int32_t>();
        for (int i = 0; i < 16; ++i) kfs.back()->cam_pose_cw_.m[i] = r.get<double>();
        kfs.back()->camera_ = &cam;
        kfs.back()->inv_level_sigma_sq_ = ils;
    }
    std::vector<std::unique_ptr<data::landmark>> lms;
    for (int j = 0; j < n_lm; ++j) {
        lms.emplace_back(new data::landmark());
        lms.back()->id_ = (unsigned)r.get<int32_t>();
        lms.back()->num_observations_ = 0;
        for (int a = 0; a < 3; ++a) lms.back()->pos_w_(a) = r.get<double>();
    }
    struct Obs {
        int kf, lm;
        unsigned idx;
    };
    std::vector<Obs> obs((size_t)n_obs);
    for (int i = 0; i < n_obs; ++i) {
        const int k = r.get<int32_t>(), j = r.get<int32_t>();
        cv::KeyPoint kp;
        kp.pt.x = r.get<float>();
        kp.pt.y = r.get<float>();
        const float x_right = r.get<float>();
        kp.octave = r.get<int32_t>();
        data::keyframe* kf = kfs[(size_t)k].get();
        const unsigned idx = kf->num_keypts_++;
        kf->undist_keypts_.push_back(kp);
        kf->keypts_.push_back(kp);
        kf->stereo_x_right_.push_back(x_right);
        kf->landmarks_.push_back(lms[(size_t)j].get());
        lms[(size_t)j]->add_observation(kf, idx);
        obs[(size_t)i] = {k, j, idx};
    }
    for (int k = 0; k < n_kf; ++k)
        if (covisible[k] && k != curr) kfs[(size_t)curr]->graph_node_->covisibilities_.push_back(kfs[(size_t)k].get());
    kfs[(size_t)curr]->graph_node_->covisibilities_.push_back(nullptr);   // upstream tolerates null / erased entries
    bool stop = stop_before != 0;
    optimize::local_bundle_adjuster(5, 10).optimize(kfs[(size_t)curr].get(), &stop);
    FILE* f = std::fopen(out, "wb");
    for (int k = 0; k < n_kf; ++k) std::fwrite(kfs[(size_t)k]->cam_pose_cw_.m, sizeof(double), 16, f);
    for (int j = 0; j < n_lm; ++j) {
        std::fwrite(lms[(size_t)j]->pos_w_.v, sizeof(double), 3, f);
        const int32_t upd = (int32_t)lms[(size_t)j]->num_normal_updates_;
        std::fwrite(&upd, 4, 1, f);
    }
    for (const auto& o : obs) {
        // erased <=> the keyframe slot was cleared AND the landmark forgot the keyframe
        const bool slot_cleared = kfs[(size_t)o.kf]->landmarks_[o.idx] == nullptr;
        const bool lm_forgot = !lms[(size_t)o.lm]->is_observed_in_keyframe(kfs[(size_t)o.kf].get());
        const uint8_t e = slot_cleared && lm_forgot ? 1 : (slot_cleared != lm_forgot ? 2 : 0);
        std::fwrite(&e, 1, 1, f);
    }
    std::fclose(f);
    std::printf("lba shim ok: %d keyframes, %d landmarks, %d observations\n", n_kf, n_lm, n_obs);
    return 0;
}

// robust::match_frame_and_keyframe: descriptors + bearings of a frame and a keyframe; dumps the matched keyframe index per frame keypoint
int run_mfk(const char* in, const char* out) {
    const auto buf = read_all(in);
    Reader r{buf.data()};
    const int n1 = r.get<int32_t>(), n2 = r.get<int32_t>(), check_orientation = r.get<int32_t>();
    const float ratio = r.get<float>();
    data::frame frm;
    data::keyframe kf;
    auto fill = [&](int n, std::vector<cv::KeyPoint>& kps, cv::Mat& desc, std::vector<Vec3_t>& bearings) {
        kps.resize((size_t)n);
        desc.create(n, 32, cv::CV_8U);
        bearings.resize((size_t)n);
        for (int i = 0; i < n; ++i) kps[(size_t)i].angle = r.get<float>();
        for (int i = 0; i < n; ++i)
            for (int b = 0; b < 32; ++b) desc.ptr(i)[b] = r.get<uint8_t>();
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) bearings[(size_t)i](a) = r.get<double>();
    };
    fill(n1, frm.keypts_, frm.descriptors_, frm.bearings_);
    fill(n2, kf.keypts_, kf.descriptors_, kf.bearings_);
    frm.num_keypts_ = (unsigned)n1;
    kf.num_keypts_ = (unsigned)n2;
    std::vector<std::unique_ptr<data::landmark>> lms;
    kf.landmarks_.assign((size_t)n2, nullptr);
    for (int i = 0; i < n2; ++i)
        if (r.get<uint8_t>()) {
            lms.emplace_back(new data::landmark());
            kf.landmarks_[(size_t)i] = lms.back().get();
        }
    match::robust matcher(ratio, check_orientation != 0);
    std::vector<std::pair<int, int>> bf;
    const unsigned n_bf = matcher.brute_force_match(frm, &kf, bf);
    std::vector<data::landmark*> matched;
    const unsigned n_inl = matcher.match_frame_and_keyframe(frm, &kf, matched);
    FILE* f = std::fopen(out, "wb");
    const int32_t hdr[2] = {(int32_t)n_bf, (int32_t)n_inl};
    std::fwrite(hdr, sizeof(hdr), 1, f);
    for (const auto& m : bf) {
        const int32_t p[2] = {m.first, m.second};
        std::fwrite(p, sizeof(p), 1, f);
    }
    for (int i = 0; i < n1; ++i) {
        int32_t owner = -1;
        for (int j = 0; matched[(size_t)i] && j < n2; ++j)
            if (kf.landmarks_[(size_t)j] == matched[(size_t)i]) owner = j;
        std::fwrite(&owner, 4, 1, f);
    }
    std::fclose(f);
    std::printf("mfk shim ok: %u brute-force matches, %u inliers\n", n_bf, n_inl);
    return 0;
}
// data::bow_vocabulary: load a vocabulary file, transform N x 32 descriptors, dump the BowVector and the FeatureVector
int run_bow(const char* vocab_path, const char* desc_path, const char* out) {
    data::bow_vocabulary voc;
    voc.loadFromBinaryFile(vocab_path);
    const auto buf = read_all(desc_path);
    const int n = (int)(buf.size() / 32);
    cv::Mat desc(n, 32, cv::CV_8U);
    for (int i = 0; i < n; ++i) std::memcpy(desc.ptr(i), buf.data() + (size_t)i * 32, 32);
    std::vector<cv::Mat> feats;
    for (int i = 0; i < n; ++i) feats.push_back(desc.row(i));
    data::bow_vector v1, v2;
    data::bow_feature_vector_t f1, f2;
    voc.transform(feats, v1, f1, 4);     // DBoW2 signature
    voc.transform(desc, 4, v2, f2);      // FBoW signature
    if (v1 != v2 || f1 != f2) return 3;
    FILE* f = std::fopen(out, "wb");
    const int32_t hdr[2] = {(int32_t)v1.size(), (int32_t)f1.size()};
    std::fwrite(hdr, sizeof(hdr), 1, f);
    for (const auto& e : v1) {
        const int32_t w = (int32_t)e.first;
        std::fwrite(&w, 4, 1, f);
        std::fwrite(&e.second, 8, 1, f);
    }
    for (const auto& e : f1) {
        const int32_t h2[2] = {(int32_t)e.first, (int32_t)e.second.size()};
        std::fwrite(h2, sizeof(h2), 1, f);
        for (const auto i : e.second) {
            const int32_t ii = (int32_t)i;
            std::fwrite(&ii, 4, 1, f);
        }
    }
    std::fclose(f);
    std::printf("bow shim ok: %d features, %zu words, %zu nodes\n", n, v1.size(), f1.size());
    return 0;
}

// io::map_database_io::load_message_pack: `mapinfo` prints what was read (no device needed); `map` runs local_bundle_adjuster::optimize on
// keyframe `curr` of the loaded map and dumps every keyframe pose (16 doubles, ascending id) and landmark position (ascending id)
int run_mapinfo(const char* in) {
    const io::loaded_map m = io::map_database_io::load_message_pack(in);
    size_t n_obs = 0, n_kp = 0, n_cov = 0;
    unsigned long long desc_sum = 0;
    for (const auto& kf : m.keyframes) {
        n_kp += kf.second->num_keypts_;
        n_cov += kf.second->graph_node_->covisibilities_.size();
        for (int r = 0; r < kf.second->descriptors_.rows; ++r)
            for (int c = 0; c < 32; ++c) desc_sum += (unsigned long long)kf.second->descriptors_.ptr(r)[c] * (unsigned)(c + 1);
    }
    for (const auto& lm : m.landmarks) n_obs += lm.second->num_observations();
    std::printf("map: %zu cameras, %zu keyframes, %zu landmarks, %zu keypoints, %zu observations, %zu covisibility links, descriptor checksum %llu\n",
                m.cameras.size(), m.keyframes.size(), m.landmarks.size(), n_kp, n_obs, n_cov, desc_sum);
    return 0;
}

int run_map(const char* in, const char* curr_id, const char* out) {
    io::loaded_map m = io::map_database_io::load_message_pack(in);
    data::keyframe* curr = m.keyframes.at((unsigned int)std::atoi(curr_id)).get();
    bool stop = false;
    optimize::local_bundle_adjuster(5, 10).optimize(curr, &stop);
    FILE* f = std::fopen(out, "wb");
    for (const auto& kf : m.keyframes) std::fwrite(kf.second->cam_pose_cw_.m, sizeof(double), 16, f);
    for (const auto& lm : m.landmarks) std::fwrite(lm.second->pos_w_.v, sizeof(double), 3, f);
    std::fclose(f);
    std::printf("map lba ok: %zu keyframes, %zu landmarks, current keyframe %u with %zu covisibilities\n", m.keyframes.size(), m.landmarks.size(),
                curr->id_, curr->graph_node_->covisibilities_.size());
    return 0;
}
// optimize::pose_optimizer::optimize(frm) through the class. File: int32 model (camera::model_type_t), int32 n, double[4] (fx fy cx cy, or
// cols rows - - for an equirectangular camera), float[8] inv_level_sigma_sq, double[16] cam_pose_cw, then per keypoint: double[3] landmark
// position, float x, float y, int32 octave. Dumps the pose, the return value and outlier_flags_.
int run_pose(const char* in, const char* out) {
    const auto buf = read_all(in);
    Reader r{buf.data()};
    const int model = r.get<int32_t>(), n = r.get<int32_t>();
    camera::base cam;
    cam.fx_ = r.get<double>();
    cam.fy_ = r.get<double>();
    cam.cx_ = r.get<double>();
    cam.cy_ = r.get<double>();
    cam.model_type_ = (camera::model_type_t)model;
    if (cam.model_type_ == camera::model_type_t::Equirectangular) {
        cam.cols_ = (unsigned)cam.fx_;
        cam.rows_ = (unsigned)cam.fy_;
    }
    data::frame frm;
    frm.camera_ = &cam;
    frm.inv_level_sigma_sq_.resize(8);
    for (auto& v : frm.inv_level_sigma_sq_) v = r.get<float>();
    for (int i = 0; i < 16; ++i) frm.cam_pose_cw_.m[i] = r.get<double>();
    std::vector<std::unique_ptr<data::landmark>> lms;
    for (int i = 0; i < n; ++i) {
        lms.emplace_back(new data::landmark());
        for (int a = 0; a < 3; ++a) lms.back()->pos_w_(a) = r.get<double>();
        cv::KeyPoint kp;
        kp.pt.x = r.get<float>();
        kp.pt.y = r.get<float>();
        kp.octave = r.get<int32_t>();
        frm.undist_keypts_.push_back(kp);
        frm.keypts_.push_back(kp);
        frm.landmarks_.push_back(lms.back().get());
    }
    frm.num_keypts_ = (unsigned)n;
    const unsigned int nv = optimize::pose_optimizer(4, 10).optimize(frm);
    FILE* f = std::fopen(out, "wb");
    std::fwrite(frm.cam_pose_cw_.m, sizeof(double), 16, f);
    const int32_t nvi = (int32_t)nv;
    std::fwrite(&nvi, 4, 1, f);
    for (int i = 0; i < n; ++i) {
        const uint8_t o = frm.outlier_flags_[(size_t)i] ? 1 : 0;
        std::fwrite(&o, 1, 1, f);
    }
    std::fclose(f);
    std::printf("pose shim ok: %d keypoints, %u valid\n", n, nv);
    return 0;
}
}   // namespace

int main(int argc, char** argv) {
    if (argc == 4 && std::string(argv[1]) == "pose") return run_pose(argv[2], argv[3]);
    if (argc == 3 && std::string(argv[1]) == "mapinfo") return run_mapinfo(argv[2]);
    if (argc == 5 && std::string(argv[1]) == "map") return run_map(argv[2], argv[3], argv[4]);
    if (argc == 5 && std::string(argv[1]) == "bow") return run_bow(argv[2], argv[3], argv[4]);
    if (argc != 4) return 2;
    if (std::string(argv[1]) == "lba") return run_lba(argv[2], argv[3]);
    if (std::string(argv[1]) == "mfk") return run_mfk(argv[2], argv[3]);
    return 2;
}
