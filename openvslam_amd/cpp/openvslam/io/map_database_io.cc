#include "map_database_io.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "msgpack_lite.h"

namespace openvslam {
namespace io {

namespace {

using msgpack_lite::value;

std::vector<uint8_t> read_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("map_database_io: cannot open " + path);
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)(n > 0 ? n : 0));
    const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size()) throw std::runtime_error("map_database_io: short read on " + path);
    return buf;
}

// convert_json_to_rotation / translation: quaternion (x, y, z, w) + t -> [R | t; 0 0 0 1]
Mat44_t pose_from(const value& rot, const value& trans) {
    if (rot.a.size() != 4 || trans.a.size() != 3) throw std::runtime_error("map_database_io: rot_cw / trans_cw size");
    const double x = rot.a[0].num(), y = rot.a[1].num(), z = rot.a[2].num(), w = rot.a[3].num();
    Mat44_t T;
    T(0, 0) = 1 - 2 * (y * y + z * z);
    T(0, 1) = 2 * (x * y - z * w);
    T(0, 2) = 2 * (x * z + y * w);
    T(1, 0) = 2 * (x * y + z * w);
    T(1, 1) = 1 - 2 * (x * x + z * z);
    T(1, 2) = 2 * (y * z - x * w);
    T(2, 0) = 2 * (x * z - y * w);
    T(2, 1) = 2 * (y * z + x * w);
    T(2, 2) = 1 - 2 * (x * x + y * y);
    for (int a = 0; a < 3; ++a) T(a, 3) = trans.a[(size_t)a].num();
    return T;
}

}   // namespace

loaded_map map_database_io::load_message_pack(const std::string& path) {
    const auto buf = read_file(path);
    const value root = msgpack_lite::reader(buf.data(), buf.size()).parse();
    if (root.kind != value::object || !root.has("cameras") || !root.has("keyframes") || !root.has("landmarks"))
        throw std::runtime_error("map_database_io: not an OpenVSLAM map database (cameras / keyframes / landmarks missing)");
    loaded_map m;
    if (root.has("frame_next_id")) m.frame_next_id = (unsigned int)root.at("frame_next_id").integer_value();
    if (root.has("keyframe_next_id")) m.keyframe_next_id = (unsigned int)root.at("keyframe_next_id").integer_value();
    if (root.has("landmark_next_id")) m.landmark_next_id = (unsigned int)root.at("landmark_next_id").integer_value();

    for (const auto& kv : root.at("cameras").o) {
        const value& j = kv.second;
        std::unique_ptr<camera::base> cam(new camera::base());
        const std::string model = j.at("model_type").s, setup = j.at("setup_type").s;
        cam->model_type_ = model == "Perspective" ? camera::model_type_t::Perspective
                           : model == "Fisheye"   ? camera::model_type_t::Fisheye
                                                  : camera::model_type_t::Equirectangular;
        cam->setup_type_ = setup == "Monocular" ? camera::setup_type_t::Monocular : setup == "Stereo" ? camera::setup_type_t::Stereo : camera::setup_type_t::RGBD;
        cam->cols_ = (unsigned int)j.at("cols").integer_value();
        cam->rows_ = (unsigned int)j.at("rows").integer_value();
        if (j.has("fx")) {
            cam->fx_ = j.at("fx").num();
            cam->fy_ = j.at("fy").num();
            cam->cx_ = j.at("cx").num();
            cam->cy_ = j.at("cy").num();
        }
        if (j.has("focal_x_baseline")) cam->focal_x_baseline_ = (float)j.at("focal_x_baseline").num();
        if (cam->fx_ > 0) cam->true_baseline_ = (float)(cam->focal_x_baseline_ / cam->fx_);
        cam->img_bounds_.max_x_ = (float)cam->cols_;   // undistorted input assumed (k1 .. k3 = 0 in the fixtures)
        cam->img_bounds_.max_y_ = (float)cam->rows_;
        m.cameras[kv.first] = std::move(cam);
    }

    for (const auto& kv : root.at("landmarks").o) {
        const value& j = kv.second;
        std::unique_ptr<data::landmark> lm(new data::landmark());
        lm->id_ = (unsigned int)std::stoul(kv.first);
        if (j.at("pos_w").a.size() != 3) throw std::runtime_error("map_database_io: pos_w size");
        for (int a = 0; a < 3; ++a) lm->pos_w_(a) = j.at("pos_w").a[(size_t)a].num();
        lm->num_observations_ = 0;
        m.landmarks[lm->id_] = std::move(lm);
    }

    for (const auto& kv : root.at("keyframes").o) {
        const value& j = kv.second;
        std::unique_ptr<data::keyframe> kf(new data::keyframe());
        kf->id_ = (unsigned int)std::stoul(kv.first);
        const auto cam_it = m.cameras.find(j.at("cam").s);
        if (cam_it == m.cameras.end()) throw std::runtime_error("map_database_io: keyframe names an unknown camera");
        kf->camera_ = cam_it->second.get();
        kf->cam_pose_cw_ = pose_from(j.at("rot_cw"), j.at("trans_cw"));
        const size_t n = (size_t)j.at("n_keypts").integer_value();
        const value &kps = j.at("keypts"), &und = j.at("undists"), &xr = j.at("x_rights"), &descs = j.at("descs"), &ids = j.at("lm_ids");
        if (kps.a.size() != n || und.a.size() != n || xr.a.size() != n || descs.a.size() != n || ids.a.size() != n)
            throw std::runtime_error("map_database_io: per-keypoint arrays disagree with n_keypts");
        kf->num_keypts_ = (unsigned int)n;
        kf->keypts_.resize(n);
        kf->undist_keypts_.resize(n);
        kf->stereo_x_right_.resize(n);
        kf->landmarks_.assign(n, nullptr);
        kf->descriptors_.create((int)n, 32, cv::CV_8U);
        for (size_t i = 0; i < n; ++i) {
            cv::KeyPoint& k = kf->keypts_[i];
            k.pt.x = (float)kps.a[i].at("pt").a[0].num();
            k.pt.y = (float)kps.a[i].at("pt").a[1].num();
            k.angle = (float)kps.a[i].at("ang").num();
            k.octave = (int)kps.a[i].at("oct").integer_value();
            kf->undist_keypts_[i] = k;
            kf->undist_keypts_[i].pt.x = (float)und.a[i].a[0].num();
            kf->undist_keypts_[i].pt.y = (float)und.a[i].a[1].num();
            kf->stereo_x_right_[i] = (float)xr.a[i].num();
            if (descs.a[i].a.size() != 8) throw std::runtime_error("map_database_io: a descriptor is 8 x uint32");
            for (int w = 0; w < 8; ++w) {   // convert_json_to_descriptors: 8 little-endian uint32 = 32 bytes
                const uint32_t u = (uint32_t)descs.a[i].a[(size_t)w].integer_value();
                std::memcpy(kf->descriptors_.ptr((int)i) + 4 * w, &u, 4);
            }
        }
        // orb_params tables: scale_factors_[l] = scale_factor * scale_factors_[l - 1] in float
        const int levels = (int)j.at("n_scale_levels").integer_value();
        const float sfac = (float)j.at("scale_factor").num();
        kf->scale_factors_.assign((size_t)levels, 1.0f);
        kf->inv_level_sigma_sq_.assign((size_t)levels, 1.0f);
        for (int l = 1; l < levels; ++l) kf->scale_factors_[(size_t)l] = sfac * kf->scale_factors_[(size_t)l - 1];
        for (int l = 0; l < levels; ++l) kf->inv_level_sigma_sq_[(size_t)l] = 1.0f / (kf->scale_factors_[(size_t)l] * kf->scale_factors_[(size_t)l]);
        m.keyframes[kf->id_] = std::move(kf);
    }

    // associations: upstream registers landmark <-> keyframe while loading, keyframes in ascending id
    for (auto& kv : root.at("keyframes").o) {
        data::keyframe* kf = m.keyframes.at((unsigned int)std::stoul(kv.first)).get();
        const value& ids = kv.second.at("lm_ids");
        for (size_t i = 0; i < ids.a.size(); ++i) {
            const int64_t id = ids.a[i].integer_value();
            if (id < 0) continue;
            const auto it = m.landmarks.find((unsigned int)id);
            if (it == m.landmarks.end()) continue;
            kf->landmarks_[i] = it->second.get();
        }
    }
    for (auto& kf : m.keyframes)
        for (size_t i = 0; i < kf.second->landmarks_.size(); ++i)
            if (kf.second->landmarks_[i]) kf.second->landmarks_[i]->add_observation(kf.second.get(), (unsigned int)i);

    // graph_node::update_connections: weight = shared landmarks; keep >= 15 (the best one if none), strongest first
    for (auto& kf : m.keyframes) {
        std::map<data::keyframe*, unsigned int> weights;
        for (data::landmark* lm : kf.second->landmarks_) {
            if (!lm) continue;
            for (const auto& obs : lm->observations_)
                if (obs.first != kf.second.get()) ++weights[obs.first];
        }
        std::vector<std::pair<unsigned int, data::keyframe*>> all(0);
        for (const auto& w : weights) all.emplace_back(w.second, w.first);
        std::sort(all.begin(), all.end(), [](const std::pair<unsigned int, data::keyframe*>& a, const std::pair<unsigned int, data::keyframe*>& b) {
            return a.first != b.first ? a.first > b.first : a.second->id_ < b.second->id_;
        });
        auto& cov = kf.second->graph_node_->covisibilities_;
        for (const auto& w : all)
            if (w.first >= 15) cov.push_back(w.second);
        if (cov.empty() && !all.empty()) cov.push_back(all.front().second);
    }
    return m;
}

}   // namespace io
}   // namespace openvslam
