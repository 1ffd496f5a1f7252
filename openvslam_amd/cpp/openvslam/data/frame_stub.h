// Minimal stand-ins for data::frame / data::keyframe / data::landmark / camera::base: only the members the matcher shims read
// (expected: src/openvslam/data/{frame,keyframe,landmark}.h, src/openvslam/camera/base.h). In an OpenVSLAM checkout the real
// headers are used instead and the shim bodies compile unchanged.
#pragma once
#include <memory>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "../../cv_stub.h"

namespace openvslam {

struct Vec2_t {   // Eigen::Vector2d stand-in
    double v[2] = {0, 0};
    double operator()(int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
};
struct Vec3_t {   // Eigen::Vector3d stand-in
    double v[3] = {0, 0, 0};
    double operator()(int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
};
struct Mat33_t {   // Eigen::Matrix3d stand-in; (r, c) access as Eigen
    double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double operator()(int r, int c) const { return m[3 * r + c]; }
    double& operator()(int r, int c) { return m[3 * r + c]; }
};
struct Mat44_t {   // Eigen::Matrix4d stand-in
    double m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double operator()(int r, int c) const { return m[4 * r + c]; }
    double& operator()(int r, int c) { return m[4 * r + c]; }
};

namespace camera {
struct image_bounds {
    float min_x_ = 0, max_x_ = 0, min_y_ = 0, max_y_ = 0;
};
enum class setup_type_t { Monocular = 0, Stereo = 1, RGBD = 2 };
enum class model_type_t { Perspective = 0, Fisheye = 1, Equirectangular = 2, RadialDivision = 3 };
class base {
public:
    setup_type_t setup_type_ = setup_type_t::Monocular;
    model_type_t model_type_ = model_type_t::Perspective;
    double fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0;   // camera::perspective's members (the shims read them through a static_cast there)
    unsigned int cols_ = 0, rows_ = 0;
    image_bounds img_bounds_;
    unsigned int num_grid_cols_ = 64, num_grid_rows_ = 48;
    float focal_x_baseline_ = 0, true_baseline_ = 0;
};
}   // namespace camera

namespace data {

class keyframe;

// data::map_database: only the lock local_bundle_adjuster takes around its write-back
class map_database {
public:
    static inline std::mutex mtx_database_;
};

using bow_feature_vector = std::map<unsigned int, std::vector<unsigned int>>;   // DBoW2::FeatureVector

class landmark {
public:
    unsigned int id_ = 0;
    bool will_be_erased() const { return will_be_erased_; }
    std::map<keyframe*, unsigned int> get_observations() const { return observations_; }
    void set_pos_in_world(const Vec3_t& pos_w) { pos_w_ = pos_w; }
    void update_normal_and_depth() { ++num_normal_updates_; }   // upstream recomputes mean_normal_ / valid distances from the observations
    void erase_observation(keyframe* keyfrm) {
        if (observations_.erase(keyfrm)) --num_observations_;
    }
    unsigned int num_normal_updates_ = 0;
    bool has_observation() const { return num_observations_ > 0; }
    cv::Mat get_descriptor() const { return descriptor_; }
    Vec3_t get_pos_in_world() const { return pos_w_; }
    Vec3_t get_obs_mean_normal() const { return mean_normal_; }
    float get_min_valid_distance() const { return 0.7 * min_valid_dist_; }   // upstream widens the stored range by 30 %
    float get_max_valid_distance() const { return 1.3 * max_valid_dist_; }
    unsigned int num_observations() const { return num_observations_; }
    bool is_observed_in_keyframe(keyframe* keyfrm) const { return observations_.count(keyfrm) != 0; }
    int get_index_in_keyframe(keyframe* keyfrm) const {
        const auto it = observations_.find(keyfrm);
        return it == observations_.end() ? -1 : (int)it->second;
    }
    void add_observation(keyframe* keyfrm, unsigned int idx) {
        if (observations_.count(keyfrm)) return;
        observations_[keyfrm] = idx;
        ++num_observations_;
    }
    inline void replace(landmark* lm);   // defined after keyframe
    Vec3_t pos_w_, mean_normal_;
    float min_valid_dist_ = 0, max_valid_dist_ = 0;
    std::map<keyframe*, unsigned int> observations_;
    bool will_be_erased_ = false;
    unsigned int num_observations_ = 1;
    cv::Mat descriptor_;
    // tracking information
    Vec2_t reproj_in_tracking_;
    float x_right_in_tracking_ = -1.0f;
    bool is_observable_in_tracking_ = false;
    int scale_level_in_tracking_ = 0;
};

// The matcher-side data of a frame -- and of the keyframe made from it -- resident in HBM (ovs_frame_dev, include/ovslam_hip.h): created by
// the first matcher that needs it, on the device the frame's extractor ran on, and shared by copies of the frame (upstream copies frames:
// last_frm = curr_frm) AND by the keyframe constructed from it (keyframe::keyframe(const frame&) copies the member: a keyframe's keypoints,
// descriptors, stereo_x_right_ and grid ARE its frame's, so creating a keyframe uploads nothing). The ONE member an integration adds to
// data::frame and data::keyframe. Keypoints and descriptors never change after the frame's constructor, so there is no invalidation;
// tracking and mapping threads may ask for the handle at the same time, so creation happens under the holder's lock, and a user keeps
// its own reference for the duration of a call (a drop() after a device failure never frees a handle another thread is using).
struct frame_device_cache {
    int device = 0;                    // HIP device of the extractor that produced the keypoints (set before the first matcher call)
    std::mutex mu;
    std::shared_ptr<void> handle;      // ovs_frame_dev with its deleter
    bool has_bearings = false;         // bearings_ attached (robust::match_for_triangulation needs them; tracking never does)
    // create: () -> std::shared_ptr<void> (throws on failure); extend: (void*) -> void, run once under the lock when `want_bearings`
    template <class Create, class Extend>
    std::shared_ptr<void> get(Create&& create, bool want_bearings, Extend&& extend) {
        std::lock_guard<std::mutex> lock(mu);
        if (!handle) {
            handle = create();
            has_bearings = false;
        }
        if (want_bearings && !has_bearings) {
            extend(handle.get());
            has_bearings = true;
        }
        return handle;
    }
    void drop() {
        std::lock_guard<std::mutex> lock(mu);
        handle.reset();
        has_bearings = false;
    }
    bool resident() {
        std::lock_guard<std::mutex> lock(mu);
        return handle != nullptr;
    }
};

class frame {
public:
    // never null, never reassigned except by whole-frame assignment (which shares the other frame's cache)
    std::shared_ptr<frame_device_cache> device_cache_ = std::make_shared<frame_device_cache>();
    unsigned int num_keypts_ = 0;
    std::vector<cv::KeyPoint> keypts_;
    std::vector<cv::KeyPoint> undist_keypts_;
    std::vector<float> stereo_x_right_;
    std::vector<Vec3_t> bearings_;
    cv::Mat descriptors_;
    std::vector<landmark*> landmarks_;
    std::vector<bool> outlier_flags_;
    std::vector<float> scale_factors_;
    std::vector<float> inv_level_sigma_sq_;
    float log_scale_factor_ = 0;
    camera::base* camera_ = nullptr;
    bow_feature_vector bow_feat_vec_;
    Mat44_t cam_pose_cw_;
    void set_cam_pose(const Mat44_t& cam_pose_cw) { cam_pose_cw_ = cam_pose_cw; }
};

class keyframe;
// data::graph_node: only the covisibility list local_bundle_adjuster walks
class graph_node {
public:
    std::vector<keyframe*> get_covisibilities() const { return covisibilities_; }
    std::vector<keyframe*> covisibilities_;
};

class keyframe {
public:
    keyframe() = default;
    //! upstream: keyframe::keyframe(const frame& frm, map_database*, bow_database*) copies the frame's members -- the device cache with them
    explicit keyframe(const frame& frm)
        : device_cache_(frm.device_cache_), num_keypts_(frm.num_keypts_), keypts_(frm.keypts_), undist_keypts_(frm.undist_keypts_),
          stereo_x_right_(frm.stereo_x_right_), bearings_(frm.bearings_), descriptors_(frm.descriptors_), landmarks_(frm.landmarks_),
          scale_factors_(frm.scale_factors_), inv_level_sigma_sq_(frm.inv_level_sigma_sq_), log_scale_factor_(frm.log_scale_factor_),
          camera_(frm.camera_), bow_feat_vec_(frm.bow_feat_vec_), cam_pose_cw_(frm.cam_pose_cw_) {}
    std::shared_ptr<frame_device_cache> device_cache_ = std::make_shared<frame_device_cache>();
    unsigned int id_ = 0;
    bool will_be_erased() const { return will_be_erased_; }
    bool will_be_erased_ = false;
    graph_node graph_node_storage_;
    graph_node* graph_node_ = &graph_node_storage_;
    void set_cam_pose(const Mat44_t& cam_pose_cw) { cam_pose_cw_ = cam_pose_cw; }
    void erase_landmark(landmark* lm) {
        const int idx = lm->get_index_in_keyframe(this);
        if (0 <= idx) landmarks_.at((size_t)idx) = nullptr;
    }
    unsigned int num_keypts_ = 0;
    std::vector<cv::KeyPoint> keypts_;
    std::vector<cv::KeyPoint> undist_keypts_;
    std::vector<float> stereo_x_right_;
    std::vector<Vec3_t> bearings_;
    cv::Mat descriptors_;
    std::vector<landmark*> landmarks_;
    std::vector<float> scale_factors_;
    std::vector<float> inv_level_sigma_sq_;
    float log_scale_factor_ = 0;
    camera::base* camera_ = nullptr;
    bow_feature_vector bow_feat_vec_;
    Mat44_t cam_pose_cw_;
    Mat44_t get_cam_pose() const { return cam_pose_cw_; }
    std::vector<landmark*> get_landmarks() const { return landmarks_; }
    landmark* get_landmark(unsigned int idx) const { return landmarks_.at(idx); }
    void add_landmark(landmark* lm, unsigned int idx) { landmarks_.at(idx) = lm; }
    Mat33_t get_rotation() const {
        Mat33_t r;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r(i, j) = cam_pose_cw_(i, j);
        return r;
    }
    Vec3_t get_translation() const {
        Vec3_t t;
        for (int i = 0; i < 3; ++i) t(i) = cam_pose_cw_(i, 3);
        return t;
    }
    Vec3_t get_cam_center() const {   // -R^T t
        Vec3_t c;
        for (int i = 0; i < 3; ++i)
            c(i) = -((cam_pose_cw_(0, i) * cam_pose_cw_(0, 3) + cam_pose_cw_(1, i) * cam_pose_cw_(1, 3)) + cam_pose_cw_(2, i) * cam_pose_cw_(2, 3));
        return c;
    }
};

// landmark::replace(lm): this landmark's observations move to lm (keyframes that already see lm just drop this one), then it is erased
inline void landmark::replace(landmark* lm) {
    if (lm == this) return;
    for (const auto& obs : observations_) {
        keyframe* kf = obs.first;
        if (!lm->is_observed_in_keyframe(kf)) {
            kf->add_landmark(lm, obs.second);
            lm->add_observation(kf, obs.second);
        } else {
            kf->add_landmark(nullptr, obs.second);
        }
    }
    observations_.clear();
    num_observations_ = 0;
    will_be_erased_ = true;
}

}   // namespace data
}   // namespace openvslam
