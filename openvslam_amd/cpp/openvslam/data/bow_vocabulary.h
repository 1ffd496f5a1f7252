// data::bow_vocabulary (expected: src/openvslam/data/bow_vocabulary.h -- a typedef of DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>
// or fbow::Vocabulary): loading an on-disk ORB vocabulary and the transform() data::frame::compute_bow / keyframe::compute_bow call.
// The tree descent runs on the MI355X (ovs_bow_transform); the two std::maps are filled here in DBoW2's order of operations:
// v[word] += weight in feature order (features whose word weight is 0 are dropped), then L1 normalisation; fv[node].push_back(i).
#pragma once
#include <ovslam_hip.h>

#include <cmath>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../cv_stub.h"
#include "../util/device_policy.h"

namespace openvslam {
namespace data {

using bow_vector = std::map<unsigned int, double>;                              // DBoW2::BowVector
using bow_feature_vector_t = std::map<unsigned int, std::vector<unsigned int>>;   // DBoW2::FeatureVector

class bow_vocabulary {
public:
    bow_vocabulary() = default;
    ~bow_vocabulary() {
        if (v_) ovs_vocab_destroy(v_);
    }
    bow_vocabulary(const bow_vocabulary&) = delete;
    bow_vocabulary& operator=(const bow_vocabulary&) = delete;

    //! DBoW2 fork: loadFromBinaryFile(orb_vocab.dbow2); the text and FBoW formats are recognised by the same reader
    void loadFromBinaryFile(const std::string& filename) { load(filename); }
    void loadFromTextFile(const std::string& filename) { load(filename); }
    //! fbow::Vocabulary::readFromFile(orb_vocab.fbow)
    void readFromFile(const std::string& filename) { load(filename); }
    bool empty() const { return v_ == nullptr; }

    //! DBoW2: transform(features, v, fv, levelsup) with one 1 x 32 cv::Mat per feature
    void transform(const std::vector<cv::Mat>& features, bow_vector& v, bow_feature_vector_t& fv, int levelsup) const {
        std::vector<uint8_t> desc(features.size() * 32);
        for (size_t i = 0; i < features.size(); ++i) std::copy(features[i].data, features[i].data + 32, &desc[i * 32]);
        run(desc.data(), (int)features.size(), levelsup, v, fv);
    }
    //! FBoW: transform(descriptors (N x 32), level, v, fv)
    void transform(const cv::Mat& descriptors, int levelsup, bow_vector& v, bow_feature_vector_t& fv) const {
        std::vector<uint8_t> desc((size_t)descriptors.rows * 32);
        for (int i = 0; i < descriptors.rows; ++i) std::copy(descriptors.ptr(i), descriptors.ptr(i) + 32, &desc[(size_t)i * 32]);
        run(desc.data(), descriptors.rows, levelsup, v, fv);
    }

private:
    void load(const std::string& filename) {
        if (v_) ovs_vocab_destroy(v_);
        v_ = nullptr;
        const int st = ovs_vocab_load_file(0, filename.c_str(), 16384, &v_, &format_);
        if (st != OVS_OK) throw std::runtime_error("bow_vocabulary: cannot load " + filename + " (" + std::to_string(st) + "): " + ovs_last_error());
    }
    void run(const uint8_t* desc, int n, int levelsup, bow_vector& v, bow_feature_vector_t& fv) const {
        v.clear();
        fv.clear();
        if (!v_) throw std::runtime_error("bow_vocabulary: no vocabulary loaded");
        if (n == 0) return;
        std::vector<int32_t> word((size_t)n), node((size_t)n);
        std::vector<double> weight((size_t)n);
        // failure policy (util/device_policy.h): one retry, then empty vectors (a frame without words matches no keyframe by BoW)
        if (!util::run_guarded("ovs_bow_transform", [&] { return ovs_bow_transform(v_, desc, n, levelsup, word.data(), weight.data(), node.data()); }, [] {}))
            return;
        for (int i = 0; i < n; ++i) {
            if (!(weight[(size_t)i] > 0)) continue;   // DBoW2: `if (w > 0)` -- stopped words carry weight 0
            v[(unsigned int)word[(size_t)i]] += weight[(size_t)i];
            fv[(unsigned int)node[(size_t)i]].push_back((unsigned int)i);
        }
        double norm = 0.0;
        for (const auto& e : v) norm += std::fabs(e.second);   // L1, in std::map order
        if (norm > 0.0)
            for (auto& e : v) e.second /= norm;
    }
    ovs_vocab* v_ = nullptr;
    int32_t format_ = 0;
};

}   // namespace data
}   // namespace openvslam
