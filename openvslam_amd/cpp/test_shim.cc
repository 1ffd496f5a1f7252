// Smoke program of the C++ class shims: reads two raw u8 frames, runs orb_extractor::extract on both and
// robust::brute_force_match between them exactly as tracking code would, and dumps the results for tests/test_cpp_shim.py.
// usage: test_shim rows cols nfeat frame_a.raw frame_b.raw out.bin
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "openvslam/feature/orb_extractor.h"
#include "openvslam/match/robust.h"

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
    std::fclose(f);
    std::printf("shim ok: %u + %u keypoints, %u matches, scale[7]=%f\n", frm.num_keypts_, keyfrm.num_keypts_, n, extractor.get_scale_factors().at(7));
    return 0;
}
