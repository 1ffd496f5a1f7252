// cv_stub.h -- minimal POD stand-ins for the OpenCV types on the hot path's public surface. OpenCV is not installed in the
// build image; in an OpenVSLAM checkout this header is replaced by <opencv2/core.hpp> and the shims compile
// unchanged (cv::KeyPoint has the same 28-byte layout as ovs_keypoint; cv::Mat exposes data/rows/cols/step).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace cv {

struct Point2f {
    float x = 0, y = 0;
};

struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

constexpr int CV_8U = 0;
constexpr int CV_8UC1 = 0;

// single-channel 8-bit matrix, owning or aliasing
struct Mat {
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t* data = nullptr;
    std::vector<uint8_t> storage;

    Mat() = default;
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8U); }
    Mat(int r, int c, int /*type*/, void* ext, size_t ext_step) : rows(r), cols(c), step(ext_step), data(static_cast<uint8_t*>(ext)) {}
    void create(int r, int c, int /*type*/) {
        if (r == rows && c == cols && !storage.empty()) return;
        storage.assign((size_t)r * c, 0);
        rows = r;
        cols = c;
        step = (size_t)c;
        data = storage.data();
    }
    bool empty() const { return rows == 0 || cols == 0 || data == nullptr; }
    int type() const { return CV_8UC1; }
    uint8_t* ptr(int r) { return data + (size_t)r * step; }
    const uint8_t* ptr(int r) const { return data + (size_t)r * step; }
    Mat row(int r) const { return Mat(1, cols, CV_8U, data + (size_t)r * step, step); }
};

using _InputArray = Mat;    // upstream: const cv::_InputArray&
using _OutputArray = Mat;   // upstream: const cv::_OutputArray&

}   // namespace cv
