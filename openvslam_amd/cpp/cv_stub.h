// cv_stub.h -- minimal stand-ins for the OpenCV types on the hot path's public surface. OpenCV is not installed in the build image
// (SURVEY.md 8(c)); in an OpenVSLAM checkout this header is replaced by <opencv2/core.hpp> and the shims compile unchanged:
// cv::KeyPoint has the same 28-byte layout as ovs_keypoint; cv::Mat is a reference-counted header (copies ALIAS the pixels, as
// OpenCV's do) exposing data / rows / cols / step; cv::_InputArray / cv::_OutputArray are the proxy types upstream's signatures
// take by const reference (`const cv::_OutputArray&` + create() const + getMat() const, exactly OpenCV's contract).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
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

// single-channel 8-bit matrix header; owning (shared, reference counted) or aliasing external memory
struct Mat {
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t* data = nullptr;
    std::shared_ptr<std::vector<uint8_t>> storage;   // null for headers over external memory

    Mat() = default;
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8U); }
    Mat(int r, int c, int /*type*/, void* ext, size_t ext_step) : rows(r), cols(c), step(ext_step), data(static_cast<uint8_t*>(ext)) {}
    void create(int r, int c, int /*type*/) {
        if (r == rows && c == cols && data) return;   // OpenCV: no reallocation when the size already fits
        storage = std::make_shared<std::vector<uint8_t>>((size_t)r * c, (uint8_t)0);
        rows = r;
        cols = c;
        step = (size_t)c;
        data = storage->data();
    }
    bool empty() const { return rows == 0 || cols == 0 || data == nullptr; }
    int type() const { return CV_8UC1; }
    bool isContinuous() const { return step == (size_t)cols; }
    uint8_t* ptr(int r) { return data + (size_t)r * step; }
    const uint8_t* ptr(int r) const { return data + (size_t)r * step; }
    Mat row(int r) const { return Mat(1, cols, CV_8U, data + (size_t)r * step, step); }
    Mat clone() const {
        Mat m(rows, cols, CV_8U);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) m.ptr(r)[c] = ptr(r)[c];
        return m;
    }
};

// read-only proxy: upstream's `const cv::_InputArray&` parameters accept a cv::Mat through this implicit conversion
class _InputArray {
public:
    _InputArray() = default;
    _InputArray(const Mat& m) : m_(&m) {}   // NOLINT: implicit, as in OpenCV
    Mat getMat() const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }

private:
    const Mat* m_ = nullptr;
};

// output proxy: create() and getMat() are const because the proxy itself is not modified, only the matrix it refers to
class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}   // NOLINT: implicit, as in OpenCV
    void create(int r, int c, int type) const { m_->create(r, c, type); }
    Mat getMat() const { return *m_; }
    void release() const { *m_ = Mat(); }

private:
    Mat* m_;
};

using InputArray = const _InputArray&;
using OutputArray = const _OutputArray&;

}   // namespace cv
