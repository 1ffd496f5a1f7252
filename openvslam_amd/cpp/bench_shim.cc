// Per-call latency of feature::orb_extractor::extract THROUGH THE CLASS BOUNDARY: caller-owned pageable image in,
// std::vector<cv::KeyPoint> + cv::Mat out, H2D / D2H included. Prints one JSON object.
// usage: bench_shim rows cols nfeat frame_a.raw frame_b.raw iters
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <ovslam_hip.h>

#include "openvslam/feature/orb_extractor.h"

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
        std::printf(", \"two_threads\": {\"solo_ms\": %.4f, \"left_ms\": %.4f, \"right_ms\": %.4f}", solo, l, r);
    }
    std::printf("}\n");
    return 0;
}
