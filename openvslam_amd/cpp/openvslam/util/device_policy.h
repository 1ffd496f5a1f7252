// What the class shims do when the device fails.
//
// Upstream's hot-path functions (orb_extractor::extract, the match:: classes, pose_optimizer::optimize, local_bundle_adjuster::optimize)
// return void or a count and cannot fail; tracking_module and mapping_module have no place to catch anything. The C ABI behind the shims
// can: a HIP error, a lost device, an allocation that does not fit. The policy, applied by run_guarded() around every ABI call of a shim:
//
//   1. OVS_OK                      -> the result, as always.
//   2. OVS_ERR_HIP                 -> the per-thread device contexts the call used are dropped (handles destroyed, a frame's device cache
//                                     released; they are rebuilt on the next use) and the call is issued ONCE more. A transient failure --
//                                     an allocation that did not fit while another module held its peak, a stream broken by a previous
//                                     fault -- ends here and the caller sees the normal result.
//   3. still failing, or any other status (OVS_ERR_NO_DEVICE, OVS_ERR_CAPACITY, ...)
//                                  -> the failure is logged (rate limited) and the function returns its EMPTY result: zero keypoints,
//                                     zero matches, zero inliers with the pose untouched, a local map left as it was. That is a state
//                                     upstream already handles -- a frame without features fails to track and tracking_module falls
//                                     back to relocalisation; a local BA that did nothing is a local BA that was aborted --, so a dead
//                                     GPU degrades the SLAM session instead of terminating the process with an exception nobody catches.
//
// There is no CPU fallback and none is wanted: the checker under oracle/ is not part of the product. Configuration
// that can never work (a fisheye camera model, a schedule other than upstream's) still throws from the constructor or first call:
// that is a programming error, not a run-time failure.
#pragma once
#include <ovslam_hip.h>

#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace openvslam {
namespace util {

struct device_failure_counters {
    std::atomic<unsigned long> failed_calls{0};   // ABI calls that returned a status other than OVS_OK
    std::atomic<unsigned long> retried{0};        // ... of which were issued a second time
    std::atomic<unsigned long> recovered{0};      // ... of which then succeeded
    std::atomic<unsigned long> degraded{0};       // calls that ended with the empty result
};
inline device_failure_counters& device_failures() {
    static device_failure_counters c;
    return c;
}

// Thrown INSIDE a guarded call by the per-thread context holders when a handle cannot be created; caught by run_guarded, never leaves it.
struct device_error : std::runtime_error {
    int status;
    device_error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

namespace detail {
inline void log_failure(const char* what, int st, const std::string& msg, const char* outcome) {
    static std::atomic<unsigned long> n{0};
    const unsigned long k = n++;
    if (k < 8 || (k & 255u) == 0)   // upstream logs through spdlog; the shim layer has only stderr
        std::fprintf(stderr, "[openvslam_amd] %s failed (status %d%s%s): %s%s\n", what, st, msg.empty() ? "" : ": ", msg.c_str(), outcome,
                     k == 7 ? " (further messages: every 256th)" : "");
}
}   // namespace detail

// call: () -> ovs_status (re-evaluates its handle look-ups every time it runs); reset: () -> void, drops the contexts `call` uses.
// Returns true when the outputs of `call` are valid, false when the caller has to return its empty result.
template <class Call, class Reset>
bool run_guarded(const char* what, Call&& call, Reset&& reset) {
    std::string msg;
    auto once = [&]() -> int {
        try {
            const int st = call();
            if (st != OVS_OK) msg = ovs_last_error();
            return st;
        } catch (const device_error& e) {
            msg = e.what();
            return e.status;
        }
    };
    int st = once();
    if (st == OVS_OK) return true;
    device_failure_counters& c = device_failures();
    ++c.failed_calls;
    if (st == OVS_ERR_HIP) {
        ++c.retried;
        reset();
        const int first = st;
        const std::string first_msg = msg;
        st = once();
        if (st == OVS_OK) {
            ++c.recovered;
            detail::log_failure(what, first, first_msg, "contexts rebuilt, the retry succeeded");
            return true;
        }
    }
    ++c.degraded;
    detail::log_failure(what, st, msg, "returning the empty result");
    return false;
}

}   // namespace util
}   // namespace openvslam
