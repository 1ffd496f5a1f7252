// What the class shims do when the device fails.
//
// Upstream's hot-path functions (orb_extractor::extract, the match:: classes, pose_optimizer::optimize, local_bundle_adjuster::optimize)
// return void or a count and cannot fail; tracking_module and mapping_module have no place to catch anything. The C ABI behind the shims
// can: a HIP error, a lost device, an allocation that does not fit. The policy, applied by run_guarded() around every ABI call of a shim,
// separates what the DEVICE did (degrade, keep the session alive) from what the CALLER did (a deterministic error: surface it):
//
//   1. OVS_OK                      -> the result, as always.
//   2. OVS_ERR_HIP                 -> the per-thread device contexts the call used are dropped (handles destroyed, a frame's device cache
//                                     released; they are rebuilt on the next use) and the call is issued ONCE more. A transient failure --
//                                     an allocation that did not fit while another module held its peak, a stream broken by a previous
//                                     fault -- ends here and the caller sees the normal result.
//   3. still OVS_ERR_HIP, or OVS_ERR_NO_DEVICE
//                                  -> the failure is logged (rate limited) and the function returns its EMPTY result: zero keypoints,
//                                     zero matches, zero inliers with the pose untouched, a local map left as it was. That is a state
//                                     upstream already handles -- a frame without features fails to track and tracking_module falls
//                                     back to relocalisation; a local BA that did nothing is a local BA that was aborted --, so a dead
//                                     GPU degrades the SLAM session instead of terminating the process with an exception nobody catches.
//   4. OVS_ERR_CAPACITY            -> the context was created too small for this problem (a candidate list beyond the matcher's entry
//                                     budget, ...): `grow` is asked to enlarge it; if it can, the call is issued once more with the larger
//                                     context. If it cannot, or the larger one is still too small, std::length_error is thrown: the
//                                     same inputs would fail the same way on every frame, and a tracker that silently never matches is
//                                     worse than a process that says why it stopped.
//   5. OVS_ERR_INVALID / OVS_ERR_ALIGN (a refused argument, an unsupported geometry, a duplicate graph edge)
//                                  -> std::invalid_argument: a programming or configuration error, as upstream's own asserts are.
//
// There is no CPU fallback and none is wanted: the checker under oracle/ is not part of the product.
// device_failures() counts every outcome, per status, for an integration's health endpoint.
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
    std::atomic<unsigned long> retried{0};        // ... of which were issued a second time (contexts rebuilt, or grown)
    std::atomic<unsigned long> recovered{0};      // ... of which then succeeded
    std::atomic<unsigned long> degraded{0};       // calls that ended with the empty result (device failures only)
    std::atomic<unsigned long> surfaced{0};       // calls that ended in an exception (deterministic caller-side errors)
    std::atomic<unsigned long> by_status[8];      // first status of every failed call, indexed by -status (1 = INVALID ... 5 = ALIGN)
    device_failure_counters() {
        for (auto& c : by_status) c.store(0);
    }
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

// call: () -> ovs_status (re-evaluates its handle look-ups every time it runs); reset: () -> void, drops the contexts `call` uses;
// grow: () -> bool, enlarges the contexts after OVS_ERR_CAPACITY (false: nothing left to enlarge).
// Returns true when the outputs of `call` are valid, false when the caller has to return its empty result; throws for caller-side errors.
template <class Call, class Reset, class Grow>
bool run_guarded(const char* what, Call&& call, Reset&& reset, Grow&& grow) {
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
    if (st < 0 && st > -8) ++c.by_status[-st];
    const int first = st;
    const std::string first_msg = msg;
    if (st == OVS_ERR_HIP) {
        ++c.retried;
        reset();
        st = once();
        if (st == OVS_OK) {
            ++c.recovered;
            detail::log_failure(what, first, first_msg, "contexts rebuilt, the retry succeeded");
            return true;
        }
    } else if (st == OVS_ERR_CAPACITY && grow()) {
        ++c.retried;
        st = once();
        if (st == OVS_OK) {
            ++c.recovered;
            detail::log_failure(what, first, first_msg, "context enlarged, the retry succeeded");
            return true;
        }
    }
    if (st == OVS_ERR_HIP || st == OVS_ERR_NO_DEVICE) {
        ++c.degraded;
        detail::log_failure(what, st, msg, "returning the empty result");
        return false;
    }
    ++c.surfaced;
    detail::log_failure(what, st, msg, "a caller-side error: throwing");
    const std::string text = std::string(what) + " (status " + std::to_string(st) + ")" + (msg.empty() ? "" : ": " + msg);
    if (st == OVS_ERR_CAPACITY) throw std::length_error(text);
    throw std::invalid_argument(text);
}
template <class Call, class Reset>
bool run_guarded(const char* what, Call&& call, Reset&& reset) {
    return run_guarded(what, call, reset, [] { return false; });
}

}   // namespace util
}   // namespace openvslam
