// ba_multi.hip -- the natively sharded local-BA linearisation (BASELINE north star: "keyframe observations are partitioned
// across up to 8 GPUs with RCCL all-reduce of the per-block JtJ / Jtr over xGMI, the sparse Schur solve staying on the host").
// One process drives n_gpus devices (what mapping_module, a single thread of a single process, would do): the edges are partitioned by
// KEYFRAME into n_gpus contiguous keyframe blocks, device d holds an ovs_ba_graph of its shard, a call uploads the state (0.48 MB at
// config 5) to every device, linearises the shards concurrently on per-device streams and sums the landmark blocks with ONE packed
// ncclAllReduce of Hll | bl | chi2 (1.92 MB at 20 k landmarks; pose blocks Hpp | bp and the per-edge Hpl are complete on the shard that
// owns the keyframe and are simply collected -- no collective). RCCL is bound lazily with dlopen, so libovslam_hip.so has no link-time
// dependency on it and a process that never shards never loads it (a torch process keeps using its own copy).
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "ovs_common.h"

namespace {

using ovs::set_last_error;

// the handful of RCCL entry points used here (rccl.h: ncclFloat64 = 8, ncclSum = 0, ncclSuccess = 0)
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool load() {
        if (lib) return true;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)   // prefer a copy the process already holds (torch ships its own)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!lib)
            for (const char* n : names)
                if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        return CommInitAll && CommDestroy && AllReduce && GroupStart && GroupEnd;
    }
};
Rccl g_rccl;

// Direct exchange over xGMI ("measure both"): every device sums the packed buffers of ALL devices itself, reading the peers'
// copies through peer-mapped pointers -- one hop, no ring, and a FIXED summation order (device 0, 1, ...), so every device holds bit-identical
// sums that do not change from run to run. 16-byte loads; the buffers are 1.92 MB at config 5, i.e. the exchange is latency-bound and a
// one-hop all-read moves (N - 1) x 1.92 MB into each device over N - 1 different links at once.
struct PeerPtrs {
    const double* in[8];
};
__global__ __launch_bounds__(256) void k_peer_sum(PeerPtrs p, int n_dev, size_t n, double* __restrict__ out) {
    const size_t n2 = n >> 1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        double2 a = reinterpret_cast<const double2*>(p.in[0])[i];
        for (int d = 1; d < n_dev; ++d) {
            const double2 b = reinterpret_cast<const double2*>(p.in[d])[i];
            a.x += b.x;
            a.y += b.y;
        }
        reinterpret_cast<double2*>(out)[i] = a;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double a = p.in[0][n - 1];
        for (int d = 1; d < n_dev; ++d) a += p.in[d][n - 1];
        out[n - 1] = a;
    }
}

struct Shard {
    int device = 0;
    int pose_lo = 0, pose_hi = 0;               // keyframes [pose_lo, pose_hi) belong to this shard
    ovs_ba_graph* graph = nullptr;
    std::vector<int32_t> mono_src, stereo_src;   // shard edge -> index in the caller's arrays
    hipStream_t stream = nullptr;
    double *d_poses = nullptr, *d_points = nullptr, *d_pose_blocks = nullptr, *d_packed = nullptr, *d_hpl = nullptr;
    double* h_hpl = nullptr;                     // pinned
    void* comm = nullptr;
    // direct exchange: the summed blocks (the peers keep reading d_packed while this device writes its sum), and the two events that order
    // the devices against each other: "my shard's blocks are complete" / "I have finished reading everybody's blocks"
    double* d_sum = nullptr;
    hipEvent_t ev_lin = nullptr, ev_sum = nullptr;
    bool sum_pending = false;
};

}   // namespace

struct ovs_ba_multi {
    int n_gpus = 0, n_pose = 0, n_pt = 0, n_mono = 0, n_stereo = 0;
    std::vector<Shard> shards;
    double* h_packed = nullptr;   // pinned: Hll | bl | chi2 from device 0
    double* h_pose = nullptr;     // pinned: per-shard Hpp | bp blocks
    int exchange = OVS_BA_EXCHANGE_RCCL;
    bool peer_ok = false;         // every pair of devices has peer access enabled
};

extern "C" {

ovs_status ovs_ba_multi_destroy(ovs_ba_multi* m) {
    if (!m) return OVS_OK;
    for (Shard& s : m->shards) {
        (void)hipSetDevice(s.device);
        if (s.stream) hipStreamSynchronize(s.stream);
        if (s.comm && g_rccl.CommDestroy) g_rccl.CommDestroy(s.comm);
        if (s.graph) ovs_ba_graph_destroy(s.graph);
        hipFree(s.d_poses);
        hipFree(s.d_points);
        hipFree(s.d_pose_blocks);
        hipFree(s.d_packed);
        hipFree(s.d_hpl);
        hipFree(s.d_sum);
        if (s.ev_lin) hipEventDestroy(s.ev_lin);
        if (s.ev_sum) hipEventDestroy(s.ev_sum);
        if (s.h_hpl) hipHostFree(s.h_hpl);
        if (s.stream) hipStreamDestroy(s.stream);
    }
    if (m->h_packed) hipHostFree(m->h_packed);
    if (m->h_pose) hipHostFree(m->h_pose);
    delete m;
    return OVS_OK;
}

ovs_status ovs_ba_multi_create(int32_t n_gpus, int32_t n_pose, const uint8_t* pose_fixed, int32_t n_pt, const ovs_ba_edge* mono, int32_t n_mono,
                               const ovs_ba_edge_stereo* stereo, int32_t n_stereo, const ovs_ba_cam* cam, double focal_x_baseline,
                               ovs_ba_multi** out) {
    if (!out || !cam || n_gpus < 1 || n_gpus > 8 || n_pose < 1 || n_pt < 1 || n_mono < 0 || n_stereo < 0 || (n_mono > 0 && !mono) ||
        (n_stereo > 0 && !stereo))
        return OVS_ERR_INVALID;
    *out = nullptr;
    if (ovs_device_count() < n_gpus) return OVS_ERR_NO_DEVICE;
    for (int i = 0; i < n_mono; ++i)
        if (mono[i].pose_idx < 0 || mono[i].pose_idx >= n_pose) return OVS_ERR_INVALID;
    for (int i = 0; i < n_stereo; ++i)
        if (stereo[i].pose_idx < 0 || stereo[i].pose_idx >= n_pose) return OVS_ERR_INVALID;
    ovs_ba_multi* m = new (std::nothrow) ovs_ba_multi();
    if (!m) return OVS_ERR_INVALID;
    m->n_gpus = n_gpus;
    m->n_pose = n_pose;
    m->n_pt = n_pt;
    m->n_mono = n_mono;
    m->n_stereo = n_stereo;
    m->shards.resize((size_t)n_gpus);
    const int per = (n_pose + n_gpus - 1) / n_gpus;   // contiguous keyframe blocks (2000 edges per keyframe at config 5: balanced)
#define M_TRY(expr)                            \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) {                \
            set_last_error(#expr, _e);         \
            ovs_ba_multi_destroy(m);           \
            return OVS_ERR_HIP;                \
        }                                      \
    } while (0)
    for (int d = 0; d < n_gpus; ++d) {
        Shard& s = m->shards[(size_t)d];
        s.device = d;
        s.pose_lo = std::min(n_pose, d * per);
        s.pose_hi = std::min(n_pose, (d + 1) * per);
        std::vector<ovs_ba_edge> em;
        std::vector<ovs_ba_edge_stereo> es;
        for (int i = 0; i < n_mono; ++i)
            if (mono[i].pose_idx >= s.pose_lo && mono[i].pose_idx < s.pose_hi) {
                em.push_back(mono[i]);
                s.mono_src.push_back(i);
            }
        for (int i = 0; i < n_stereo; ++i)
            if (stereo[i].pose_idx >= s.pose_lo && stereo[i].pose_idx < s.pose_hi) {
                es.push_back(stereo[i]);
                s.stereo_src.push_back(i);
            }
        const ovs_status st = ovs_ba_graph_create(d, n_pose, pose_fixed, n_pt, em.data(), (int32_t)em.size(), es.data(), (int32_t)es.size(), cam,
                                                  focal_x_baseline, &s.graph);
        if (st != OVS_OK) {
            ovs_ba_multi_destroy(m);
            return st;
        }
        M_TRY(hipSetDevice(d));
        M_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        const size_t ne = std::max<size_t>(em.size() + es.size(), 1);
        M_TRY(hipMalloc(&s.d_poses, sizeof(double) * 7 * (size_t)n_pose));
        M_TRY(hipMalloc(&s.d_points, sizeof(double) * 3 * (size_t)n_pt));
        M_TRY(hipMalloc(&s.d_pose_blocks, sizeof(double) * 42 * (size_t)n_pose));
        M_TRY(hipMalloc(&s.d_packed, sizeof(double) * (12 * (size_t)n_pt + 4)));
        M_TRY(hipMalloc(&s.d_hpl, sizeof(double) * 18 * ne));
        M_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.h_hpl), sizeof(double) * 18 * ne, hipHostMallocDefault));
        if (n_gpus > 1) {
            M_TRY(hipMalloc(&s.d_sum, sizeof(double) * (12 * (size_t)n_pt + 4)));
            M_TRY(hipEventCreateWithFlags(&s.ev_lin, hipEventDisableTiming));
            M_TRY(hipEventCreateWithFlags(&s.ev_sum, hipEventDisableTiming));
        }
    }
    if (n_gpus > 1) {   // peer access for the direct exchange; failing to get it only disables that variant
        bool ok = true;
        for (int a = 0; a < n_gpus && ok; ++a) {
            M_TRY(hipSetDevice(a));
            for (int b = 0; b < n_gpus && ok; ++b) {
                if (a == b) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) {
                    ok = false;
                    break;
                }
                const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
                (void)hipGetLastError();
            }
        }
        m->peer_ok = ok;
    }
    M_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_packed), sizeof(double) * (12 * (size_t)n_pt + 4), hipHostMallocDefault));
    M_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->h_pose), sizeof(double) * 42 * (size_t)n_pose * (size_t)n_gpus, hipHostMallocDefault));
#undef M_TRY
    if (n_gpus > 1) {
        if (!g_rccl.load()) {
            ovs_ba_multi_destroy(m);
            return OVS_ERR_NO_DEVICE;   // no RCCL: a sharded run is impossible (ovs_last_error stays empty: not a HIP failure)
        }
        std::vector<void*> comms((size_t)n_gpus, nullptr);
        std::vector<int> devs((size_t)n_gpus);
        for (int d = 0; d < n_gpus; ++d) devs[(size_t)d] = d;
        if (g_rccl.CommInitAll(comms.data(), n_gpus, devs.data()) != 0) {
            ovs_ba_multi_destroy(m);
            return OVS_ERR_HIP;
        }
        for (int d = 0; d < n_gpus; ++d) m->shards[(size_t)d].comm = comms[(size_t)d];
    }
    *out = m;
    return OVS_OK;
}

ovs_status ovs_ba_multi_linearize(ovs_ba_multi* m, const double* poses, const double* points, double huber_mono, double huber_stereo, double* Hpp,
                                  double* bp, double* Hll, double* bl, double* Hpl, double* chi2) {
    if (!m || !poses || !points || !Hpp || !bp || !Hll || !bl || !Hpl || !chi2) return OVS_ERR_INVALID;
    const size_t np = (size_t)m->n_pose, npt = (size_t)m->n_pt;
    const bool peer = m->n_gpus > 1 && m->exchange == OVS_BA_EXCHANGE_PEER;
    if (peer && !m->peer_ok) return OVS_ERR_NO_DEVICE;
    // 1. state to every device, shard linearisations (asynchronous, one stream per device)
    for (Shard& s : m->shards) {
        OVS_HIP_TRY(hipSetDevice(s.device));
        // a peer may still be reading this device's blocks for the previous call's sum
        for (Shard& o : m->shards)
            if (o.sum_pending && &o != &s) OVS_HIP_TRY(hipStreamWaitEvent(s.stream, o.ev_sum, 0));
        OVS_HIP_TRY(hipMemcpyAsync(s.d_poses, poses, sizeof(double) * 7 * np, hipMemcpyHostToDevice, s.stream));
        OVS_HIP_TRY(hipMemcpyAsync(s.d_points, points, sizeof(double) * 3 * npt, hipMemcpyHostToDevice, s.stream));
        const ovs_status st = ovs_ba_graph_linearize_dev(s.graph, s.d_poses, s.d_points, huber_mono, huber_stereo, s.d_pose_blocks,
                                                         s.d_pose_blocks + 36 * np, s.d_packed, s.d_packed + 9 * npt, s.d_hpl,
                                                         s.d_packed + 12 * npt, s.stream);
        if (st != OVS_OK) return st;
    }
    for (Shard& s : m->shards) s.sum_pending = false;
    // 2. THE exchange step: Hll | bl | chi2[2] summed across the devices (max |diag| is not a sum: excluded)
    const double* packed_of_dev0 = m->shards[0].d_packed;
    if (peer) {
        const size_t n = 12 * npt + 2;
        PeerPtrs pp{};
        for (size_t d = 0; d < m->shards.size(); ++d) pp.in[d] = m->shards[d].d_packed;
        for (Shard& s : m->shards) {
            OVS_HIP_TRY(hipSetDevice(s.device));
            OVS_HIP_TRY(hipEventRecord(s.ev_lin, s.stream));
        }
        for (Shard& s : m->shards) {
            OVS_HIP_TRY(hipSetDevice(s.device));
            for (Shard& o : m->shards)
                if (&o != &s) OVS_HIP_TRY(hipStreamWaitEvent(s.stream, o.ev_lin, 0));
            const unsigned blocks = (unsigned)std::min<size_t>((n / 2 + 255) / 256, 1024);
            hipLaunchKernelGGL(k_peer_sum, dim3(std::max(blocks, 1u)), dim3(256), 0, s.stream, pp, m->n_gpus, n, s.d_sum);
            OVS_HIP_TRY(hipGetLastError());
            OVS_HIP_TRY(hipEventRecord(s.ev_sum, s.stream));
            s.sum_pending = true;
        }
        packed_of_dev0 = m->shards[0].d_sum;
    } else if (m->n_gpus > 1) {
        if (g_rccl.GroupStart() != 0) return OVS_ERR_HIP;
        for (Shard& s : m->shards) {
            OVS_HIP_TRY(hipSetDevice(s.device));
            if (g_rccl.AllReduce(s.d_packed, s.d_packed, 12 * npt + 2, /*ncclFloat64*/ 8, /*ncclSum*/ 0, s.comm, s.stream) != 0) return OVS_ERR_HIP;
        }
        if (g_rccl.GroupEnd() != 0) return OVS_ERR_HIP;
    }
    // 3. collect: landmark blocks from device 0, pose blocks and Hpl from their owners
    for (size_t d = 0; d < m->shards.size(); ++d) {
        Shard& s = m->shards[d];
        OVS_HIP_TRY(hipSetDevice(s.device));
        if (d == 0) OVS_HIP_TRY(hipMemcpyAsync(m->h_packed, packed_of_dev0, sizeof(double) * (12 * npt + 2), hipMemcpyDeviceToHost, s.stream));
        OVS_HIP_TRY(hipMemcpyAsync(m->h_pose + 42 * np * d, s.d_pose_blocks, sizeof(double) * 42 * np, hipMemcpyDeviceToHost, s.stream));
        const size_t ne = s.mono_src.size() + s.stereo_src.size();
        if (ne) OVS_HIP_TRY(hipMemcpyAsync(s.h_hpl, s.d_hpl, sizeof(double) * 18 * ne, hipMemcpyDeviceToHost, s.stream));
    }
    for (Shard& s : m->shards) {
        OVS_HIP_TRY(hipSetDevice(s.device));
        OVS_HIP_TRY(hipStreamSynchronize(s.stream));
    }
    std::memcpy(Hll, m->h_packed, sizeof(double) * 9 * npt);
    std::memcpy(bl, m->h_packed + 9 * npt, sizeof(double) * 3 * npt);
    chi2[0] = m->h_packed[12 * npt];
    chi2[1] = m->h_packed[12 * npt + 1];
    for (size_t d = 0; d < m->shards.size(); ++d) {
        const Shard& s = m->shards[d];
        const double* hp = m->h_pose + 42 * np * d;
        for (int k = s.pose_lo; k < s.pose_hi; ++k) {
            std::memcpy(Hpp + 36 * (size_t)k, hp + 36 * (size_t)k, sizeof(double) * 36);
            std::memcpy(bp + 6 * (size_t)k, hp + 36 * np + 6 * (size_t)k, sizeof(double) * 6);
        }
        for (size_t i = 0; i < s.mono_src.size(); ++i) std::memcpy(Hpl + 18 * (size_t)s.mono_src[i], s.h_hpl + 18 * i, sizeof(double) * 18);
        for (size_t i = 0; i < s.stereo_src.size(); ++i)
            std::memcpy(Hpl + 18 * ((size_t)m->n_mono + (size_t)s.stereo_src[i]), s.h_hpl + 18 * (s.mono_src.size() + i), sizeof(double) * 18);
    }
    return OVS_OK;
}

ovs_status ovs_ba_multi_set_exchange(ovs_ba_multi* m, int32_t exchange) {
    if (!m || (exchange != OVS_BA_EXCHANGE_RCCL && exchange != OVS_BA_EXCHANGE_PEER)) return OVS_ERR_INVALID;
    if (exchange == OVS_BA_EXCHANGE_PEER && m->n_gpus > 1 && !m->peer_ok) return OVS_ERR_NO_DEVICE;   // no peer access between some pair
    m->exchange = exchange;
    return OVS_OK;
}

}   // extern "C"
