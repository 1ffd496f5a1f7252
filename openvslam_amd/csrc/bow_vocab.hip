// bow_vocab.hip -- SURVEY 8(f) #4: DBoW2 TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(feature, word_id, weight, nid, levelsup),
// the per-descriptor vocabulary-tree descent inside data::frame::compute_bow / data::keyframe::compute_bow (expected call site:
// src/openvslam/data/frame.cc `bow_vocab_->transform(util::converter::to_desc_vec(descriptors_), bow_vec_, bow_feat_vec_, 4)`;
// DBoW2 is an un-vendored third-party dependency: its published algorithm is restated, see oracle/ORACLE_SPEC.md rule 29).
//
// Per descriptor: from the root, at every level take the child with the smallest Hamming distance (the FIRST minimum in child order:
// DBoW2 compares with `d < best_d`), remember the node reached at level L - levelsup, stop at a leaf -> (word id, word weight, node id).
// The BoW vector (sum of weights per word, L1-normalised) and the feature vector (node -> feature indices) are std::maps filled in
// feature order; they stay with the shim, which gets the three numbers per feature from here.
//
// Layout: the vocabulary lives in HBM as a CSR of children with the child descriptors stored in CSR order, so the k children of a node
// are k consecutive 32-byte rows (k = 10 in ORBvoc: 320 B, one or two 256-byte requests). One 16-lane row of a wave per descriptor, one
// lane per child; the minimum over the row is a key reduction on (distance << 8 | child) with row-local shuffles. A frame's 2000
// descriptors make 500 waves; the top levels of the tree (1 + 10 + 100 + 1000 nodes = 36 KB) stay in L2, the two bottom levels are
// random 320-byte reads (1.3 MB per frame) -- the kernel is latency-bound and tiny next to the extraction.
#include <algorithm>
#include <cstring>
#include <vector>

#include "ovs_common.h"

struct ovs_vocab {
    int device = 0;
    int n_nodes = 0, depth = 0, max_children = 0;
    int32_t* d_child_start = nullptr;   // n_nodes + 1
    int32_t* d_children = nullptr;      // CSR payload: node ids
    uint8_t* d_child_desc = nullptr;    // descriptor of children[i] at row i
    double* d_weight = nullptr;         // per node
    int32_t* d_word = nullptr;          // per node, -1 for inner nodes
    // staging for the host entry point
    uint8_t* d_desc = nullptr;
    int32_t *d_out_word = nullptr, *d_out_node = nullptr;
    double* d_out_weight = nullptr;
    int cap = 0;
    hipStream_t stream = nullptr;
};

namespace {

__global__ __launch_bounds__(256) void k_bow_transform(const int32_t* __restrict__ child_start, const int32_t* __restrict__ children,
                                                      const uint8_t* __restrict__ child_desc, const double* __restrict__ node_weight,
                                                      const int32_t* __restrict__ node_word, int depth, int levelsup,
                                                      const uint8_t* __restrict__ desc, const int32_t* __restrict__ counts, int cap,
                                                      int n_single, int32_t* __restrict__ out_word, double* __restrict__ out_weight,
                                                      int32_t* __restrict__ out_node) {
    const int sub = threadIdx.x & 15;
    const int frame = blockIdx.y;
    const int f = blockIdx.x * 16 + (threadIdx.x >> 4);   // feature of this 16-lane row
    const int n = counts ? min(counts[frame], cap) : n_single;
    if (f >= n) return;   // whole rows leave together: the shuffles below stay inside a row
    const size_t slot = (size_t)frame * cap + f;
    uint32_t q[8];
    {
        const uint4* p = reinterpret_cast<const uint4*>(desc + slot * 32);
        const uint4 a = p[0], b = p[1];
        q[0] = a.x, q[1] = a.y, q[2] = a.z, q[3] = a.w, q[4] = b.x, q[5] = b.y, q[6] = b.z, q[7] = b.w;
    }
    const int nid_level = depth - levelsup;
    int node = 0, nid = 0, level = 0;
    for (;;) {
        const int c0 = child_start[node], c1 = child_start[node + 1];
        if (c0 == c1) break;   // leaf
        ++level;
        uint32_t best = 0xFFFFFFFFu;   // (distance << 16) | child position inside the node
        for (int base = c0; base < c1; base += 16) {
            const int i = base + sub;
            uint32_t key = 0xFFFFFFFFu;
            if (i < c1) {
                const uint4* p = reinterpret_cast<const uint4*>(child_desc + (size_t)i * 32);
                const uint4 a = p[0], b = p[1];
                const uint32_t d = (__popc(q[0] ^ a.x) + __popc(q[1] ^ a.y)) + (__popc(q[2] ^ a.z) + __popc(q[3] ^ a.w)) +
                                   (__popc(q[4] ^ b.x) + __popc(q[5] ^ b.y)) + (__popc(q[6] ^ b.z) + __popc(q[7] ^ b.w));
                key = (d << 16) | (uint32_t)(i - c0);
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, off, 16));
            best = min(best, key);   // equal distances: the smaller child position wins = DBoW2's strict `<` in child order
        }
        node = children[c0 + (int)(best & 0xFFFFu)];
        if (level == nid_level) nid = node;
    }
    if (sub == 0) {
        out_word[slot] = node_word[node];
        out_weight[slot] = node_weight[node];
        out_node[slot] = nid_level <= 0 ? 0 : nid;   // DBoW2: levelsup >= L -> the root
    }
}

void vocab_free(ovs_vocab* v) {
    if (!v) return;
    hipFree(v->d_child_start);
    hipFree(v->d_children);
    hipFree(v->d_child_desc);
    hipFree(v->d_weight);
    hipFree(v->d_word);
    hipFree(v->d_desc);
    hipFree(v->d_out_word);
    hipFree(v->d_out_node);
    hipFree(v->d_out_weight);
    if (v->stream) hipStreamDestroy(v->stream);
    delete v;
}

}   // namespace

extern "C" {

ovs_status ovs_vocab_create(int32_t device, int32_t n_nodes, const int32_t* child_start, const int32_t* children, const uint8_t* node_desc,
                            const double* node_weight, const int32_t* node_word_id, int32_t depth, int32_t max_features, ovs_vocab** out) {
    if (!out || n_nodes < 1 || !child_start || !node_desc || !node_weight || !node_word_id || depth < 0 || max_features < 1) return OVS_ERR_INVALID;
    if (child_start[0] != 0) return OVS_ERR_INVALID;
    const int n_edges = child_start[n_nodes];
    if (n_edges < 0 || n_edges >= n_nodes + 1 || (n_edges > 0 && !children)) return OVS_ERR_INVALID;
    int max_children = 0;
    for (int i = 0; i < n_nodes; ++i) {
        const int c = child_start[i + 1] - child_start[i];
        if (c < 0 || c > 65535) return OVS_ERR_INVALID;
        max_children = std::max(max_children, c);
    }
    for (int i = 0; i < n_edges; ++i)
        if (children[i] <= 0 || children[i] >= n_nodes) return OVS_ERR_INVALID;   // node 0 is the root and nobody's child
    if (ovs_device_count() <= device || device < 0) return OVS_ERR_NO_DEVICE;
    OVS_HIP_TRY(hipSetDevice(device));
    ovs_vocab* v = new ovs_vocab();
    v->device = device;
    v->n_nodes = n_nodes;
    v->depth = depth;
    v->max_children = max_children;
    v->cap = max_features;
    std::vector<uint8_t> cd((size_t)32 * std::max(n_edges, 1));
    for (int i = 0; i < n_edges; ++i) std::memcpy(&cd[(size_t)32 * i], node_desc + (size_t)32 * children[i], 32);
    hipError_t e = hipSuccess;
#define V_TRY(expr)                    \
    if ((e = (expr)) != hipSuccess) {  \
        ovs::set_last_error(#expr, e); \
        vocab_free(v);                 \
        return OVS_ERR_HIP;            \
    }
    V_TRY(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    V_TRY(hipMalloc(&v->d_child_start, sizeof(int32_t) * ((size_t)n_nodes + 1)));
    V_TRY(hipMalloc(&v->d_children, sizeof(int32_t) * (size_t)std::max(n_edges, 1)));
    V_TRY(hipMalloc(&v->d_child_desc, cd.size()));
    V_TRY(hipMalloc(&v->d_weight, sizeof(double) * (size_t)n_nodes));
    V_TRY(hipMalloc(&v->d_word, sizeof(int32_t) * (size_t)n_nodes));
    V_TRY(hipMalloc(&v->d_desc, (size_t)32 * max_features));
    V_TRY(hipMalloc(&v->d_out_word, sizeof(int32_t) * (size_t)max_features));
    V_TRY(hipMalloc(&v->d_out_node, sizeof(int32_t) * (size_t)max_features));
    V_TRY(hipMalloc(&v->d_out_weight, sizeof(double) * (size_t)max_features));
    V_TRY(hipMemcpy(v->d_child_start, child_start, sizeof(int32_t) * ((size_t)n_nodes + 1), hipMemcpyHostToDevice));
    if (n_edges) V_TRY(hipMemcpy(v->d_children, children, sizeof(int32_t) * (size_t)n_edges, hipMemcpyHostToDevice));
    V_TRY(hipMemcpy(v->d_child_desc, cd.data(), cd.size(), hipMemcpyHostToDevice));
    V_TRY(hipMemcpy(v->d_weight, node_weight, sizeof(double) * (size_t)n_nodes, hipMemcpyHostToDevice));
    V_TRY(hipMemcpy(v->d_word, node_word_id, sizeof(int32_t) * (size_t)n_nodes, hipMemcpyHostToDevice));
#undef V_TRY
    *out = v;
    return OVS_OK;
}

ovs_status ovs_vocab_destroy(ovs_vocab* v) {
    if (!v) return OVS_ERR_INVALID;
    hipSetDevice(v->device);
    vocab_free(v);
    return OVS_OK;
}

ovs_status ovs_bow_transform_dev(ovs_vocab* v, const uint8_t* d_desc, const int32_t* d_counts, int32_t batch, int32_t cap, int32_t levelsup,
                                 int32_t* d_word_id, double* d_weight, int32_t* d_node_id, void* stream) {
    if (!v || !d_desc || !d_counts || batch < 1 || cap < 1 || levelsup < 0 || !d_word_id || !d_weight || !d_node_id) return OVS_ERR_INVALID;
    hipLaunchKernelGGL(k_bow_transform, dim3((cap + 15) / 16, batch), dim3(256), 0, (hipStream_t)stream, v->d_child_start, v->d_children,
                       v->d_child_desc, v->d_weight, v->d_word, v->depth, levelsup, d_desc, d_counts, cap, 0, d_word_id, d_weight, d_node_id);
    OVS_HIP_TRY(hipGetLastError());
    return OVS_OK;
}

ovs_status ovs_bow_transform(ovs_vocab* v, const uint8_t* desc, int32_t n, int32_t levelsup, int32_t* word_id, double* weight,
                             int32_t* node_id) {
    if (!v || n < 0 || levelsup < 0) return OVS_ERR_INVALID;
    if (n == 0) return OVS_OK;
    if (!desc || !word_id || !weight || !node_id) return OVS_ERR_INVALID;
    if (n > v->cap) return OVS_ERR_CAPACITY;
    OVS_HIP_TRY(hipSetDevice(v->device));
    hipStream_t s = v->stream;
    OVS_HIP_TRY(hipMemcpyAsync(v->d_desc, desc, (size_t)32 * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_bow_transform, dim3((n + 15) / 16, 1), dim3(256), 0, s, v->d_child_start, v->d_children, v->d_child_desc, v->d_weight,
                       v->d_word, v->depth, levelsup, v->d_desc, (const int32_t*)nullptr, v->cap, n, v->d_out_word, v->d_out_weight,
                       v->d_out_node);
    OVS_HIP_TRY(hipGetLastError());
    OVS_HIP_TRY(hipMemcpyAsync(word_id, v->d_out_word, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(weight, v->d_out_weight, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipMemcpyAsync(node_id, v->d_out_node, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    OVS_HIP_TRY(hipStreamSynchronize(s));
    return OVS_OK;
}

}   // extern "C"
