"""data::bow_vocabulary (DBoW2 ORB vocabulary) on the MI355X: the per-descriptor tree descent of `transform`.

The vocabulary is a plain tree description (see `synth.synth_vocabulary` for the layout) or one of upstream's vocabulary files: DBoW2 text,
the DBoW2 fork's binary `orb_vocab.dbow2` and FBoW `orb_vocab.fbow` are parsed by `load_vocabulary_tree` / `load_vocabulary`
(`csrc/bow_vocab_io.hip`; layouts restated from the formats' published descriptions, oracle/ORACLE_SPEC.md rule 30 -- no real vocabulary file is
in the container, tests write synthetic ones with `tools/vocab_io.py`)."""
import ctypes as C

import numpy as np

from . import _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class vocabulary:
    """DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> stand-in. vocab = dict(child_start [n_nodes + 1], children, desc [n_nodes, 32],
    weight [n_nodes], word_id [n_nodes] (-1 for inner nodes), depth)."""

    def __init__(self, vocab, max_features=8192, device=0):
        self._L = _lib.lib()
        _lib.require_device()
        self._cs = np.ascontiguousarray(vocab["child_start"], np.int32)
        self._ch = np.ascontiguousarray(vocab["children"], np.int32)
        self._nd = np.ascontiguousarray(vocab["desc"], np.uint8).reshape(-1, 32)
        self._nw = np.ascontiguousarray(vocab["weight"], np.float64)
        self._wi = np.ascontiguousarray(vocab["word_id"], np.int32)
        self.depth = int(vocab["depth"])
        h = C.c_void_p()
        _lib.check(self._L.ovs_vocab_create(device, len(self._wi), _p(self._cs), _p(self._ch), _p(self._nd), _p(self._nw), _p(self._wi),
                                            self.depth, int(max_features), C.byref(h)), "ovs_vocab_create")
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.ovs_vocab_destroy(h)

    def transform_features(self, descriptors, levelsup=4):
        """per feature: (word_id, weight, node_id) -- what the device computes."""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word, weight, node = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1)), np.zeros(max(n, 1), np.int32)
        _lib.check(self._L.ovs_bow_transform(self._h, _p(d), n, int(levelsup), _p(word), _p(weight), _p(node)), "ovs_bow_transform")
        return word[:n].copy(), weight[:n].copy(), node[:n].copy()

    def transform(self, descriptors, levelsup=4):
        """transform(features, bow_vec, bow_feat_vec, levelsup): returns (bow_vec {word: value}, bow_feat_vec {node: [feature indices]}),
        filled in feature order and L1-normalised as DBoW2 does (TF-IDF weighting, L1 scoring)."""
        word, weight, node = self.transform_features(descriptors, levelsup)
        return assemble(word, weight, node)

    def transform_batch_dev(self, d_desc, d_counts, d_word, d_weight, d_node, levelsup=4, stream=None):
        """Device-resident: descriptors (B, cap, 32) uint8 and counts (B,) int32 as orb_extractor.extract_batch_dev leaves them."""
        B, cap = d_desc.shape[0], d_desc.shape[1]
        _lib.check(self._L.ovs_bow_transform_dev(self._h, d_desc.data_ptr(), d_counts.data_ptr(), B, cap, int(levelsup), d_word.data_ptr(),
                                                 d_weight.data_ptr(), d_node.data_ptr(), stream), "ovs_bow_transform_dev")


def assemble(word, weight, node):
    """BowVector / FeatureVector from the per-feature triples, in DBoW2's order of operations."""
    bow_vec, feat_vec = {}, {}
    for i in range(len(word)):
        if weight[i] > 0:
            w = int(word[i])
            bow_vec[w] = bow_vec.get(w, 0.0) + float(weight[i])
            feat_vec.setdefault(int(node[i]), []).append(i)
    norm = 0.0
    for w in sorted(bow_vec):           # std::map order
        norm += abs(bow_vec[w])
    if norm > 0.0:
        for w in bow_vec:
            bow_vec[w] /= norm
    return dict(sorted(bow_vec.items())), dict(sorted(feat_vec.items()))


def load_vocabulary_tree(path):
    """Parse an on-disk ORB vocabulary (DBoW2 text, the DBoW2 fork's binary .dbow2, FBoW .fbow; detected from the content) on the host:
    returns (vocab dict for `vocabulary`, format code 1 / 2 / 3). No device needed (ovs_vocab_tree_load)."""
    L = _lib.lib()
    h, fmt, n, depth = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
    _lib.check(L.ovs_vocab_tree_load(str(path).encode(), C.byref(h), C.byref(fmt), C.byref(n), C.byref(depth)), "ovs_vocab_tree_load")
    try:
        n = n.value
        out = dict(child_start=np.zeros(n + 1, np.int32), children=np.zeros(max(n - 1, 1), np.int32), desc=np.zeros((n, 32), np.uint8),
                   weight=np.zeros(n), word_id=np.zeros(n, np.int32), depth=depth.value)
        _lib.check(L.ovs_vocab_tree_arrays(h, _p(out["child_start"]), _p(out["children"]), _p(out["desc"]), _p(out["weight"]), _p(out["word_id"])),
                   "ovs_vocab_tree_arrays")
        out["children"] = out["children"][:n - 1]
    finally:
        L.ovs_vocab_tree_free(h)
    return out, fmt.value


def load_vocabulary(path, max_features=8192, device=0):
    """data::bow_vocabulary from a file: parse + upload (the C ABI does both in ovs_vocab_load_file; this mirror keeps the host arrays)."""
    tree, _ = load_vocabulary_tree(path)
    return vocabulary(tree, max_features=max_features, device=device)
