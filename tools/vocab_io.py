#!/usr/bin/env python3
"""Writers for the three on-disk ORB vocabulary formats data::bow_vocabulary loads (DBoW2 text, the DBoW2 fork's binary .dbow2, FBoW .fbow),
used as the committed fixture generator: the real orb_vocab.dbow2 / orb_vocab.fbow are not in the container, so tests write a synthetic
vocabulary (openvslam_amd.synth.synth_vocabulary) in each format and read it back through ovs_vocab_tree_load / ovs_vocab_load_file.
Layouts as restated in openvslam_amd/csrc/bow_vocab_io.hip. A vocab = dict(child_start, children, desc [n, 32], weight, word_id, depth).

usage: tools/vocab_io.py <format: text|dbow2|fbow> <out path> [k depth seed]"""
import struct
import sys

import numpy as np


def _parents(vocab):
    n = len(vocab["word_id"])
    parent = -np.ones(n, np.int64)
    cs, ch = vocab["child_start"], vocab["children"]
    for p in range(n):
        parent[ch[cs[p]:cs[p + 1]]] = p
    return parent


def _check_dbow2_order(vocab):
    """DBoW2 assigns node ids in creation (file) order and word ids in order of the leaves: the tree must already be numbered that way."""
    parent = _parents(vocab)
    n = len(parent)
    assert (parent[1:] < np.arange(1, n)).all() and parent[0] == -1
    leaves = np.flatnonzero(np.diff(vocab["child_start"]) == 0)
    assert np.array_equal(vocab["word_id"][leaves], np.arange(len(leaves)))
    k = int(np.diff(vocab["child_start"]).max())
    return parent, k


def write_dbow2_text(path, vocab):
    parent, k = _check_dbow2_order(vocab)
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (k, vocab["depth"], 0, 0))   # scoring L1_NORM = 0, weighting TF_IDF = 0
        for i in range(1, len(parent)):
            leaf = vocab["word_id"][i] >= 0
            f.write("%d %d %s %s\n" % (parent[i], 1 if leaf else 0, " ".join(str(int(b)) for b in vocab["desc"][i]), repr(float(vocab["weight"][i]))))


def write_dbow2_binary(path, vocab, header_counts_root=True):
    """header_counts_root: the fork's saveToBinaryFile writes nb_nodes = m_nodes.size() INCLUDING the root and then nb_nodes - 1 records
    (recalled; the default). False writes the record count itself (this generator's convention before round 3; the loader takes both)."""
    parent, k = _check_dbow2_order(vocab)
    n = len(parent)
    with open(path, "wb") as f:
        f.write(struct.pack("<IIiiii", n if header_counts_root else n - 1, 41, k, vocab["depth"], 0, 0))
        for i in range(1, n):
            f.write(struct.pack("<i", int(parent[i])) + bytes(vocab["desc"][i]) + struct.pack("<f", float(vocab["weight"][i]))
                    + (b"\x01" if vocab["word_id"][i] >= 0 else b"\x00"))


def write_fbow(path, vocab, alignment=8):
    """Blocks in breadth-first order: block 0 = the root's children; a block holds the children of one node."""
    cs, ch = vocab["child_start"], vocab["children"]
    k = int(np.diff(cs).max())
    desc_wp = -(-32 // alignment) * alignment
    feature_off = -(-4 // alignment) * alignment                  # after {u16 N, u16 is_leaf}
    child_off = feature_off + k * desc_wp
    block_size = -(-(child_off + k * 8) // alignment) * alignment
    blocks, queue = [], [0]
    block_of = {}
    while queue:                                                   # assign block ids breadth first
        node = queue.pop(0)
        if cs[node + 1] > cs[node]:
            block_of[node] = len(blocks)
            blocks.append(node)
            queue.extend(int(c) for c in ch[cs[node]:cs[node + 1]])
    data = bytearray(block_size * len(blocks))
    for b, node in enumerate(blocks):
        kids = [int(c) for c in ch[cs[node]:cs[node + 1]]]
        all_leaf = all(cs[c + 1] == cs[c] for c in kids)
        off = b * block_size
        struct.pack_into("<HH", data, off, len(kids), 1 if all_leaf else 0)
        for s, c in enumerate(kids):
            data[off + feature_off + s * desc_wp: off + feature_off + s * desc_wp + 32] = bytes(vocab["desc"][c])
            if cs[c + 1] == cs[c]:
                idc = 0x80000000 | int(vocab["word_id"][c])
            else:
                idc = block_of[c]
            struct.pack_into("<If", data, off + child_off + s * 8, idc, float(vocab["weight"][c]))
    name = b"orb".ljust(50, b"\x00")
    params = struct.pack("<50s2xII4xQQQQQiiI4x", name, alignment, len(blocks), desc_wp, block_size, feature_off, child_off, len(data), 0, 32, k)
    assert len(params) == 120
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", 55824124) + params + bytes(data))


def fbow_node_order(vocab):
    """The node numbering the FBoW reader produces (ids in block order, slots in child order): new index -> source node."""
    cs, ch = vocab["child_start"], vocab["children"]
    order, queue = [0], [0]
    while queue:
        node = queue.pop(0)
        kids = [int(c) for c in ch[cs[node]:cs[node + 1]]]
        if kids:
            order.extend(kids)
            queue.extend(kids)
    return np.array(order)


WRITERS = {"text": write_dbow2_text, "dbow2": write_dbow2_binary, "fbow": write_fbow}

if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from openvslam_amd.synth import synth_vocabulary
    fmt, out = sys.argv[1], sys.argv[2]
    k, depth, seed = (int(v) for v in (sys.argv[3:6] + ["10", "4", "0"][len(sys.argv) - 3:]))
    WRITERS[fmt](out, synth_vocabulary(k=k, depth=depth, seed=seed))
    print("wrote", out)
