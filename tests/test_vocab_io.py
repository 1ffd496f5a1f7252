"""the on-disk vocabulary formats (DBoW2 text, DBoW2-fork binary .dbow2, FBoW .fbow). The real vocabulary files are not
in the container, so tools/vocab_io.py (the committed generator) writes a synthetic vocabulary in each format; the reader must return
the same tree (CPU tier: parsing needs no device) and the device transform of the loaded tree must equal the oracle's on the source tree."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _vocab(k=7, depth=4, seed=5):
    from openvslam_amd.synth import synth_vocabulary
    return synth_vocabulary(k=k, depth=depth, seed=seed)


@pytest.mark.parametrize("fmt,code", [("text", 1), ("dbow2", 2), ("fbow", 3)])
def test_formats_round_trip_on_the_host(tmp_path, fmt, code):
    import vocab_io
    from openvslam_amd import bow
    v = _vocab()
    path = tmp_path / ("orb_vocab." + fmt)
    vocab_io.WRITERS[fmt](str(path), v)
    got, fcode = bow.load_vocabulary_tree(path)
    assert fcode == code and got["depth"] == v["depth"]
    n = len(v["word_id"])
    if fmt == "fbow":   # FBoW numbers inner nodes in block order: compare through that permutation
        order = vocab_io.fbow_node_order(v)
        inv = np.empty(n, np.int64)
        inv[order] = np.arange(n)
        assert len(got["word_id"]) == n
        assert np.array_equal(got["desc"][1:], v["desc"][order][1:])     # the root has no descriptor on disk
        assert np.array_equal(got["word_id"], v["word_id"][order])
        assert np.array_equal(got["weight"][1:], v["weight"][order].astype(np.float32).astype(np.float64)[1:])
        # children of new node i = the source children, renumbered, in the same order
        for i in range(n):
            src = order[i]
            want = inv[v["children"][v["child_start"][src]:v["child_start"][src + 1]]]
            assert np.array_equal(got["children"][got["child_start"][i]:got["child_start"][i + 1]], want)
    else:
        assert np.array_equal(got["child_start"], v["child_start"]) and np.array_equal(got["children"], v["children"])
        assert np.array_equal(got["desc"][1:], v["desc"][1:]) and np.array_equal(got["word_id"], v["word_id"])   # no root descriptor on disk
        w = v["weight"] if fmt == "text" else v["weight"].astype(np.float32).astype(np.float64)   # the binary format stores floats
        assert np.array_equal(got["weight"][1:], w[1:])
    assert (got["weight"][got["word_id"] < 0] >= 0).all()


def test_dbow2_binary_header_conventions(tmp_path):
    """The fork's saveToBinaryFile writes nb_nodes INCLUDING the root and nb_nodes - 1 records (what a real orb_vocab.dbow2 carries, as
    recalled; the generator's default); a header that counts the records is read the same way; any other count, or a file whose size is
    not 24 + 41 * records, is refused."""
    import struct
    import vocab_io
    from openvslam_amd import _lib, bow
    v = _vocab(k=5, depth=3)
    n = len(v["word_id"])
    a, b = tmp_path / "root_counted.dbow2", tmp_path / "records_counted.dbow2"
    vocab_io.write_dbow2_binary(str(a), v)
    vocab_io.write_dbow2_binary(str(b), v, header_counts_root=False)
    ra, rb = a.read_bytes(), b.read_bytes()
    assert struct.unpack_from("<I", ra)[0] == n and struct.unpack_from("<I", rb)[0] == n - 1 and len(ra) == len(rb) == 24 + 41 * (n - 1)
    ga, gb = bow.load_vocabulary_tree(a)[0], bow.load_vocabulary_tree(b)[0]
    for key in ("child_start", "children", "word_id", "weight"):
        assert np.array_equal(ga[key], gb[key]) and (key == "weight" or np.array_equal(ga[key], v[key]))
    assert np.array_equal(ga["desc"], gb["desc"])
    for bad in (n + 1, n - 2, 0):
        c = tmp_path / "bad.dbow2"
        c.write_bytes(struct.pack("<I", bad) + ra[4:])
        with pytest.raises(_lib.OvsError):
            bow.load_vocabulary_tree(c)
    c.write_bytes(ra[:-7])   # a partial last record
    with pytest.raises(_lib.OvsError):
        bow.load_vocabulary_tree(c)


def test_garbage_and_truncated_files_are_refused(tmp_path):
    import vocab_io
    from openvslam_amd import _lib, bow
    p = tmp_path / "junk.bin"
    p.write_bytes(os.urandom(4096))
    with pytest.raises(_lib.OvsError):
        bow.load_vocabulary_tree(p)
    v = _vocab(k=4, depth=3)
    for fmt in ("dbow2", "fbow"):
        q = tmp_path / ("v." + fmt)
        vocab_io.WRITERS[fmt](str(q), v)
        raw = q.read_bytes()
        q.write_bytes(raw[:len(raw) // 2])
        with pytest.raises(_lib.OvsError):
            bow.load_vocabulary_tree(q)
    with pytest.raises(_lib.OvsError):
        bow.load_vocabulary_tree(tmp_path / "does_not_exist.dbow2")
    # FBoW header fields that would wrap a u64 sum / product or ask for far more nodes than the file can hold: refused, no exception
    # crosses the ABI (file offsets: nblocks 64, desc_size_bytes_wp 72, block_size_bytes_wp 80, feature_off_start 88, child_off_start 96,
    # total_size 104, m_k 120)
    import struct
    q = tmp_path / "v.fbow"
    vocab_io.WRITERS["fbow"](str(q), v)
    raw = bytearray(q.read_bytes())
    bow.load_vocabulary_tree(q)
    for off, fmt, val in ((104, "<Q", 2**64 - 64), (88, "<Q", 2**64 - 16), (96, "<Q", 2**64 - 8), (72, "<Q", 2**63), (80, "<Q", 4),
                          (64, "<I", 2**32 - 1), (120, "<I", 65535)):
        bad = bytearray(raw)
        struct.pack_into(fmt, bad, off, val)
        q.write_bytes(bytes(bad))
        with pytest.raises(_lib.OvsError):
            bow.load_vocabulary_tree(q)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["text", "dbow2", "fbow"])
def test_loaded_vocabulary_transforms_like_the_source_tree(oracle, tmp_path, fmt):
    import ctypes as C
    import vocab_io
    from openvslam_amd import _lib, bow
    v = _vocab(k=10, depth=4, seed=9)
    path = tmp_path / ("orb_vocab." + fmt)
    vocab_io.WRITERS[fmt](str(path), v)
    rng = np.random.default_rng(2)
    leaves = np.flatnonzero(v["word_id"] >= 0)
    from openvslam_amd.synth import flip_bits
    desc = np.stack([flip_bits(rng, v["desc"][leaves[rng.integers(len(leaves))]], 30) for _ in range(1500)])
    voc = bow.load_vocabulary(path)
    word, weight, node = voc.transform_features(desc, 4)
    src = dict(v)
    if fmt != "text":
        src["weight"] = v["weight"].astype(np.float32).astype(np.float64)
    ww, wwt, wn = oracle.bow_transform(src, desc, 4)
    assert np.array_equal(word, ww) and np.array_equal(weight, wwt)
    if fmt == "fbow":
        order = vocab_io.fbow_node_order(v)
        assert np.array_equal(order[node], wn)      # same nodes, FBoW numbering
    else:
        assert np.array_equal(node, wn)
    # the one-call C entry: ovs_vocab_load_file
    L = _lib.lib()
    h, f = C.c_void_p(), C.c_int32()
    _lib.check(L.ovs_vocab_load_file(0, str(path).encode(), 4096, C.byref(h), C.byref(f)), "ovs_vocab_load_file")
    assert f.value == {"text": 1, "dbow2": 2, "fbow": 3}[fmt]
    L.ovs_vocab_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["dbow2", "fbow"])
def test_bow_vocabulary_class(oracle, tmp_path, fmt):
    """data::bow_vocabulary through the C++ class: loadFromBinaryFile + both transform() signatures (DBoW2's and FBoW's) fill the
    BowVector / FeatureVector as DBoW2 does (feature order, weight-0 words dropped, L1 normalisation)."""
    import subprocess
    import vocab_io
    from openvslam_amd import bow
    from openvslam_amd.synth import flip_bits
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "openvslam_amd", "cpp")])
    v = _vocab(k=10, depth=4, seed=11)
    path = tmp_path / ("orb_vocab." + fmt)
    vocab_io.WRITERS[fmt](str(path), v)
    rng = np.random.default_rng(4)
    leaves = np.flatnonzero(v["word_id"] >= 0)
    desc = np.stack([flip_bits(rng, v["desc"][leaves[rng.integers(len(leaves))]], 25) for _ in range(1200)])
    desc.tofile(tmp_path / "desc.bin")
    subprocess.check_call([os.path.join(ROOT, "openvslam_amd", "cpp", "test_lba_shim"), "bow", str(path), str(tmp_path / "desc.bin"),
                           str(tmp_path / "bow.bin")])
    raw = (tmp_path / "bow.bin").read_bytes()
    nw, nn = (int(x) for x in np.frombuffer(raw[:8], np.int32))
    off = 8
    got_v = {}
    for _ in range(nw):
        w = int(np.frombuffer(raw[off:off + 4], np.int32)[0])
        got_v[w] = float(np.frombuffer(raw[off + 4:off + 12], np.float64)[0])
        off += 12
    got_f = {}
    for _ in range(nn):
        node, cnt = (int(x) for x in np.frombuffer(raw[off:off + 8], np.int32))
        off += 8
        got_f[node] = [int(x) for x in np.frombuffer(raw[off:off + 4 * cnt], np.int32)]
        off += 4 * cnt
    assert off == len(raw)
    src = dict(v)
    src["weight"] = v["weight"].astype(np.float32).astype(np.float64)
    word, weight, node = oracle.bow_transform(src, desc, 4)
    if fmt == "fbow":
        order = vocab_io.fbow_node_order(v)
        inv = np.empty(len(order), np.int64)
        inv[order] = np.arange(len(order))
        node = inv[node]
    want_v, want_f = bow.assemble(word, weight, node)
    assert got_v == want_v and got_f == want_f and len(want_v) > 100
