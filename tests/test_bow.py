"""DBoW2 transform: oracle KATs on the CPU, GPU parity through the C ABI."""
import numpy as np
import pytest

from openvslam_amd import synth


def _brute(vocab, d, levelsup):
    """independent numpy restatement of the descent for one descriptor"""
    node, level, nid = 0, 0, 0
    cs, ch = vocab["child_start"], vocab["children"]
    while cs[node] != cs[node + 1]:
        level += 1
        kids = ch[cs[node]:cs[node + 1]]
        dist = np.unpackbits(vocab["desc"][kids] ^ d[None], axis=1).sum(1)
        node = int(kids[int(np.argmin(dist))])      # argmin = first minimum
        if level == vocab["depth"] - levelsup:
            nid = node
    return vocab["word_id"][node], vocab["weight"][node], (0 if vocab["depth"] - levelsup <= 0 else nid)


def test_oracle_descent_by_hand(oracle):
    vocab = synth.synth_vocabulary(k=6, depth=3, seed=2)
    rng = np.random.default_rng(0)
    leaves = np.flatnonzero(vocab["word_id"] >= 0)
    desc = np.stack([synth.flip_bits(rng, vocab["desc"][leaves[i % len(leaves)]], 20) for i in range(300)])
    desc[:20] = rng.integers(0, 256, (20, 32), dtype=np.uint8)
    for levelsup in (0, 1, 2, 3, 5):
        w, wt, nd = oracle.bow_transform(vocab, desc, levelsup)
        for i in range(len(desc)):
            bw, bwt, bnd = _brute(vocab, desc[i], levelsup)
            assert (w[i], wt[i], nd[i]) == (bw, bwt, bnd)
    # levelsup = 0 -> the node is the word's own leaf; levelsup >= depth -> the root
    w, _, nd = oracle.bow_transform(vocab, desc, 0)
    assert np.array_equal(vocab["word_id"][nd], w)
    assert (oracle.bow_transform(vocab, desc, 3)[2] == 0).all()


def test_oracle_first_minimum_on_ties(oracle):
    """two children at the same distance: DBoW2's strict `<` keeps the earlier one"""
    d0 = np.zeros(32, np.uint8)
    a, b = d0.copy(), d0.copy()
    a[0], b[5] = 0x0F, 0xF0          # both at distance 4 from d0
    vocab = dict(child_start=np.array([0, 2, 2, 2], np.int32), children=np.array([1, 2], np.int32), desc=np.stack([d0, a, b]),
                 weight=np.array([0.0, 1.5, 2.5]), word_id=np.array([-1, 0, 1], np.int32), depth=1)
    w, wt, nd = oracle.bow_transform(vocab, d0[None], 0)
    assert (w[0], wt[0], nd[0]) == (0, 1.5, 1)
    vocab["children"] = np.array([2, 1], np.int32)
    w, wt, nd = oracle.bow_transform(vocab, d0[None], 0)
    assert (w[0], wt[0], nd[0]) == (1, 2.5, 2)


def test_assemble_matches_dbow2_rules():
    from openvslam_amd.bow import assemble
    word = np.array([3, 1, 3, 7], np.int32)
    weight = np.array([2.0, 0.0, 1.0, 1.0])
    node = np.array([10, 11, 10, 12], np.int32)
    bv, fv = assemble(word, weight, node)
    assert list(bv) == [3, 7] and bv[3] == 0.75 and bv[7] == 0.25     # weight-0 feature dropped, L1-normalised
    assert fv == {10: [0, 2], 12: [3]}


@pytest.mark.gpu
@pytest.mark.parametrize("k,depth,n", [(10, 4, 2000), (10, 5, 4000), (6, 3, 300), (20, 2, 1000), (10, 4, 1), (10, 4, 0)])
def test_gpu_bow_transform(oracle, k, depth, n):
    from openvslam_amd import bow
    vocab = synth.synth_vocabulary(k=k, depth=depth, seed=k + depth)
    rng = np.random.default_rng(n)
    leaves = np.flatnonzero(vocab["word_id"] >= 0)
    desc = np.zeros((n, 32), np.uint8)
    for i in range(n):
        desc[i] = synth.flip_bits(rng, vocab["desc"][leaves[rng.integers(0, len(leaves))]], 30) if i % 5 else rng.integers(0, 256, 32, dtype=np.uint8)
    v = bow.vocabulary(vocab, max_features=4096)
    for levelsup in (4, 0, 2, 9):
        got = v.transform_features(desc, levelsup)
        want = oracle.bow_transform(vocab, desc, levelsup)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    bv, fv = v.transform(desc, 4)
    wbv, wfv = bow.assemble(*oracle.bow_transform(vocab, desc, 4))
    assert bv == wbv and fv == wfv
    if n > 1:
        w4 = oracle.bow_transform(vocab, desc, 4)
        assert abs(sum(bv.values()) - 1.0) < 1e-12 and sum(len(x) for x in fv.values()) == int((w4[1] > 0).sum())


@pytest.mark.gpu
def test_gpu_bow_transform_chained_from_extractor(oracle):
    """device-resident path: the extractor's batched device outputs feed the transform without visiting the host"""
    import torch
    from openvslam_amd import bow, feature
    vocab = synth.synth_vocabulary(k=10, depth=4, seed=5)
    B, rows, cols = 3, 480, 752
    frames = np.stack([synth.synth_frame(rows, cols, seed=40 + i) for i in range(B)])
    ex = feature.orb_extractor(feature.orb_params(1000), max_rows=rows, max_cols=cols, max_batch=B)
    cap = ex.max_keypoints
    d_img = torch.from_numpy(frames).cuda()
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ex.extract_batch_dev(d_img, d_kps, d_desc, d_cnt, stream=s)
    v = bow.vocabulary(vocab)
    d_word = torch.full((B, cap), -7, dtype=torch.int32, device="cuda")
    d_wt = torch.zeros((B, cap), dtype=torch.float64, device="cuda")
    d_node = torch.full((B, cap), -7, dtype=torch.int32, device="cuda")
    v.transform_batch_dev(d_desc, d_cnt, d_word, d_wt, d_node, levelsup=4, stream=s)
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    for b in range(B):
        n = int(cnt[b])
        want = oracle.bow_transform(vocab, d_desc[b, :n].cpu().numpy(), 4)
        assert n > 500
        assert np.array_equal(d_word[b, :n].cpu().numpy(), want[0]) and np.array_equal(d_wt[b, :n].cpu().numpy(), want[1])
        assert np.array_equal(d_node[b, :n].cpu().numpy(), want[2])
        assert (d_word[b, n:].cpu().numpy() == -7).all()
