"""The data-parallel quad-tree formulation (tools/tree_model.py, mirrored by csrc/orb_tree.hip) == the oracle's std::list
restatement of orb_extractor::distribute_keypoints_via_tree, including output ORDER, on adversarial candidate sets."""
import os
import sys
import zlib

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from tree_model import tree_model  # noqa: E402


def _cands(rng, n, W, H, kind):
    if kind == "uniform":
        pts = set()
        while len(pts) < n:
            pts.add((int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))))
    elif kind == "cluster":
        pts = set()
        cx, cy = int(rng.integers(W // 4, 3 * W // 4)), int(rng.integers(H // 4, 3 * H // 4))
        while len(pts) < n:
            x = int(np.clip(rng.normal(cx, 12), 3, W - 4))
            y = int(np.clip(rng.normal(cy, 12), 3, H - 4))
            pts.add((x, y))
    else:  # mixed
        pts = set()
        while len(pts) < n // 2:
            pts.add((int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))))
        cx, cy = int(rng.integers(W // 4, 3 * W // 4)), int(rng.integers(H // 4, 3 * H // 4))
        while len(pts) < n:
            pts.add((int(np.clip(rng.normal(cx, 20), 3, W - 4)), int(np.clip(rng.normal(cy, 20), 3, H - 4))))
    pts = sorted(pts, key=lambda p: (p[1] // 64, p[0] // 64, p[1], p[0]))   # an emission-like order
    xs = np.array([p[0] for p in pts], np.float32)
    ys = np.array([p[1] for p in pts], np.float32)
    sc = rng.integers(8, 60, size=len(pts)).astype(np.float32)   # few distinct values => many response ties
    return xs, ys, sc


@pytest.mark.parametrize("kind", ["uniform", "cluster", "mixed"])
@pytest.mark.parametrize("shape", [(1882, 1042), (714, 442), (498, 263), (300, 700), (1203, 338), (64, 64)])
def test_model_equals_oracle(oracle, kind, shape):
    W, H = shape
    rng = np.random.default_rng(zlib.crc32(repr((kind, shape)).encode()))
    for n in (1, 2, 3, 17, 200, 1500, 6000):
        if kind != "uniform" and n > 2000:
            continue
        if n > (W - 6) * (H - 6) // 2:
            continue
        xs, ys, sc = _cands(rng, n, W, H, kind)
        for N in (1, 5, 60, 122, 434, 869, 3000):
            want = oracle.distribute_via_tree(xs, ys, sc, 19, 19 + W, 19, 19 + H, N)
            got = tree_model(xs, ys, sc, 19, 19 + W, 19, 19 + H, N)
            assert list(want) == list(got), (kind, shape, n, N)


@pytest.mark.parametrize("kind", ["uniform", "cluster", "mixed"])
@pytest.mark.parametrize("shape", [(1882, 1042), (498, 263), (300, 700), (64, 64)])
def test_sorted_path_code_formulation_equals_oracle(oracle, kind, shape):
    """The candidate formulation for the next kernel version (tools/tree_model.py tree_model_sorted): candidates sorted once by
    (root, path code), every node a range of the sorted array, no per-pass candidate sweeps -- same keypoints in the same order."""
    from tree_model import tree_model_sorted
    W, H = shape
    rng = np.random.default_rng(zlib.crc32(repr(("sorted", kind, shape)).encode()))
    for n in (1, 2, 17, 200, 1500):
        if n > (W - 6) * (H - 6) // 2:
            continue
        xs, ys, sc = _cands(rng, n, W, H, kind)
        for N in (1, 5, 122, 434, 3000):
            want = oracle.distribute_via_tree(xs, ys, sc, 19, 19 + W, 19, 19 + H, N)
            got = tree_model_sorted(xs, ys, sc, 19, 19 + W, 19, 19 + H, N)
            assert list(want) == list(got), (kind, shape, n, N)
