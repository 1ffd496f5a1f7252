#!/usr/bin/env python3
"""GPU side of tests/test_nversion.py::test_small_tie_heavy_cases_of_every_list_matcher (also a -m gpu test since round 5:
tests/test_gpu_fuzz.py::test_tie_heavy_small_cases): small random cases whose outcome only the tie rules decide -- descriptors from three base patterns,
positions on a 5-px lattice (window edges), a handful of angles -- through the HIP matchers and the CPU oracle. Prints one line per matcher,
exits 1 on any difference. Usage: python tools/tie_fuzz_gpu.py [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as oracle   # noqa: E402  (the checker)
from openvslam_amd import match       # noqa: E402


def run(cases=300, seed=2024):
    """-> {matcher: number of cases whose result differs from the oracle's}"""
    oracle.build()
    rng = np.random.default_rng(seed)
    cols, rows = 200, 120
    gp, ogp = match.grid_params(cols, rows), oracle.grid_params(cols, rows)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    base = rng.integers(0, 256, (3, 32), dtype=np.uint8)

    def frame(n):
        k = np.zeros(n, oracle.KP_DTYPE)
        k["x"] = (rng.integers(0, 41, n) * 5).astype(np.float32)
        k["y"] = (rng.integers(0, 25, n) * 5).astype(np.float32)
        k["octave"] = rng.integers(0, 3, n)
        k["angle"] = rng.choice(np.array([0, 15, 45, 100, 200, 355], np.float32), n)
        d = base[rng.integers(0, 3, n)].copy()
        for j in range(n):
            for b in rng.integers(0, 256, rng.integers(0, 4)):
                d[j, b >> 3] ^= np.uint8(1 << (b & 7))
        return k, d

    def bow(n):
        fv = {}
        for i, node in enumerate(rng.integers(0, 4, n)):
            fv.setdefault(int(node), []).append(i)
        return fv

    bad = {"brute_force": 0, "area": 0, "bow_frame": 0, "bow_keyframes": 0, "frame_and_landmarks": 0}
    ctx = {}
    for case in range(cases):
        n1, n2 = int(rng.integers(1, 41)), int(rng.integers(1, 41))
        k1, d1 = frame(n1)
        k2, d2 = frame(n2)
        ratio = float(rng.choice([0.6, 0.9, 1.0]))
        orient = bool(rng.integers(0, 2))
        key = (ratio, orient)
        if key not in ctx:
            ctx[key] = (match.robust(ratio, orient, max_n1=64, max_n2=64), match.area(ratio, orient, max_targets=64, max_queries=64),
                        match.bow_tree(ratio, orient, max_targets=64, max_queries=64), match.projection(ratio, orient, max_targets=64, max_queries=64))
        m_rob, m_area, m_bow, m_proj = ctx[key]
        v = (rng.random(n2) < 0.8).astype(np.uint8)
        bad["brute_force"] += int(not np.array_equal(m_rob.brute_force_match(d1, d2, v), oracle.robust_brute_force_match(d1, d2, v, ratio)))
        margin = int(rng.choice([5, 10, 20]))
        prev_g = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], 1), np.float32).reshape(-1, 2)
        prev_o = prev_g.copy()
        gn, got = m_area.match_in_consistent_area(gp, k1, d1, k2, d2, prev_g, margin)
        wn, want = oracle.area_match_in_consistent_area(ogp, k1, d1, k2, d2, prev_o, margin, ratio, orient)
        bad["area"] += int(not (gn == wn and np.array_equal(got, want) and np.array_equal(prev_g, prev_o)))
        f1, f2 = bow(n1), bow(n2)
        l1, l2 = (rng.random(n1) < 0.8).astype(np.uint8), (rng.random(n2) < 0.8).astype(np.uint8)
        gn, got = m_bow.match_frame_and_keyframe(k1, d1, f1, k2, d2, f2, l1)
        wn, want = oracle.bow_match_frame_and_keyframe(k1, d1, f1, k2, d2, f2, ratio, orient, l1)
        bad["bow_frame"] += int(not (gn == wn and np.array_equal(got, want)))
        gn, got = m_bow.match_keyframes(k1, d1, f1, k2, d2, f2, l1, l2)
        wn, want = oracle.bow_match_keyframes(k1, d1, f1, k2, d2, f2, ratio, orient, l1, l2)
        bad["bow_keyframes"] += int(not (gn == wn and np.array_equal(got, want)))
        m = int(rng.integers(1, 30))
        lk, ld = frame(m)
        lm_xy = np.ascontiguousarray(np.stack([lk["x"], lk["y"]], 1), np.float32).reshape(-1, 2)
        lvl = lk["octave"].astype(np.int32)
        occ = (rng.random(n1) < 0.1).astype(np.uint8)
        got, gn = m_proj.match_frame_and_landmarks(gp, k1, d1, sf, lm_xy, lvl, ld, float(margin), frm_occupied=occ)
        want, wn = oracle.projection_match_frame_and_landmarks(ogp, k1, d1, sf, lm_xy, lvl, ld, float(margin), ratio, frm_occupied=occ)
        bad["frame_and_landmarks"] += int(not (gn == wn and np.array_equal(got, want)))
    return bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    bad = run(cases)
    for name, n in bad.items():
        print("%-22s %4d cases, %d differ from the oracle" % (name, cases, n))
    sys.exit(1 if any(bad.values()) else 0)


if __name__ == "__main__":
    main()
