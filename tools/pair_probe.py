"""ovs_orb_extract_pair against two ovs_orb_extract calls: identical results, time per stereo frame. Usage (GPU box): python tools/pair_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvslam_amd import feature
from openvslam_amd.synth import synth_frame
rows, cols = 1080, 1920
a = synth_frame(rows, cols, seed=31)
b = synth_frame(rows, cols, seed=31, shift=(9, 0), noise_seed=3)
ex1 = feature.orb_extractor(feature.orb_params(2000), max_rows=rows, max_cols=cols)
ex2 = feature.orb_extractor(feature.orb_params(2000), max_rows=rows, max_cols=cols, max_batch=2)
ka, da = ex1.extract(a)
kb, db = ex1.extract(b)
(pka, pda), (pkb, pdb) = ex2.extract_pair(a, b)
same = all(np.array_equal(x, y) for x, y in ((ka, pka), (da, pda), (kb, pkb), (db, pdb)))
print("pair == two single extracts:", same, len(ka), len(kb))
def med(fn, n=200):
    for _ in range(10): fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    ts.sort(); return ts[n // 2] * 1e3
print("extract(left) %.3f ms; extract(left) + extract(right) sequential %.3f ms; extract_pair %.3f ms" % (
    med(lambda: ex1.extract(a)), med(lambda: (ex1.extract(a), ex1.extract(b))), med(lambda: ex2.extract_pair(a, b))))
m = np.ones((rows, cols), np.uint8); m[:200, :300] = 0
(qa, qda), (qb, qdb) = ex2.extract_pair(a, b, m, m)
ra, rda = ex1.extract(a, m); rb, rdb = ex1.extract(b, m)
print("with masks:", all(np.array_equal(x, y) for x, y in ((ra, qa), (rda, qda), (rb, qb), (rdb, qdb))))
