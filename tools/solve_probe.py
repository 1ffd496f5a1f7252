import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from openvslam_amd import ba
rng = np.random.default_rng(1)
for n in (288, 96, 600):
    q, _ = np.linalg.qr(rng.standard_normal((n, n))); w = np.logspace(0, 6, n); S = (q * w) @ q.T; S = 0.5 * (S + S.T)
    rhs = S @ rng.standard_normal(n)
    for _ in range(3): x = ba.dense_solve(S, rhs)
    print(n, np.abs(x - np.linalg.solve(S, rhs)).max())
