"""Time k_hamming_near / k_bf_resolve alone on synthetic descriptor sets (GPU): random (no near pairs), one exact copy per query,
and clustered near-duplicates. python tools/time_near.py [--batch 128] [--n 2000]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvslam_amd import _lib, match   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    B, n = a.batch, a.n
    L = _lib.lib()
    mt = match.robust(0.9, False, max_n1=n, max_n2=n, max_batch=B, device=0)
    _lib.check(L.ovs_matcher_profile_enable(mt._h, 1), "profile_enable")
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (B, n, 32), dtype=np.uint8)
    other = rng.integers(0, 256, (B, n, 32), dtype=np.uint8)
    flips = base.copy()
    # clustered: every 8 consecutive descriptors differ from their group leader in <= 8 bits
    clus = base.copy()
    for g in range(8):
        noise = np.zeros((B, n // 8, 32), np.uint8)
        idx = rng.integers(0, 32, (B, n // 8))
        np.put_along_axis(noise, idx[..., None], (1 << rng.integers(0, 8, (B, n // 8, 1))).astype(np.uint8), axis=2)
        clus[:, g:(n // 8) * 8:8] = base[:, 0:(n // 8) * 8:8] ^ noise
    cases = {"random_no_near": (base, other), "one_exact_copy": (base, flips[:, ::-1].copy()), "clusters_of_8": (clus, clus[:, ::-1].copy())}
    cnt = torch.full((B,), n, dtype=torch.int32, device="cuda")
    pairs = torch.zeros((B, n, 2), dtype=torch.int32, device="cuda")
    mc = torch.zeros((B,), dtype=torch.int32, device="cuda")
    for name, (d1, d2) in cases.items():
        t1 = torch.from_numpy(d1).cuda()
        t2 = torch.from_numpy(d2).cuda()
        for _ in range(3):
            mt.brute_force_match_batch_dev(t1, cnt, t2, cnt, pairs, mc)
        torch.cuda.synchronize()
        st = (C.c_float * 2)()
        nc = C.c_int(0)
        _lib.check(L.ovs_matcher_profile_read(mt._h, st, C.byref(nc)), "profile_read")
        for _ in range(a.iters):
            mt.brute_force_match_batch_dev(t1, cnt, t2, cnt, pairs, mc)
        torch.cuda.synchronize()
        _lib.check(L.ovs_matcher_profile_read(mt._h, st, C.byref(nc)), "profile_read")
        calls = max(nc.value, 1)
        print(f"{name}: near {st[0] / calls:.4f} ms, resolve {st[1] / calls:.4f} ms, matches/problem {float(mc.float().mean()):.1f}")


if __name__ == "__main__":
    main()
