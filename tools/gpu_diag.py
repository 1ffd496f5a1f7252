#!/usr/bin/env python3
"""First-contact diagnostics on a GPU box: per-stage GPU-vs-oracle comparison with readable output (not a test)."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from openvslam_amd import _lib, feature, match  # noqa: E402
from openvslam_amd.synth import synth_frame  # noqa: E402


def main():
    L = _lib.lib()
    print("devices", L.ovs_device_count())
    rows, cols, nf = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 752, 1000)
    img = synth_frame(rows, cols, seed=0)
    ex = feature.orb_extractor(feature.orb_params(max_num_keypts=nf), max_rows=rows, max_cols=cols)
    ox = ob.OrbExtractor(ob.make_params(nf))
    t0 = time.time()
    gk, gd = ex.extract(img)
    print("gpu extract", time.time() - t0, len(gk))
    t0 = time.time()
    wk, wd = ox.extract(img)
    print("cpu extract", time.time() - t0, len(wk))
    for l in range(8):
        a, b = ex.image_pyramid(l), ox.level_image(l)
        print("pyr", l, a.shape, b.shape, "mismatch px:", int((a != b).sum()) if a.shape == b.shape else "shape")
    for l in range(8):
        gx, gy, gs = ex.debug_candidates(l)
        wx, wy, ws = ox.level_candidates(l)
        g = set(zip(gx.tolist(), gy.tolist(), gs.tolist()))
        w = set(zip(wx.tolist(), wy.tolist(), ws.tolist()))
        same_order = len(gx) == len(wx) and np.array_equal(gx, wx) and np.array_equal(gy, wy)
        print("cand", l, len(gx), len(wx), "gpu-only", len(g - w), "cpu-only", len(w - g), "order ok", same_order)
        if g != w:
            print("   gpu-only sample", sorted(g - w)[:5], "cpu-only sample", sorted(w - g)[:5])
    print("level counts gpu", ex.debug_level_counts().tolist(), "cpu", [ox.level_num_keypts(l) for l in range(8)])
    n = min(len(gk), len(wk))
    for f in ("x", "y", "size", "angle", "response", "octave"):
        bad = np.nonzero(gk[f][:n].view(np.uint32) != wk[f][:n].view(np.uint32))[0]
        print("kp field", f, "mismatches", len(bad), bad[:5].tolist(),
              [(float(gk[f][i]), float(wk[f][i])) for i in bad[:3]])
    badd = np.nonzero((gd[:n] != wd[:n]).any(axis=1))[0]
    print("desc mismatching rows", len(badd), badd[:10].tolist())
    if len(badd):
        i = badd[0]
        print("  bits differing in row", i, int(np.unpackbits(gd[i] ^ wd[i]).sum()))
    # matcher
    try:
        rng = np.random.default_rng(0)
        d2 = rng.integers(0, 256, size=(2000, 32), dtype=np.uint8)
        d1 = d2[rng.permutation(2000)].copy()
        for i in range(2000):
            bits = np.unpackbits(d1[i]); bits[rng.permutation(256)[:int(rng.integers(0, 60))]] ^= 1; d1[i] = np.packbits(bits)
        m = match.robust(0.9, False, max_n1=2048, max_n2=2048)
        t0 = time.time(); got = m.brute_force_match(d1, d2); tg = time.time() - t0
        t0 = time.time(); want = ob.robust_brute_force_match(d1, d2, None, 0.9); tc = time.time() - t0
        print("match gpu", tg, "cpu", tc, len(got), len(want), "equal", np.array_equal(got, want))
    except Exception:
        traceback.print_exc()


if __name__ == "__main__":
    main()
