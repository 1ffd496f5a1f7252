#!/usr/bin/env python3
"""BASELINE configs[0] (752x480, 1000 features): area::match_in_consistent_area, margin 100, both frames resident -- 200 calls, median; run under
rocprofv3 --kernel-trace --stats for the per-kernel split. usage: python tools/area_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvslam_amd import feature, match, synth   # noqa: E402

a = synth.synth_frame(480, 752, seed=0)
b = synth.synth_frame(480, 752, seed=0, shift=(5, 0), noise_seed=4242)
ex = feature.orb_extractor(feature.orb_params(1000), max_rows=480, max_cols=752)
ka, da = ex.extract(a)
kb, db = ex.extract(b)
gp = match.grid_params(752, 480)
am = match.area(0.9, True, max_targets=2048, max_queries=2048)
prev0 = np.ascontiguousarray(np.stack([ka["x"], ka["y"]], 1), np.float32)
fa, fb = match.frame_dev(gp, ka, da), match.frame_dev(gp, kb, db)
ts = []
for i in range(210):
    p = prev0.copy()
    t0 = time.perf_counter()
    n, m = am.match_in_consistent_area(gp, fa, None, fb, None, p, 100)
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[10:])
print("area match resident: median %.1f us, min %.1f us, %d matches" % (ts[len(ts) // 2] * 1e6, ts[0] * 1e6, n))
