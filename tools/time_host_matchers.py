"""Wall time of the host-pointer matcher entry points (config 0 area match, config 3 projection match) over many calls; run under
rocprofv3 --kernel-trace --stats to split kernel time from API / copy overhead. Usage (GPU box): python tools/time_host_matchers.py"""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvslam_amd import feature, match, synth

a = synth.synth_frame(480, 752, seed=0)
b = synth.synth_frame(480, 752, seed=0, shift=(5, 0), noise_seed=4242)
ex = feature.orb_extractor(feature.orb_params(1000), max_rows=480, max_cols=752)
ka, da = ex.extract(a)
kb, db = ex.extract(b)
gp = match.grid_params(752, 480)
am = match.area(0.9, True, max_targets=2048, max_queries=2048)
prev = np.ascontiguousarray(np.stack([ka["x"], ka["y"]], 1), np.float32)
N = 200
am.match_in_consistent_area(gp, ka, da, kb, db, prev.copy(), 100)
t = time.perf_counter()
for _ in range(N):
    am.match_in_consistent_area(gp, ka, da, kb, db, prev.copy(), 100)
print("area match (1000 x 1000, margin 100): %.3f ms per call" % ((time.perf_counter() - t) / N * 1e3))
k, d = synth.synth_keypoints(4000, 1920, 3840, seed=1)
lm = synth.synth_landmarks(k, d, 10000, 1920, 3840, seed=2, n_from_frame=5200)
sf = np.cumprod(np.concatenate([[1.0], np.full(7, 1.2)]).astype(np.float32)).astype(np.float32)
pm = match.projection(0.8, True, max_targets=4096, max_queries=16384)
gp4 = match.grid_params(3840, 1920)
pm.match_frame_and_landmarks(gp4, k, d, sf, lm["xy"], lm["level"], lm["desc"], 5.0, lm_valid=lm["valid"])
t = time.perf_counter()
for _ in range(N):
    pm.match_frame_and_landmarks(gp4, k, d, sf, lm["xy"], lm["level"], lm["desc"], 5.0, lm_valid=lm["valid"])
print("projection match (4000 keypoints, 10000 landmarks): %.3f ms per call" % ((time.perf_counter() - t) / N * 1e3))
