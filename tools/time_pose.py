"""Latency of ovs_pose_optimize (host entry, one frame) -- usage (GPU box): python tools/time_pose.py [n_obs] [iters] [stereo_frac]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvslam_amd import ba, synth   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
stereo_frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
T0, obs, cam, bf, _ = synth.synth_pose_frame(ba.POSE_OBS_DTYPE, n, 7, stereo_frac, 0.1, 1.0)
for _ in range(10):
    ba.pose_optimize(T0, obs, cam, bf)
ts = []
for _ in range(iters):
    t = time.perf_counter()
    ba.pose_optimize(T0, obs, cam, bf)
    ts.append(time.perf_counter() - t)
ts = np.array(ts) * 1e3
print("pose_optimize stereo_frac=%.1f n=%d: median %.4f ms, p95 %.4f, min %.4f" % (stereo_frac, n, np.median(ts), np.percentile(ts, 95), ts.min()))
T0e, obse, cols, rows, _ = synth.synth_pose_frame_equirect(ba.POSE_OBS_DTYPE, n, 7)
for _ in range(10):
    ba.pose_optimize_equirect(T0e, obse, cols, rows)
ts = []
for _ in range(iters):
    t = time.perf_counter()
    ba.pose_optimize_equirect(T0e, obse, cols, rows)
    ts.append(time.perf_counter() - t)
ts = np.array(ts) * 1e3
print("pose_optimize_equirect n=%d: median %.4f ms, p95 %.4f, min %.4f" % (n, np.median(ts), np.percentile(ts, 95), ts.min()))
