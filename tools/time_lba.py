"""Time ovs_local_ba_optimize on BASELINE config 5 (perturbed start). Usage (GPU box): python tools/time_lba.py [device|host] [repetitions]"""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvslam_amd import ba, synth

d = synth.synth_local_ba(seed=0, pose_noise=0.03, point_noise=0.03)
if len(sys.argv) > 1:
    ba.local_ba_set_solver(sys.argv[1])   # device | host
print("solver:", ba.local_ba_get_solver())
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    t = time.perf_counter()
    r = ba.local_ba_optimize(d["poses"], d["pose_fixed"], d["points"], d["edges"], d["cam"])
    print("local_ba_optimize %.1f ms, iterations %s, chi2 %.1f -> %.1f" % ((time.perf_counter() - t) * 1e3, r["info"][4:], r["info"][0], r["info"][3]))
