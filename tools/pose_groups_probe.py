"""Time ovs_pose_optimize on synthetic frames (OVS_POSE_GROUPS selects the number of workgroups per frame; read once per process).
Usage (GPU box): [OVS_POSE_GROUPS=g] python tools/pose_groups_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvslam_amd import ba, synth
from oracle import binding as ob
for n in (300, 700, 1000, 1300, 2000, 4000):
    meds = []
    for seed in range(7, 15):
        T0, pobs, pcam, pbf, _ = synth.synth_pose_frame(ob.POSE_OBS_DTYPE, n, seed)
        for _ in range(10):
            T, out, nv = ba.pose_optimize(T0, pobs, pcam, pbf)
        ts = []
        for _ in range(60):
            t = time.perf_counter()
            T, out, nv = ba.pose_optimize(T0, pobs, pcam, pbf)
            ts.append(time.perf_counter() - t)
        ts.sort()
        meds.append(ts[30] * 1e3)
    print("groups=%s n=%d mean of 8 frames' medians %.3f ms (min %.3f max %.3f)" % (os.environ.get("OVS_POSE_GROUPS", "auto"), n, sum(meds) / len(meds), min(meds), max(meds)))
