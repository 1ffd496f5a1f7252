"""One linearisation (ovs_ba_graph_linearize_dev through bench.bench_local_ba) at BASELINE config 5 and at the ten times larger map of bench.py's
local_ba_large, in this process's environment (OVS_BA_LM_PER_WG=128|256 forces the landmarks per workgroup of k_linearize). GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

for kw in (dict(), dict(iters=10, n_pose=200, n_pt=100000, obs_per_pose=5000)):
    for rep in range(3):
        r = bench.bench_local_ba(1, 0, None, torch, **kw)
        print("OVS_BA_LM_PER_WG=%s  %s: %.4f ms per linearisation" % (os.environ.get("OVS_BA_LM_PER_WG", "auto"), r["workload"][:60], r["ms_per_linearisation"]), flush=True)
