"""Round-4 probe: local BA with no / one free keyframe, device and host solver against the oracle (iteration counts, chi2, states)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import lba
from openvslam_amd import ba
from test_ba import _lba_scene
d, mono, st, bf, _, _ = _lba_scene(5, n_pose=6, n_pt=600, obs_per_pose=300, stereo_frac=0.2)
for n_free in (0, 1, 2):
    fixed = np.ones(len(d["poses"]), np.uint8)
    fixed[len(fixed) - n_free:] = 0
    want = lba.local_ba_optimize(d["poses"], fixed, d["points"], mono, d["cam"], st, bf)
    for solver in ("device", "host"):
        ba.local_ba_set_solver(solver)
        got = ba.local_ba_optimize(d["poses"], fixed, d["points"], mono, d["cam"], st, bf)
        print(n_free, solver, "iters", got["info"][4:], want["info"][4:], "chi2", got["info"][:4], want["info"][:4],
              "dpose %.2e dpts %.2e" % (np.abs(got["poses"] - want["poses"]).max(), np.abs(got["points"] - want["points"]).max()))
ba.local_ba_set_solver("device")
