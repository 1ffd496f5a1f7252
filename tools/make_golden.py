#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the CPU oracle.

The reference's own tests hold NO golden vectors for this path and the reference source is absent, so
these fixtures pin the ORACLE (and through it the HIP path) against drift; they do not pin parity with upstream
(PARITY UNPINNED). Re-run only when ORACLE_SPEC.md changes on purpose."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from openvslam_amd.synth import synth_frame  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
os.makedirs(out, exist_ok=True)
# config 1 geometry (EuRoC cam0 752x480, 1000 features): two frames 5 px apart + their brute-force matches
a = synth_frame(480, 752, seed=0)
b = synth_frame(480, 752, seed=0, shift=(5, 0), noise_seed=4242)
ox = ob.OrbExtractor(ob.make_params(1000))
ka, da = ox.extract(a)
cand = [np.stack(ox.level_candidates(l)) for l in range(8)]
kb, db = ox.extract(b)
pairs = ob.robust_brute_force_match(da, db, None, 0.9)
np.savez_compressed(os.path.join(out, "orb_752x480_seed0.npz"), kps_a=ka, desc_a=da, kps_b=kb, desc_b=db, pairs_ab_ratio09=pairs,
                    n_cand=np.array([c.shape[1] for c in cand]), cand_l7=cand[7])
# a small odd-sized frame exercising partial cells on every side
c = synth_frame(203, 331, seed=5)
oc = ob.OrbExtractor(ob.make_params(300))
kc, dc = oc.extract(c)
np.savez_compressed(os.path.join(out, "orb_331x203_seed5.npz"), kps=kc, desc=dc)
print("wrote", len(ka), len(kb), len(pairs), len(kc))
