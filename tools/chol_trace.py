#!/usr/bin/env python3
"""Phase times of the dense solver (ba_solve.hip) at BASELINE config 5's size: run with OVS_BA_TRACE=1 (the solver's timed instantiation prints thread
0's wall-clock intervals per phase on stderr), with and without OVS_CHOL_RESIDENT=0 for the through-memory kernel. usage: OVS_BA_TRACE=1 python tools/chol_trace.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import sys, numpy as np; sys.path.insert(0, %r); from openvslam_amd import ba; rng = np.random.default_rng(1); "
        "q, _ = np.linalg.qr(rng.standard_normal((288, 288))); S = (q * np.logspace(0, 3, 288)) @ q.T; S = 0.5 * (S + S.T); "
        "[ba.dense_solve(S, np.ones(288)) for _ in range(3)]; ba.dense_solve(S[:96, :96].copy(), np.ones(96))") % ROOT
for env in ({}, {"OVS_CHOL_RESIDENT": "0"}):
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, OVS_BA_TRACE="1", **env), capture_output=True, text=True)
    sys.stderr.write("# %s\n" % (env or "default (k_chol_resident)"))
    sys.stderr.write("".join(ln + "\n" for ln in r.stderr.splitlines() if "dense solve" in ln))
    if r.returncode:
        sys.stderr.write(r.stderr[-1500:])
