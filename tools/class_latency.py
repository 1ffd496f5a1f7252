#!/usr/bin/env python3
"""per-call latency of orb_extractor::extract through the C++ class boundary (openvslam_amd/cpp/bench_shim), H2D / D2H
included. Prints the JSON object bench_shim emits. usage: tools/class_latency.py [rows cols nfeat iters]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(rows=1080, cols=1920, nfeat=2000, iters=200):
    from openvslam_amd.synth import synth_frame
    exe = os.path.join(ROOT, "openvslam_amd", "cpp", "bench_shim")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, "a.raw"), os.path.join(td, "b.raw")
        synth_frame(rows, cols, seed=31).tofile(a)
        synth_frame(rows, cols, seed=31, shift=(3, 2), noise_seed=7).tofile(b)
        out = subprocess.check_output([exe, str(rows), str(cols), str(nfeat), a, b, str(iters)], timeout=600)
    return json.loads(out.decode().strip().splitlines()[-1])


if __name__ == "__main__":
    args = [int(v) for v in sys.argv[1:5]]
    print(json.dumps(measure(*args)))
