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


def _run(exe, rows, cols, nfeat, iters, seed, shift, noise_seed):
    from openvslam_amd.synth import synth_frame
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, "a.raw"), os.path.join(td, "b.raw")
        synth_frame(rows, cols, seed=seed).tofile(a)
        synth_frame(rows, cols, seed=seed, shift=shift, noise_seed=noise_seed).tofile(b)
        out = subprocess.check_output([exe, str(rows), str(cols), str(nfeat), a, b, str(iters)], timeout=600)
    return json.loads(out.decode().strip().splitlines()[-1])


def measure(rows=1080, cols=1920, nfeat=2000, iters=200, scenes=4):
    """The first scene's full report (seed 31, as in rounds 2-3) plus, under "tracking_per_frame_scenes", the tracked-frame stage medians of
    `scenes` different frame pairs and their mean: pose_optimizer's time depends on how many rejected Levenberg-Marquardt trials the frame's
    converged rounds end on (rounding noise: +-20 % between frames of one size), so one frame pair is one sample of it."""
    exe = os.path.join(ROOT, "openvslam_amd", "cpp", "bench_shim")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    first = _run(exe, rows, cols, nfeat, iters, 31, (3, 2), 7)
    per = []
    keys = ("extract_median_ms", "match_current_and_last_frames_median_ms", "pose_optimize_median_ms", "match_frame_and_landmarks_median_ms",
            "frame_total_median_ms", "matches_cl", "pose_inliers")
    for k in range(scenes):
        r = first if k == 0 else _run(exe, rows, cols, nfeat, max(20, iters // 2), 31 + 10 * k, (3 + k, 2 - k), 7 + k)
        t = r.get("tracking_per_frame") or {}
        per.append(dict({"seed": 31 + 10 * k}, **{q: t.get(q) for q in keys}))
    first["tracking_per_frame_scenes"] = per
    ok = [p for p in per if p.get("frame_total_median_ms") is not None]
    if ok:
        first["tracking_per_frame_mean_of_scenes"] = {q: round(sum(p[q] for p in ok) / len(ok), 4) for q in keys[:5]}
    return first


if __name__ == "__main__":
    args = [int(v) for v in sys.argv[1:5]]
    print(json.dumps(measure(*args)))
