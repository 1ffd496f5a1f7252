"""Where does bench.py's local_ba.ms_per_linearisation go? Per-call wall and HIP-event times of the 20 timed linearisations, (a) in a fresh
process and (b) after the headline workload of bench.py ran in the same process. Diagnostic; prints to stdout."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def probe(tag, iters=20):
    from openvslam_amd import ba
    from openvslam_amd.synth import synth_local_ba
    d = synth_local_ba(n_pose=50, n_pt=20000, obs_per_pose=2000, seed=0)
    poses = torch.from_numpy(d["poses"]).cuda()
    fixed = torch.from_numpy(d["pose_fixed"]).cuda()
    pts = torch.from_numpy(d["points"]).cuda()
    edges = torch.from_numpy(d["edges"].view(np.uint8)).cuda()
    lin = ba.local_ba_linearizer(d["cam"], d["huber_delta"])
    for _ in range(3):
        lin.linearize(poses, fixed, pts, edges)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    wall = []
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(iters):
        t = time.perf_counter()
        lin.linearize(poses, fixed, pts, edges)
        ev[i + 1].record()
        wall.append((time.perf_counter() - t) * 1e3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    dev = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
    print("[%s] total %.3f ms over %d calls = %.4f ms/call" % (tag, dt, iters, dt / iters))
    print("   host ms per call :", " ".join("%.3f" % w for w in wall))
    print("   device ms per call:", " ".join("%.3f" % w for w in dev))
    # a second region right behind it
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        lin.linearize(poses, fixed, pts, edges)
    torch.cuda.synchronize()
    print("   second region: %.4f ms/call" % ((time.perf_counter() - t0) * 1e3 / iters))


if __name__ == "__main__":
    probe("fresh process")
    import bench
    sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--live-pmc", "0", "--no-ba"]
    try:
        bench.main()
    except SystemExit as e:
        print("bench.main exit:", e)
    probe("after bench.main (--no-ba)")
