"""A/B of kernel variants: the same device-resident batch through ovs_orb_extract_batch_dev under different tuning switches
(OVS_FAST_CELLS, OVS_FAST_TIMING, ... -- the library reads them ONCE per process, so every variant runs in its own child process),
per-stage HIP-event times with every kernel alone on the GPU, and a hash comparison of all outputs between the variants.
Usage (GPU box): python tools/ab_extract.py [batch] [reps] [VAR=VAL,VAR=VAL ...]   (each further argument is one variant's environment)"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROWS, COLS = 1080, 1920


def child(B, REPS):
    import numpy as np
    import torch

    from openvslam_amd import _lib, feature, synth

    frames = torch.from_numpy(synth.synth_video(ROWS, COLS, B, seed=100)).cuda()
    ex = feature.orb_extractor(feature.orb_params(2000, 1.2, 8, 20, 7), max_rows=ROWS, max_cols=COLS, max_batch=B)
    ex.set_fast_split(False)
    cap = ex.max_keypoints
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
    for _ in range(2):
        ex.extract_batch_dev(frames, kps, desc, cnt, stream=s)
    torch.cuda.synchronize()
    _lib.check(L.ovs_orb_profile_enable(ex._h, 1), "profile_enable")
    st = (C.c_float * 4)()
    nc = C.c_int32()
    _lib.check(L.ovs_orb_profile_read(ex._h, st, C.byref(nc)), "profile_read")
    for _ in range(REPS):
        ex.extract_batch_dev(frames, kps, desc, cnt, stream=s)
        torch.cuda.synchronize()
    _lib.check(L.ovs_orb_profile_read(ex._h, st, C.byref(nc)), "profile_read")
    _lib.check(L.ovs_orb_profile_enable(ex._h, 0), "profile_enable")
    k = max(nc.value, 1)
    h = hashlib.sha256()
    h.update(cnt.cpu().numpy().tobytes())
    h.update(kps.cpu().numpy().tobytes())
    h.update(desc.cpu().numpy().tobytes())
    for l in (1, 4, 7):
        h.update(np.ascontiguousarray(ex.image_pyramid(l, frame=B - 1)).tobytes())
    for l in (0, 3, 7):
        h.update(np.sort(np.stack(ex.debug_candidates(l, frame=B - 1)), axis=1).tobytes())
    print(json.dumps({"ms": [v / k for v in st], "keypoints": int(cnt.sum()), "sha": h.hexdigest()}), flush=True)


if __name__ == "__main__":
    if os.environ.get("OVS_AB_CHILD"):
        child(int(sys.argv[1]), int(sys.argv[2]))
        sys.exit(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    variants = sys.argv[3:] or ["", "OVS_FAST_CELLS=4"]
    ref = None
    for v in variants:
        env = dict(os.environ, OVS_AB_CHILD="1")
        for kv in filter(None, v.split(",")):
            key, val = kv.split("=", 1)
            env[key] = val
        out = subprocess.run([sys.executable, os.path.abspath(__file__), str(B), str(REPS)], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            print("%-40s FAILED rc=%d\n%s" % (v or "(default)", out.returncode, out.stderr[-2000:]), flush=True)
            continue
        r = json.loads(line[-1])
        if ref is None:
            ref = r["sha"]
        ms = r["ms"]
        print("%-40s B=%d  pyramid %.4f  fast %.4f  tree %.4f  describe %.4f ms per launch (sum %.4f), %d keypoints, outputs == first variant: %s"
              % (v or "(default)", B, ms[0], ms[1], ms[2], ms[3], sum(ms), r["keypoints"], r["sha"] == ref), flush=True)
        timing = [l for l in out.stderr.splitlines() if "timing]" in l]
        if timing:
            print("    " + timing[-1], flush=True)
