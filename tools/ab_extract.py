"""A/B of kernel variants inside one process (round 3): the same device-resident batch through ovs_orb_extract_batch_dev with the
environment switches the launchers read per call (OVS_FAST_V3, OVS_RESIZE_V3, ...), per-stage HIP-event times with every kernel alone
on the GPU, and a byte comparison of all outputs between the variants.
Usage (GPU box): python tools/ab_extract.py [batch] [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from openvslam_amd import _lib, feature, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ROWS, COLS = 1080, 1920
frames = torch.from_numpy(synth.synth_video(ROWS, COLS, B, seed=100)).cuda()
ex = feature.orb_extractor(feature.orb_params(2000, 1.2, 8, 20, 7), max_rows=ROWS, max_cols=COLS, max_batch=B)
ex.set_fast_split(False)
cap = ex.max_keypoints
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream


def run(env):
    for k in ("OVS_FAST_V3", "OVS_RESIZE_V3", "OVS_DESCRIBE_V3"):
        os.environ.pop(k, None)
    os.environ.update(env)
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
    pyr = [ex.image_pyramid(l, frame=B - 1) for l in (1, 4, 7)]
    cands = [np.sort(np.stack(ex.debug_candidates(l, frame=B - 1)), axis=1) for l in (0, 3, 7)]
    return [v / k for v in st], kps.cpu().numpy().view(np.uint8), desc.cpu().numpy(), cnt.cpu().numpy(), pyr, cands


variants = [("v3 all", {"OVS_FAST_V3": "1", "OVS_RESIZE_V3": "1", "OVS_DESCRIBE_V3": "1"}), ("fast v4", {"OVS_RESIZE_V3": "1", "OVS_DESCRIBE_V3": "1"}),
            ("resize v4", {"OVS_FAST_V3": "1", "OVS_DESCRIBE_V3": "1"}), ("describe v4", {"OVS_FAST_V3": "1", "OVS_RESIZE_V3": "1"}), ("v4 all", {})]
ref = None
for name, env in variants:
    ms, kps, desc, cnt, pyr, cands = run(env)
    same = ""
    if ref is None:
        ref = (kps, desc, cnt, pyr, cands)
    else:
        ok_out = np.array_equal(cnt, ref[2]) and np.array_equal(kps, ref[0]) and np.array_equal(desc, ref[1])
        ok_pyr = all(np.array_equal(a, b) for a, b in zip(pyr, ref[3]))
        ok_cand = all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(cands, ref[4]))
        same = "  outputs == v3: %s, pyramid planes: %s, candidates: %s" % (ok_out, ok_pyr, ok_cand)
    print("%-12s B=%d  pyramid %.4f  fast %.4f  tree %.4f  describe %.4f ms per launch (sum %.4f), %d keypoints%s"
          % (name, B, ms[0], ms[1], ms[2], ms[3], sum(ms), int(cnt.sum()), same), flush=True)
