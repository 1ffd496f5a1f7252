"""Stage times of the extractor under environment switches the launchers read per call (round 3 tuning aid).
Usage (GPU box): python tools/fast_phases.py <batch> <reps> VAR=value[,VAR2=value2] [VAR=value ...]
Each argument after reps is one configuration (comma-separated assignments; "-" = no switches)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from openvslam_amd import _lib, feature, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
configs = sys.argv[3:] or ["-"]
frames = torch.from_numpy(synth.synth_video(1080, 1920, B, seed=100)).cuda()
ex = feature.orb_extractor(feature.orb_params(2000, 1.2, 8, 20, 7), max_rows=1080, max_cols=1920, max_batch=B)
ex.set_fast_split(False)
cap = ex.max_keypoints
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
ref = None
for cfg in configs:
    keys = []
    if cfg != "-":
        for a in cfg.split(","):
            k, v = a.split("=")
            os.environ[k] = v
            keys.append(k)
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
    n = max(nc.value, 1)
    out = (kps.cpu().numpy().tobytes(), desc.cpu().numpy().tobytes(), cnt.cpu().numpy().tobytes())
    if ref is None:
        ref = out
    print("%-40s B=%d  pyramid %.4f  fast %.4f  tree %.4f  describe %.4f ms per launch  (keypoints %d, outputs == first config: %s)"
          % (cfg, B, st[0] / n, st[1] / n, st[2] / n, st[3] / n, int(cnt.sum()), out == ref), flush=True)
    for k in keys:
        os.environ.pop(k, None)
