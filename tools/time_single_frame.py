"""Latency of ONE device-resident frame through ovs_orb_extract_batch_dev (B = 1): wall time per call and, under rocprofv3 --kernel-trace,
the kernels it consists of. Usage (GPU box): python tools/time_single_frame.py [rows cols nfeat]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from openvslam_amd import feature, synth

rows, cols, nfeat = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1080, 1920, 2000)
img = torch.from_numpy(synth.synth_frame(rows, cols, seed=3)[None]).cuda()
ex = feature.orb_extractor(feature.orb_params(nfeat), max_rows=rows, max_cols=cols, max_batch=1)
cap = ex.max_keypoints
kps = torch.zeros((1, cap, 7), dtype=torch.float32, device="cuda")
desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros((1,), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    ex.extract_batch_dev(img, kps, desc, cnt, stream=s)
torch.cuda.synchronize()
N = 200
t = time.perf_counter()
for _ in range(N):
    ex.extract_batch_dev(img, kps, desc, cnt, stream=s)
    torch.cuda.synchronize()          # per-frame latency: the tracker needs the keypoints before it can go on
print("%dx%d, %d features: %.3f ms per frame (device-resident, synchronised per frame), %d keypoints" % (cols, rows, nfeat, (time.perf_counter() - t) / N * 1e3, int(cnt.item())))

# per-stage HIP-event times of the single-frame chain (pyramid | FAST | tree | describe), level-0 split off so that the stages are serial
import ctypes as C
from openvslam_amd import _lib
L = _lib.lib()
for split in (0, 1):
    ex.set_fast_split(bool(split))
    _lib.check(L.ovs_orb_profile_enable(ex._h, 1), "profile_enable")
    st = (C.c_float * 4)()
    nc = C.c_int32()
    for _ in range(3):
        ex.extract_batch_dev(img, kps, desc, cnt, stream=s)
        torch.cuda.synchronize()
    _lib.check(L.ovs_orb_profile_read(ex._h, st, C.byref(nc)), "profile_read")
    for _ in range(50):
        ex.extract_batch_dev(img, kps, desc, cnt, stream=s)
        torch.cuda.synchronize()
    _lib.check(L.ovs_orb_profile_read(ex._h, st, C.byref(nc)), "profile_read")
    k = max(nc.value, 1)
    print("fast_split %d: pyramid %.1f us, FAST %.1f us, tree %.1f us, describe %.1f us (main-stream stage boundaries)" % (split, *(1e3 * v / k for v in st)))
    _lib.check(L.ovs_orb_profile_enable(ex._h, 0), "profile_enable")
