#!/usr/bin/env python3
"""A/B of the one-launch pyramid (k_pyramid_chain) against the level-by-level kernels: HIP-event time of the pyramid stage alone (level-0 split
off, so nothing runs beside it) at 1, 2, 8, 32 frames per launch, 1920x1080, 8 levels. The switch is
ovs_orb_set_pyramid_chain (frames per launch up to which the chain runs). usage: python tools/pyr_chain_ab.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvslam_amd import _lib, feature            # noqa: E402
from openvslam_amd.synth import synth_video       # noqa: E402


def main():
    rows, cols = 1080, 1920
    L = _lib.lib()
    for B in (1, 2, 8, 32):
        frames = torch.from_numpy(synth_video(rows, cols, max(8, B), seed=5)[:B]).cuda()
        ex = feature.orb_extractor(feature.orb_params(2000, 1.2, 8, 20, 7), max_rows=rows, max_cols=cols, max_batch=B)
        ex.set_fast_split(False)
        cap = ex.max_keypoints
        kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
        s = torch.cuda.Stream()
        res = {}
        for chain in (0, 1):
            ex.set_pyramid_chain(64 if chain else 0)
            for _ in range(5):
                ex.extract_batch_dev(frames, kps, desc, cnt, stream=s.cuda_stream)
            torch.cuda.synchronize()
            _lib.check(L.ovs_orb_profile_enable(ex._h, 1), "profile_enable")
            for _ in range(20):
                ex.extract_batch_dev(frames, kps, desc, cnt, stream=s.cuda_stream)
                torch.cuda.synchronize()
            st4 = (C.c_float * 4)()
            nc = C.c_int32()
            _lib.check(L.ovs_orb_profile_read(ex._h, st4, C.byref(nc)), "profile_read")
            _lib.check(L.ovs_orb_profile_enable(ex._h, 0), "profile_enable")
            res[chain] = [v / max(nc.value, 1) * 1e3 for v in st4]
        print("B=%3d  pyramid stage: level-by-level %8.1f us   one launch %8.1f us   (fast %0.1f / %0.1f, tree %0.1f / %0.1f, describe %0.1f / %0.1f us)" % (
            B, res[0][0], res[1][0], res[0][1], res[1][1], res[0][2], res[1][2], res[0][3], res[1][3]))


if __name__ == "__main__":
    main()
