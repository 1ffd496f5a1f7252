"""The native multi-device local-BA linearisation (ovs_ba_multi_*) on N devices of one node: both exchange variants -- the packed RCCL
all-reduce and the direct xGMI peer exchange -- timed and checked against the one-device result ("measure both").
ctypes only (no torch import), so bench.py can run it as a subprocess of rank 0 with a time limit; prints ONE JSON line.
Usage: python tools/ba_multi_bench.py [n_gpus] [n_pose n_pt obs_per_pose] [iters]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from openvslam_amd import _lib
from openvslam_amd.ba import BaCam, EDGE_DTYPE
from openvslam_amd.synth import synth_local_ba


def run(n_gpus, n_pose, n_pt, obs, iters):
    L = _lib.lib()
    d = synth_local_ba(n_pose=n_pose, n_pt=n_pt, obs_per_pose=obs, seed=0)
    mono = np.ascontiguousarray(d["edges"], EDGE_DTYPE)
    fixed = np.ascontiguousarray(d["pose_fixed"], np.uint8)
    cam = BaCam(*d["cam"])
    P, X = np.ascontiguousarray(d["poses"]), np.ascontiguousarray(d["points"])
    huber = float(d["huber_delta"])

    def linearize(h):
        out = dict(Hpp=np.zeros((n_pose, 6, 6)), bp=np.zeros((n_pose, 6)), Hll=np.zeros((n_pt, 3, 3)), bl=np.zeros((n_pt, 3)),
                   Hpl=np.zeros((len(mono), 6, 3)), chi2=np.zeros(2))
        _lib.check(L.ovs_ba_multi_linearize(h, P.ctypes.data, X.ctypes.data, huber, 0.0, out["Hpp"].ctypes.data, out["bp"].ctypes.data,
                                            out["Hll"].ctypes.data, out["bl"].ctypes.data, out["Hpl"].ctypes.data, out["chi2"].ctypes.data),
                   "ovs_ba_multi_linearize")
        return out

    def make(n):
        h = C.c_void_p()
        _lib.check(L.ovs_ba_multi_create(n, n_pose, fixed.ctypes.data, n_pt, mono.ctypes.data, len(mono), None, 0, C.byref(cam), 0.0, C.byref(h)),
                   "ovs_ba_multi_create")
        return h

    res = {"n_gpus": n_gpus, "workload": "%d keyframes x %d observations, %d landmarks" % (n_pose, obs, n_pt),
           "allreduce_bytes": (12 * n_pt + 2) * 8, "note": "host entry: every call uploads the state and downloads all blocks (PCIe-inclusive)"}
    h1 = make(1)
    try:
        ref = linearize(h1)
        t = time.perf_counter()
        for _ in range(iters):
            linearize(h1)
        res["one_device_ms"] = round((time.perf_counter() - t) / iters * 1e3, 4)
    finally:
        L.ovs_ba_multi_destroy(h1)
    if n_gpus > 1:
        h = make(n_gpus)
        try:
            for name, mode in (("rccl", 0), ("peer", 1)):
                st = L.ovs_ba_multi_set_exchange(h, mode)
                if st != 0:
                    res[name + "_error"] = _lib.STATUS_NAMES.get(st, st)
                    continue
                out = linearize(h)
                ok_exact = all(np.array_equal(out[k], ref[k]) for k in ("Hpl", "Hpp", "bp"))
                rel = max(float(np.abs(out[k] - ref[k]).max() / np.abs(ref[k]).max()) for k in ("Hll", "bl", "chi2"))
                t = time.perf_counter()
                for _ in range(iters):
                    out2 = linearize(h)
                res[name + "_ms"] = round((time.perf_counter() - t) / iters * 1e3, 4)
                res[name + "_pose_blocks_and_Hpl_bit_equal_to_one_device"] = bool(ok_exact)
                res[name + "_landmark_sums_max_rel_diff"] = rel
                res[name + "_reproducible"] = bool(all(np.array_equal(out[k], out2[k]) for k in out))
        finally:
            L.ovs_ba_multi_destroy(h)
    return res


if __name__ == "__main__":
    a = sys.argv[1:]
    n_gpus = int(a[0]) if a else _lib.lib().ovs_device_count()
    n_pose, n_pt, obs = (int(a[1]), int(a[2]), int(a[3])) if len(a) >= 4 else (50, 20000, 2000)
    iters = int(a[4]) if len(a) >= 5 else 10
    try:
        print(json.dumps(run(n_gpus, n_pose, n_pt, obs, iters)))
    except Exception as ex:   # a measurement aid: report, never hang the caller
        print(json.dumps({"n_gpus": n_gpus, "error": repr(ex)}))
