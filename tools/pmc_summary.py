#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output (counter_collection.csv) per kernel: mean counter value per dispatch.
Usage: pmc_summary.py <dir containing p1..p4 subdirs>"""
import csv
import glob
import os
import sys
from collections import defaultdict

PMC_STAGE_SOURCES = {"fast": ("orb_fast.hip",), "pyramid": ("orb_pyramid.hip",), "tree": ("orb_tree.hip",),
                     "describe": ("orb_describe.hip", "orb_pattern.inc"), "match_near": ("match_hamming.hip",),
                     "match_resolve": ("match_hamming.hip",)}


def pmc_stage_fingerprint(src_dir, stage, read=None):
    """sha256[:16] of the sources a stage's kernels are compiled from: its .hip file(s) and the shared headers of csrc/."""
    import hashlib
    read = read or (lambda fn: open(os.path.join(src_dir, fn), "rb").read())
    hsh = hashlib.sha256()
    for fn in sorted(set(PMC_STAGE_SOURCES[stage]) | {f for f in os.listdir(src_dir) if f.endswith(".h")}):
        hsh.update(fn.encode() + b"\0" + read(fn))
    return hsh.hexdigest()[:16]



root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [values per dispatch]
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("# mean per dispatch; source: rocprofv3 --pmc (separate passes), %s" % root)
for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_WAVE_CYCLES", [0]))):
    if "ovs" not in k and not k.startswith("k_") and "void k_" not in k:
        continue
    print(k)
    for c in names:
        v = acc[k].get(c)
        if v:
            print("    %-26s n=%-4d mean=%.6g" % (c, len(v), sum(v) / len(v)))

# ---- per-launch HBM-side traffic for bench.py's roofline.traffic (profiles/pmc_traffic.json when --json is given).
# Units / corrections (MI355X_MICROARCH.md HBM section, re-calibrated on this box with tools/ubench/hbm_calib.hip -- see
# profiles/r01_hbm_calib.txt): FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports exactly HALF the bytes of a coalesced
# read stream (4-byte and 16-byte per lane alike), WRITE_SIZE is exact. Both count requests on the L2's fabric side, i.e. lines
# re-fetched by another XCD's L2 and Infinity-Cache hits are included.
if "--json" in sys.argv:
    import json
    # a stage's kernels (the pyramid of a batch is three k_resize_pair_u8 launches + one k_resize_linear_u8 since round 6: the stage's figure is the
    # SUM over its launches divided by the number of stage calls = launches of k_describe, one per extract call)
    stage_of = {"ovs::k_fast_cells": "fast", "ovs::k_resize_linear_u8": "pyramid", "ovs::k_resize_pair_u8": "pyramid", "ovs::k_tree": "tree",
                "ovs::k_describe": "describe", "ovs::k_hamming_near": "match_near", "ovs::k_bf_resolve": "match_resolve"}
    call_marker = {"pyramid": "ovs::k_describe"}   # stages whose calls are not one launch each: counted by this kernel's launches
    stages = sorted(set(stage_of.values()))
    def kernels_of(st):
        return [n for n in acc if stage_of.get(n.replace("void ", "").split("<")[0].strip().split("(")[0].strip()) == st]   # exact kernel (k_hamming_near, not k_hamming_near_popc)
    def calls_of(st, counter):
        if st in call_marker:
            mk = [n for n in acc if n.replace("void ", "").split("<")[0].strip().split("(")[0].strip() == call_marker[st]]
            return sum(len(acc[n].get(counter, [])) for n in mk) or 1
        return sum(len(acc[n].get(counter, [])) for n in kernels_of(st)) or 1
    out = {}
    for st in stages:
        kk = kernels_of(st)
        if not kk:
            continue
        f = sum(sum(acc[n].get("FETCH_SIZE", [])) for n in kk)
        w = sum(sum(acc[n].get("WRITE_SIZE", [])) for n in kk)
        out[st] = int((2.0 * f / calls_of(st, "FETCH_SIZE") + w / calls_of(st, "WRITE_SIZE")) * 1024.0)
    # VALU / SALU wave-instructions per stage call (SQ pass) for bench.py's roofline_valu
    for cname, key in (("SQ_INSTS_VALU", "insts_valu"), ("SQ_INSTS_SALU", "insts_salu")):
        d = {}
        for st in stages:
            vals = [v for n in kernels_of(st) for v in acc[n].get(cname, [])]
            if vals:
                d[st] = int(sum(vals) / calls_of(st, cname))
        if d:
            out[key] = d
    # fingerprints of the kernel sources the counters were collected from, per stage: bench.py refuses a stage's figure when that stage's
    # sources have changed since
    src_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openvslam_amd", "csrc")
    out["csrc_sha16_by_stage"] = {st: pmc_stage_fingerprint(src_dir, st) for st in PMC_STAGE_SOURCES}
    if "--batch" in sys.argv:
        out["batch"] = int(sys.argv[sys.argv.index("--batch") + 1])   # frames per launch the passes ran at (bench.py scales by it)
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    print("traffic bytes per stage call:", out)
