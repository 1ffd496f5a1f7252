#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output (counter_collection.csv) per kernel: mean counter value per dispatch.
Usage: pmc_summary.py <dir containing p1..p4 subdirs>"""
import csv
import glob
import os
import sys
from collections import defaultdict

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
    stage_of = {"ovs::k_fast_cells": "fast", "ovs::k_resize_linear_u8": "pyramid", "ovs::k_tree": "tree", "ovs::k_describe": "describe",
                "ovs::k_hamming_near": "match_near", "ovs::k_bf_resolve": "match_resolve"}
    launches_per_call = {"pyramid": 7}
    out = {}
    for k, st in stage_of.items():
        kk = [n for n in acc if n.replace("void ", "").split("<")[0].strip() == k]   # exact kernel (k_hamming_near, not k_hamming_near_popc)
        if not kk:
            continue
        f = sum(sum(acc[n].get("FETCH_SIZE", [])) for n in kk)
        w = sum(sum(acc[n].get("WRITE_SIZE", [])) for n in kk)
        nf = sum(len(acc[n].get("FETCH_SIZE", [])) for n in kk) or 1
        nw = sum(len(acc[n].get("WRITE_SIZE", [])) for n in kk) or 1
        per_launch = (2.0 * f / nf + w / nw) * 1024.0
        out[st] = int(per_launch * launches_per_call.get(st, 1))
    # VALU / SALU wave-instructions per stage call (SQ pass) for bench.py's roofline_valu
    for cname, key in (("SQ_INSTS_VALU", "insts_valu"), ("SQ_INSTS_SALU", "insts_salu")):
        d = {}
        for k, st in stage_of.items():
            kk = [n for n in acc if n.replace("void ", "").split("<")[0].strip() == k]
            vals = [v for n in kk for v in acc[n].get(cname, [])]
            if vals:
                d[st] = int(sum(vals) / len(vals) * launches_per_call.get(st, 1))
        if d:
            out[key] = d
    # fingerprint of the kernel sources the counters were collected from: bench.py refuses a traffic figure whose kernels have changed since
    import hashlib
    src_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openvslam_amd", "csrc")
    hsh = hashlib.sha256()
    for fn in sorted(os.listdir(src_dir)):
        if fn.endswith((".h", ".inc")) or fn.startswith(("orb_", "match_hamming")):   # the kernels the traffic file covers (extraction, brute-force matcher) and the shared headers
            hsh.update(open(os.path.join(src_dir, fn), "rb").read())
    out["csrc_sha16"] = hsh.hexdigest()[:16]
    if "--batch" in sys.argv:
        out["batch"] = int(sys.argv[sys.argv.index("--batch") + 1])   # frames per launch the passes ran at (bench.py scales by it)
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    print("traffic bytes per stage call:", out)
