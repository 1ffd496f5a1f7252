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
            k = row.get("Kernel_Name", "?").split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("# mean per dispatch; source: rocprofv3 --pmc (separate passes), %s" % root)
for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_WAVE_CYCLES", [0]))):
    if not (k.startswith("ovs::") or "ovs" in k):
        continue
    print(k)
    for c in names:
        v = acc[k].get(c)
        if v:
            print("    %-26s n=%-4d mean=%.6g" % (c, len(v), sum(v) / len(v)))
