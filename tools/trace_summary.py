#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace csv (count / avg / min / max microseconds, share of kernel time); kernels
launched with several grid sizes (the 7 pyramid levels) are also broken out per grid. Usage: trace_summary.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(list)
per_grid = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc[name].append(dur)
    per_grid[(name, r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))].append(dur)
tot = sum(sum(v) for v in acc.values()) or 1.0
print("# source: rocprofv3 --kernel-trace --stats (csv); durations in microseconds")
print("%-60s %7s %10s %10s %10s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-60s %7d %10.2f %10.2f %10.2f %6.2f%%" % (name[:60], len(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
print("# per grid size (threads) for multi-shape kernels")
for (name, gx, gy, gz), v in sorted(per_grid.items(), key=lambda kv: (kv[0][0], -sum(kv[1]))):
    if len({k for k in per_grid if k[0] == name}) > 1 and "ovs::" in name:   # (templated kernels are reported as "void ovs::k_...<...>")
        print("%-40s grid %8s x %6s x %4s %7d %10.2f" % (name[:40], gx, gy, gz, len(v), sum(v) / len(v)))
