#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / avg / min / max / share, as rocprofv3 --stats would
print. Usage: rocpd_summary.py <results.db> [title]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), max(vgpr_count), "
                        "max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by sum(end-start) desc"))
tot = sum(r[5] for r in rows) or 1
print("# %s" % (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]))
print("# source: rocprofv3 --kernel-trace --stats (rocpd sqlite `kernels` view); durations in microseconds")
print("%-64s %7s %11s %11s %11s %7s %5s %5s %7s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "scratch"))
for r in rows:
    name = r[0].split("(")[0]
    print("%-64s %7d %11.2f %11.2f %11.2f %6.2f%% %5s %5s %7s %7s" % (name[:64], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / tot,
                                                                    r[6], r[7], r[8], r[9]))
