"""Evidence for the two-extractor (stereo left / right std::thread) case of openvslam_amd/cpp/bench_shim: from a rocprofv3 --kernel-trace csv of
the whole program, find the kernels that ran while a kernel of ANOTHER queue was running, and compare them with the same kernels running alone.
Usage: python tools/two_thread_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ovs::", "")
    if not name.startswith("k_"):
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name))
rows.sort()
# The program runs its phases one after the other (single extractor loops, then two extractors on two threads, then the tracked frame); a phase
# shows in the RATE of k_describe launches (one per extract() call). Blocks of 50 consecutive calls: wall time, calls per second, how long the GPU
# was busy (union of all kernels in the block's span) and how many kernels ran at once while it was.
desc = [i for i, r in enumerate(rows) if r[3].startswith("k_describe")]
print("# blocks of 50 extract() calls (k_describe launches) in program order")
print("%5s %10s %12s %10s %12s %14s" % ("block", "wall ms", "ms per call", "busy %", "concurrency", "kernel ms/call"))
for b0 in range(0, len(desc) - 49, 50):
    i0, i1 = desc[b0], desc[b0 + 49]
    t0, t1 = rows[i0][0], rows[i1][1]
    span = sorted((s, e) for (s, e, q, n) in rows if s >= t0 and e <= t1)
    union = 0
    cs, ce = span[0]
    for s, e in span[1:]:
        if s > ce:
            union += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    union += ce - cs
    ksum = sum(e - s for s, e in span)
    print("%5d %10.2f %12.4f %10.0f %12.2f %14.4f" % (b0 // 50, (t1 - t0) / 1e6, (t1 - t0) / 1e6 / 50, 100.0 * union / (t1 - t0), ksum / union, ksum / 1e6 / 50))
# per kernel: median duration in the blocks whose call rate is the two-thread one (two calls in flight) against the single-extractor blocks
per = {}
for b0 in range(0, len(desc) - 49, 50):
    i0, i1 = desc[b0], desc[b0 + 49]
    t0, t1 = rows[i0][0], rows[i1][1]
    per_call = (t1 - t0) / 1e6 / 50
    per[b0] = per_call
fast = sorted(per.values())[: max(1, len(per) // 6)]   # the fastest sixth of the blocks: two threads
thr = fast[-1] * 1.05
two = defaultdict(list)
one = defaultdict(list)
for b0, pc in per.items():
    i0, i1 = desc[b0], desc[b0 + 49]
    t0, t1 = rows[i0][0], rows[i1][1]
    for (s, e, q, n) in rows:
        if s >= t0 and e <= t1:
            (two if pc <= thr else one)[n].append((e - s) / 1e3)
print("# median kernel duration (us): single-extractor blocks | two-thread blocks (ms per call <= %.4f)" % thr)
for n in sorted(two, key=lambda n: -sum(two[n])):
    a, b = sorted(one.get(n, [0.0])), sorted(two[n])
    print("%-36s %10.1f | %10.1f   x%.2f" % (n[:36], a[len(a) // 2], b[len(b) // 2], b[len(b) // 2] / max(a[len(a) // 2], 1e-9)))
