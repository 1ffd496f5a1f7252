#!/bin/bash
# Kernel durations and PMC passes for the local-BA kernels: the linearisation alone at config 5 and at ten times the map (tools/lba_lin_sizes.py),
# and a whole ovs_local_ba_optimize (tools/time_lba.py). Separate passes, --kernel-trace only. Usage (GPU box, repo root): tools/gpu_lba_pmc.sh <tag>
tag=$1
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
root=$PWD
run() {   # name, counters..., then -- command
  name=$1; shift
  ctr=""
  while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done
  shift
  if [ -n "$ctr" ]; then pmc="--pmc $ctr"; else pmc="--stats"; fi
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace $pmc --output-format csv -d $out/$name -o p -- "$@" > $out/$name.log 2>&1 )
}
for w in lin opt; do
  if [ $w = lin ]; then cmd="python $root/tools/lba_lin_sizes.py"; else cmd="python $root/tools/time_lba.py device 3"; fi
  run ${w}_trace -- $cmd
  run ${w}_p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -- $cmd
  run ${w}_p2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE -- $cmd
  run ${w}_p3 FETCH_SIZE -- $cmd
  run ${w}_p4 WRITE_SIZE -- $cmd
done
python - $out <<'PY' | tee $out/summary.txt
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for w in ("lin", "opt"):
    print("== %s: %s" % (w, "tools/lba_lin_sizes.py (3 + 20 linearisations at config 5, then 3 + 10 at 200 keyframes x 5000 observations)" if w == "lin" else "tools/time_lba.py device 3"))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(root, w + "_trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ovs::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, w + "_p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ovs::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        if not k.startswith("k_"):
            continue
        d = sorted(dur[k])
        print("%-28s calls %5d  avg %8.1f us  median %8.1f  max %8.1f" % (k[:28], len(d), sum(d) / len(d), d[len(d) // 2], d[-1]))
        if w == "lin" and k.startswith("k_lin"):
            # the two problem sizes separately: the large problem's launches are the slow ones
            big = [x for x in d if x > 3 * d[0]]
            small = [x for x in d if x <= 3 * d[0]]
            if big and small:
                print("%-28s   config 5: %d launches avg %.1f us; large map: %d launches avg %.1f us" % ("", len(small), sum(small) / len(small), len(big), sum(big) / len(big)))
        for c in sorted(acc.get(k, {})):
            v = acc[k][c]
            print("      %-24s n=%-4d mean=%.5g  max=%.5g" % (c, len(v), sum(v) / len(v), max(v)))
    if w == "lin":
        # traffic of ONE linearisation per problem size: the launches of the large map are the upper half of each kernel's sorted counter values.
        # FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: it reports half the bytes of a coalesced read stream;
        # for the gathers of these kernels the factor is an upper bound).
        import json
        out = {}
        for size, pick in (("config5", lambda v: v[: len(v) // 2]), ("large", lambda v: v[-(len(v) // 3):])):
            tot = 0.0
            det = {}
            for k in acc:
                if not k.startswith(("k_lin", "k_reduce_scalars")):
                    continue
                f = sorted(acc[k].get("FETCH_SIZE", [0.0]))
                wv = sorted(acc[k].get("WRITE_SIZE", [0.0]))
                fb = pick(f) or f
                wb = pick(wv) or wv
                b = 2 * 1024 * sum(fb) / len(fb) + 1024 * sum(wb) / len(wb)
                det[k] = round(b)
                tot += b
            out[size] = {"traffic_bytes_per_linearisation": round(tot), "per_kernel": det}
        json.dump(out, open(os.path.join(root, "pmc_lba.json"), "w"), indent=1)
        print(json.dumps(out))
PY
find $out -name '*.csv' -size +2M -delete; find $out -name '*.db' -delete
