#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration passes (separate --pmc passes, as the guide prescribes)
export TMPDIR=/tmp
out=$PWD/gpurun_out/calib
mkdir -p $out
bin=$PWD/tools/ubench/hbm_calib
cd /tmp
$bin > $out/bytes.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -o f -- $bin > $out/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/w -o w -- $bin > $out/w.log 2>&1
cd - > /dev/null
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(open("$out/bytes.txt").read())
for k in sorted(acc):
    if "calib" in k:
        print(k, {c: sum(v) / len(v) for c, v in acc[k].items()})
PY
