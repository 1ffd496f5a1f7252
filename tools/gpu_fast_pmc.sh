#!/bin/bash
# PMC passes (SQ issue mix, SQ stalls, FETCH_SIZE) over the FAST kernel alone: tools/ab_extract.py's child (64 frames, 3 repetitions) under
# rocprofv3, once per variant. Usage (GPU box, repo root): tools/gpu_fast_pmc.sh <tag> "<VAR=VAL,VAR=VAL>" ...
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
i=0
for v in "$@"; do
  i=$((i+1))
  envs=$(echo "$v" | tr ',' ' ')
  for pass in 1 2 3; do
    case $pass in
      1) ctr="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" ;;
      2) ctr="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" ;;
      3) ctr="FETCH_SIZE" ;;
    esac
    ( cd /tmp && env $envs OVS_AB_CHILD=1 timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/v${i}p${pass} -o p -- python $OLDPWD/tools/ab_extract.py 64 3 > $out/v${i}p${pass}.log 2>&1 )
  done
  echo "== variant $i: $v" | tee -a $out/summary.txt
  python - $out $i <<'PY' | tee -a $out/summary.txt
import csv, glob, os, sys
from collections import defaultdict
root, i = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(os.path.join(root, "v%sp*" % i, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_fast" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c in sorted(acc):
    print("    %-24s n=%-3d mean=%.5g" % (c, len(acc[c]), sum(acc[c]) / len(acc[c])))
PY
done
find $out -name '*.csv' -size +2M -delete; find $out -name '*.db' -delete
