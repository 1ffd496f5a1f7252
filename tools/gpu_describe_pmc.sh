#!/bin/bash
# describe: counters with and without the XCD mapping (p1 issue mix incl. LDS conflicts, p2 waits, p3 fetch)
cd /root/repo
export TMPDIR=/tmp
for v in 1 0; do
  out=$PWD/gpurun_out/describe_pmc_x$v
  mkdir -p $out
  cmd="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ba --overlap 0 --batch 128 --fast-split 0"
  ( cd /tmp
    OVS_DESCRIBE_XCD=$v timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p3 -o p3 -- $cmd > $out/p3.log 2>&1
    OVS_DESCRIBE_XCD=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $out/p1 -o p1 -- $cmd > $out/p1.log 2>&1
    OVS_DESCRIBE_XCD=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $out/p2 -o p2 -- $cmd > $out/p2.log 2>&1
  )
  python tools/pmc_summary.py $out > $out/summary.txt 2>&1
  find $out -name '*.csv' -size +8M -delete
  echo "=== OVS_DESCRIBE_XCD=$v"; grep -A24 "k_describe" $out/summary.txt | head -30
done
