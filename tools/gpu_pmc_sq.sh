#!/bin/bash
# SQ-only PMC passes (issue mix + stalls) over a short bench run. Usage: tools/gpu_pmc_sq.sh <tag>
tag=${1:-pmcsq}
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
cmd="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ba --overlap 0 --batch 128 --fast-split 0"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $out/p1 -o p1 -- $cmd > $out/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -o p2 -- $cmd > $out/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/p3 -o p3 -- $cmd > $out/p3.log 2>&1
cd - > /dev/null
python tools/pmc_summary.py $out > $out/summary.txt 2>&1
find $out -name '*.csv' -size +8M -delete
grep -A18 "k_fast_cells" $out/summary.txt | head -24
