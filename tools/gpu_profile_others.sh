#!/bin/bash
# rocprofv3 kernel trace + PMC passes (separate, kernel-trace only: the gpurun guard) of tools/profile_other_kernels.py.
# Usage (GPU box, repo root): tools/gpu_profile_others.sh <tag>  -> gpurun_out/<tag>/{trace_summary.txt,pmc_summary.txt}
tag=${1:-others}
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
cmd="python $PWD/tools/profile_other_kernels.py"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o t -- $cmd > $out/t.log 2>&1
export OVS_PROFILE_ITERS=2
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $out/p1 -o p1 -- $cmd > $out/p1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -o p2 -- $cmd > $out/p2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p3 -o p3 -- $cmd > $out/p3.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/p4 -o p4 -- $cmd > $out/p4.log 2>&1
cd - > /dev/null
python tools/trace_summary.py $(ls $out/t/*kernel_trace.csv $out/t/*/*kernel_trace.csv 2>/dev/null | head -1) > $out/trace_summary.txt 2>&1
python tools/pmc_summary.py $out > $out/pmc_summary.txt 2>&1
find $out -name '*.csv' -size +8M -delete
find $out -name '*.db' -delete
head -60 $out/trace_summary.txt
