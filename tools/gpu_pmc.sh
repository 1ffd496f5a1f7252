#!/bin/bash
# PMC passes over a short bench run (separate passes: SQ issue mix, SQ stalls, TCC read bytes, TCC write bytes, MFMA).
# Usage (on the GPU box, from the repo root): tools/gpu_pmc.sh <tag>
tag=${1:-pmc}
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
cmd="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ba --overlap 0 --batch 128 --fast-split 0"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $out/p1 -o p1 -- $cmd > $out/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -o p2 -- $cmd > $out/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p3 -o p3 -- $cmd > $out/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/p4 -o p4 -- $cmd > $out/p4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA --output-format csv -d $out/p5 -o p5 -- $cmd > $out/p5.log 2>&1
cd - > /dev/null
python tools/pmc_summary.py $out --json $out/pmc_traffic.json --batch 128 > $out/summary.txt 2>&1
# keep only the small artefacts
find $out -name '*.csv' -size +8M -delete
ls -la $out $out/p1 2>/dev/null | head -40
