#!/bin/bash
# round 3, call b: FAST v4 phase costs + PMC of v4 / v3
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
R=$PWD
timeout 600 python tools/fast_phases.py 64 5 > gpurun_out/r3b/phases.txt 2>&1
cat gpurun_out/r3b/phases.txt
cd /tmp
for v in v4 v3; do
  if [ $v = v3 ]; then export OVS_FAST_V3=1; else unset OVS_FAST_V3; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/r3b/$v/p1 -o p1 -- python $R/tools/fast_phases.py 64 2 0 > $R/gpurun_out/r3b/$v.p1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3b/$v/p2 -o p2 -- python $R/tools/fast_phases.py 64 2 0 > $R/gpurun_out/r3b/$v.p2.log 2>&1
done
cd $R
for v in v4 v3; do echo "== $v"; python tools/pmc_summary.py gpurun_out/r3b/$v 2>&1 | grep -A 22 "k_fast_cells" | head -50; done
find gpurun_out/r3b -name '*.csv' -size +4M -delete
