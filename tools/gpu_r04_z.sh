#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
OVS_BA_TRACE=1 timeout 120 python tools/solve_probe.py 2>&1 | grep -v "^\[k_chol" | tail -3
OVS_BA_TRACE=1 timeout 120 python tools/solve_probe.py 2>&1 | grep "^\[k_chol" | awk 'NR%3==0' > gpurun_out/r04z_solve_phases.txt; cat gpurun_out/r04z_solve_phases.txt
timeout 900 python -m pytest tests/test_gpu_ba.py -q 2>&1 | grep -E "passed|failed|^E  " | head -5
timeout 300 python tools/time_lba.py device 6 2>&1 | tail -2
