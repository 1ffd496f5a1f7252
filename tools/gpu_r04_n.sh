#!/bin/bash
cd /root/repo
OVS_BA_TRACE=1 timeout 120 python tools/solve_probe.py 2>&1 | grep -v "^\[k_chol" | tail -4
OVS_BA_TRACE=1 timeout 120 python tools/solve_probe.py 2>&1 | grep "^\[k_chol" | awk 'NR%3==0' 
timeout 900 python -m pytest tests/test_gpu_ba.py -q 2>&1 | tail -6
OVS_BA_TRACE=1 timeout 300 python tools/time_lba.py device 5 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04n_prof -o lba -- python /root/repo/tools/time_lba.py device 3 > /dev/null 2>&1
cd /root/repo; f=$(find gpurun_out/r04n_prof -name '*kernel_stats.csv' | head -1); head -4 "$f" | cut -c1-120; cp "$f" gpurun_out/r04n_lba_kernel_stats.csv
find gpurun_out/r04n_prof -name '*.csv' -size +4M -delete; find gpurun_out/r04n_prof -name '*.db' -delete
