#!/bin/bash
# round 4, call K2: kernel trace of ovs_local_ba_optimize with the device solver
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04k_prof -o lba -- python /root/repo/tools/time_lba.py device 3 > /dev/null 2>&1
cd /root/repo; f=$(find gpurun_out/r04k_prof -name '*kernel_stats.csv' | head -1); head -16 "$f" | cut -c1-200; cp "$f" gpurun_out/r04k_lba_kernel_stats.csv
find gpurun_out/r04k_prof -name '*.csv' -size +4M -delete; find gpurun_out/r04k_prof -name '*.db' -delete
