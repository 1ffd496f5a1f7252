#!/bin/bash
# round 4, call K: the reduced camera system on the device (k_chol_solve / k_pose_update): unit tests, LBA parity, timing of both solvers
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ba.py -q -x -k "dense_solve" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_ba.py -q 2>&1 | tail -8
OVS_BA_TRACE=1 timeout 300 python tools/time_lba.py device 5 > gpurun_out/r04k_lba_device.txt 2>&1; tail -8 gpurun_out/r04k_lba_device.txt
OVS_BA_TRACE=1 timeout 300 python tools/time_lba.py host 4 > gpurun_out/r04k_lba_host.txt 2>&1; tail -4 gpurun_out/r04k_lba_host.txt
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r04k_prof -o lba -- python /root/repo/tools/time_lba.py device 3 > /dev/null 2>&1
cd /root/repo; f=$(ls gpurun_out/r04k_prof/*/*kernel_stats.csv gpurun_out/r04k_prof/*kernel_stats.csv 2>/dev/null | head -1); head -14 "$f" | cut -c1-160
