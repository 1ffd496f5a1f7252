#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_ba_multi.py -q 2>&1 | tail -6
OVS_BA_TRACE=1 timeout 300 python tools/time_lba.py device 5 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04o_prof -o lba -- python /root/repo/tools/time_lba.py device 3 > /dev/null 2>&1
cd /root/repo; f=$(find gpurun_out/r04o_prof -name '*kernel_stats.csv' | head -1); head -14 "$f" | cut -c1-60,180-260; cp "$f" gpurun_out/r04o_lba_kernel_stats.csv
find gpurun_out/r04o_prof -name '*.csv' -size +4M -delete; find gpurun_out/r04o_prof -name '*.db' -delete
