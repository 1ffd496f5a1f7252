#!/bin/bash
# round 4, call L: k_chol_solve v2 (C tiles prefetched under the factorisation, 4 tiles per wave in flight, next panel written to LDS,
# rsq-based pivots, right-looking backward substitution with prefetch): tests + timing + kernel stats
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ba.py -q 2>&1 | tail -8
OVS_BA_TRACE=1 timeout 300 python tools/time_lba.py device 5 > gpurun_out/r04l_lba_device.txt 2>&1; tail -4 gpurun_out/r04l_lba_device.txt
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04l_prof -o lba -- python /root/repo/tools/time_lba.py device 3 > /dev/null 2>&1
cd /root/repo; f=$(find gpurun_out/r04l_prof -name '*kernel_stats.csv' | head -1); head -16 "$f" | cut -c1-120; cp "$f" gpurun_out/r04l_lba_kernel_stats.csv
find gpurun_out/r04l_prof -name '*.csv' -size +4M -delete; find gpurun_out/r04l_prof -name '*.db' -delete
