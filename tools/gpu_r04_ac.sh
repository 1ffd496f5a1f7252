#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ba.py -q 2>&1 | grep -E "passed|failed|^E  " | head -5
timeout 300 python tools/time_lba.py device 6 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04ac_prof -o lba -- python /root/repo/tools/time_lba.py device 3 > /dev/null 2>&1
cd /root/repo; f=$(find gpurun_out/r04ac_prof -name '*kernel_stats.csv' | head -1); python - "$f" <<'PY' > gpurun_out/r04ac_lba_kernel_stats.txt
import csv,sys
print("# rocprofv3 --kernel-trace --stats -- python tools/time_lba.py device 3   (3 calls of ovs_local_ba_optimize at BASELINE config 5: 15 LM trials each)")
for r in csv.DictReader(open(sys.argv[1])):
    print("%-30s calls %4s avg %9.1f us  %7s %%" % (r["Name"].split("(")[0][:30], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
head -12 gpurun_out/r04ac_lba_kernel_stats.txt
find gpurun_out/r04ac_prof -name '*.csv' -size +1M -delete; find gpurun_out/r04ac_prof -name '*.db' -delete
