#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ba.py -q 2>&1 | grep -E "passed|failed"
OVS_BA_TRACE=1 timeout 300 python tools/time_lba.py device 6 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04q_prof -o lba -- python /root/repo/tools/time_lba.py device 3 > /dev/null 2>&1
cd /root/repo; f=$(find gpurun_out/r04q_prof -name '*kernel_stats.csv' | head -1); python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-28s calls %4s avg %9.1f us  %5s %%" % (r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
cp "$f" gpurun_out/r04q_lba_kernel_stats.csv
find gpurun_out/r04q_prof -name '*.csv' -size +4M -delete; find gpurun_out/r04q_prof -name '*.db' -delete
