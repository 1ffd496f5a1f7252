#!/bin/bash
# round 4, call AI: the 900-case campaign again (device solver; single-view landmarks get the looser point bound)
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python tools/fuzz_parity.py --cases 900 --seed 4601 --out gpurun_out/r04ah_fuzz_a.txt > /dev/null 2>&1; echo "rc=$?"; tail -1 gpurun_out/r04ah_fuzz_a.txt | cut -c1-200
grep "local_ba" gpurun_out/r04ah_fuzz_a.txt | awk '{print $NF}' | sort | uniq -c
grep -c "single-view" gpurun_out/r04ah_fuzz_a.txt
