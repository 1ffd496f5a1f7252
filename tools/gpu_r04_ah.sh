#!/bin/bash
# round 4, call AH: a larger fuzz campaign on the final kernels
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python tools/fuzz_parity.py --cases 900 --seed 4601 --out gpurun_out/r04ah_fuzz_a.txt > /dev/null 2>&1; echo "fuzz a rc=$?"; tail -1 gpurun_out/r04ah_fuzz_a.txt
timeout 600 python tools/fuzz_parity.py --contention 300 --seed 4602 --out gpurun_out/r04ah_fuzz_contention.txt > /dev/null 2>&1; echo "fuzz c rc=$?"; tail -1 gpurun_out/r04ah_fuzz_contention.txt
timeout 600 python tools/fuzz_parity.py --big 40 --seed 4603 --out gpurun_out/r04ah_fuzz_big.txt > /dev/null 2>&1; echo "fuzz big rc=$?"; tail -1 gpurun_out/r04ah_fuzz_big.txt
grep -c "variant=...1" gpurun_out/r04ah_fuzz_a.txt
