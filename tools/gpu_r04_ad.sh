#!/bin/bash
# round 4, call AD: fuzz campaigns on the final kernels of the round (after the local-BA and pose-optimiser work) (extract family with the trig variant in the mix, matcher families, contention family)
cd /root/repo
mkdir -p gpurun_out
timeout 700 python tools/fuzz_parity.py --cases 220 --seed 4401 --out gpurun_out/r04ad_fuzz_a.txt > /dev/null 2>&1; echo "fuzz a rc=$?"; tail -2 gpurun_out/r04ad_fuzz_a.txt
timeout 500 python tools/fuzz_parity.py --cases 150 --seed 4402 --out gpurun_out/r04ad_fuzz_b.txt > /dev/null 2>&1; echo "fuzz b rc=$?"; tail -2 gpurun_out/r04ad_fuzz_b.txt
timeout 300 python tools/fuzz_parity.py --contention 80 --seed 4403 --out gpurun_out/r04ad_fuzz_contention.txt > /dev/null 2>&1; echo "fuzz c rc=$?"; tail -2 gpurun_out/r04ad_fuzz_contention.txt
timeout 300 python tools/fuzz_parity.py --big 12 --seed 4404 --out gpurun_out/r04ad_fuzz_big.txt > /dev/null 2>&1; echo "fuzz big rc=$?"; tail -2 gpurun_out/r04ad_fuzz_big.txt
grep -c "variant=...1" gpurun_out/r04ad_fuzz_a.txt gpurun_out/r04ad_fuzz_b.txt
