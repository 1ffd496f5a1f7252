#!/bin/bash
# round 3: full GPU suite + fuzz campaign (variants, new describe / pyramid / FAST kernels, equirect optimiser)
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r03_pytest_gpu.txt
cat gpurun_out/r03_pytest_gpu.txt
timeout 1500 python tools/fuzz_parity.py --cases 90 --seed 31 --out gpurun_out/r03_fuzz_parity.txt > /dev/null 2>&1; echo "fuzz rc=$?"
tail -3 gpurun_out/r03_fuzz_parity.txt
timeout 600 python tools/fuzz_parity.py --big 8 --seed 32 --out gpurun_out/r03_fuzz_parity_big.txt > /dev/null 2>&1; echo "fuzz big rc=$?"
tail -2 gpurun_out/r03_fuzz_parity_big.txt
