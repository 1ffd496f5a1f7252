#!/bin/bash
# round 4, call AE: 64-row tiles in the pyramid kernel: parity + per-stage time at 256 frames + single frame
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -q 2>&1 | grep -E "passed|failed|^E  " | head -5
timeout 600 python tools/ab_extract.py 256 6 "" > gpurun_out/r04ae_ab.txt 2>&1; tail -3 gpurun_out/r04ae_ab.txt | cut -c1-200
timeout 600 python tools/ab_extract.py 128 6 "" >> gpurun_out/r04ae_ab.txt 2>&1; tail -1 gpurun_out/r04ae_ab.txt | cut -c1-200
timeout 300 python tools/fuzz_parity.py --cases 60 --seed 4501 --out gpurun_out/r04ae_fuzz.txt > /dev/null 2>&1; tail -1 gpurun_out/r04ae_fuzz.txt
