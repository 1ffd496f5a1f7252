#!/bin/bash
# round 3, call a: parity of the v4 FAST / resize kernels, A/B timing against v3
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -15 > gpurun_out/r3a/pytest_orb.txt
cat gpurun_out/r3a/pytest_orb.txt
timeout 600 python tools/ab_extract.py 64 6 > gpurun_out/r3a/ab.txt 2>&1
cat gpurun_out/r3a/ab.txt
