#!/bin/bash
mkdir -p gpurun_out/r3e
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r3e/pytest_gpu.txt
cat gpurun_out/r3e/pytest_gpu.txt
