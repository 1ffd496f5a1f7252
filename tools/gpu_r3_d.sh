#!/bin/bash
mkdir -p gpurun_out/r3d
timeout 1200 python -m pytest tests/test_cpp_shim.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 > gpurun_out/r3d/pytest_shim.txt
cat gpurun_out/r3d/pytest_shim.txt
