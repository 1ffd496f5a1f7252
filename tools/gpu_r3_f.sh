#!/bin/bash
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_gpu_window.py tests/test_cpp_shim.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 > gpurun_out/r3f/pytest.txt
cat gpurun_out/r3f/pytest.txt
python tools/class_latency.py > gpurun_out/r3f/class_latency.json 2>&1
cat gpurun_out/r3f/class_latency.json | tail -3
