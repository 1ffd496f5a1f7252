#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|^FAILED" > gpurun_out/r04s_pytest_gpu.txt; cat gpurun_out/r04s_pytest_gpu.txt
