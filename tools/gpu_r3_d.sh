#!/bin/bash
mkdir -p gpurun_out/r3d
timeout 1200 python -m pytest tests/test_gpu_pose.py tests/test_gpu_ba.py -x -q 2>&1 | tail -15 > gpurun_out/r3d/pytest.txt
cat gpurun_out/r3d/pytest.txt
