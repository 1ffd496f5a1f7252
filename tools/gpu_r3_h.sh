#!/bin/bash
# round 3: variants on the device
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_edge_cases.py tests/test_gpu_stereo.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3h_orb.log
cat gpurun_out/r3h_orb.log
