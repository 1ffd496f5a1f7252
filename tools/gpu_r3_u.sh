#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | grep -v "^Extension\|amdgpu.ids" | tail -4 > gpurun_out/r3u.log
python tools/fast_phases.py 256 5 - OVS_RESIZE_V4=1 - OVS_RESIZE_V4=1 >> gpurun_out/r3u.log 2>&1
python tools/fast_phases.py 32 5 - OVS_RESIZE_V4=1 >> gpurun_out/r3u.log 2>&1
python tools/fast_phases.py 1 20 - OVS_RESIZE_V4=1 >> gpurun_out/r3u.log 2>&1
grep -v amdgpu.ids gpurun_out/r3u.log
