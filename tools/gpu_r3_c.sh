#!/bin/bash
mkdir -p gpurun_out/r3c
timeout 600 python tools/fast_phases.py 64 20 OVS_FAST_V3=1 - OVS_FAST_DIAM8=1 - > gpurun_out/r3c/sweep_a.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3c/sweep_a.txt
export OVS_LIB_PATH=$PWD/openvslam_amd/libovslam_hip_occ8.so
timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -2
timeout 600 python tools/fast_phases.py 64 20 OVS_FAST_V3=1 - OVS_FAST_DIAM8=1 OVS_FAST_CELLS=4 OVS_FAST_CELLS=8 - > gpurun_out/r3c/sweep_occ8.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3c/sweep_occ8.txt
