#!/bin/bash
# round 4, call U: pose optimiser of one frame spread over several workgroups (grid barrier per pass): parity, determinism, latency
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pose.py tests/test_gpu_fuzz.py -q 2>&1 | grep -E "passed|failed|^E  " | head
OVS_POSE_GROUPS=1 timeout 300 python tools/pose_groups_probe.py 2>&1 | grep "^groups" > gpurun_out/r04u_pose_groups.txt
timeout 300 python tools/pose_groups_probe.py 2>&1 | grep "^groups" >> gpurun_out/r04u_pose_groups.txt
OVS_POSE_GROUPS=4 timeout 300 python tools/pose_groups_probe.py 2>&1 | grep "^groups" >> gpurun_out/r04u_pose_groups.txt
cat gpurun_out/r04u_pose_groups.txt
