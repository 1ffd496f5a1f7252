#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pose.py -q 2>&1 | grep -E "passed|failed|^E  " | head
for g in 1 2 4 8; do OVS_POSE_GROUPS=$g timeout 300 python tools/pose_groups_probe.py 2>&1 | grep "^groups"; done > gpurun_out/r04u_pose_groups_xcd.txt
cat gpurun_out/r04u_pose_groups_xcd.txt
