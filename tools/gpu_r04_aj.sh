#!/bin/bash
# round 4, call AJ: the pose optimiser's retries 1..9 evaluated in one pass: bit-equality with the sequential form, parity, timing
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pose.py tests/test_gpu_fuzz.py -q 2>&1 | grep -E "passed|failed|^E  " | head -6
(for b in 1 0; do for g in 1 4; do OVS_POSE_BATCH_RETRIES=$b OVS_POSE_GROUPS=$g timeout 300 python tools/pose_groups_probe.py 2>&1 | grep "^groups" | sed "s/^/batch_retries=$b /"; done; done) > gpurun_out/r04aj_pose_batched_retries.txt
cat gpurun_out/r04aj_pose_batched_retries.txt | cut -c1-110
