#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pose.py tests/test_gpu_fuzz.py tests/test_cpp_shim.py -q -m gpu 2>&1 | grep -E "passed|failed|^E  " | head
for g in 1 4; do OVS_POSE_GROUPS=$g timeout 300 python tools/pose_groups_probe.py 2>&1 | grep "^groups"; done > gpurun_out/r04w_pose_groups.txt
cat gpurun_out/r04w_pose_groups.txt
timeout 400 python tools/class_latency.py 1080 1920 2000 200 > gpurun_out/r04w_class_latency.json 2> gpurun_out/r04w_class_latency.err
python -c "
import json
d=json.load(open('gpurun_out/r04w_class_latency.json'))
print(json.dumps(d.get('tracking_per_frame'))[:300])"
