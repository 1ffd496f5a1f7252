#!/bin/bash
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_pose.py -q -k "saturated" 2>&1 | grep -E "passed|failed|^E  |Error" | head
