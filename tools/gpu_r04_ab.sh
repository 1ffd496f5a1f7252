#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_orb.py -q 2>&1 | grep -E "passed|failed|^E  " | head
