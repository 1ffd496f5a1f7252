#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_fuzz.py -q 2>&1 | grep -E "passed|failed|^E  " | head -5
timeout 300 python tools/time_lba.py device 6 2>&1 | tail -2
timeout 300 python tools/time_lba.py host 4 2>&1 | tail -1
