#!/bin/bash
cd /root/repo
OVS_BA_TRACE=1 timeout 120 python tools/solve_probe.py 2>&1 | grep -v "^\[k_chol" | tail -3
OVS_BA_TRACE=1 timeout 120 python tools/solve_probe.py 2>&1 | grep "^\[k_chol" | awk 'NR%3==0'
timeout 900 python -m pytest tests/test_gpu_ba.py -q 2>&1 | tail -3
timeout 300 python tools/time_lba.py device 5 2>&1 | tail -2
