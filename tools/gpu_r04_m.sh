#!/bin/bash
cd /root/repo
OVS_BA_TRACE=1 timeout 120 python tools/solve_probe.py 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_ba.py -q -k "equirect and 0.03-True" 2>&1 | grep -E "assert|Error|error|passed|failed" | head -20
