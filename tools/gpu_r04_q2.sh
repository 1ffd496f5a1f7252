#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ba.py -q -x 2>&1 | grep -E "^E |passed|failed|^tests/|Error" | head -30
