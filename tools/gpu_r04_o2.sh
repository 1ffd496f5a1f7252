#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_cpp_shim.py -q -m gpu 2>&1 | tail -6
OVS_BA_TRACE=1 timeout 300 python tools/time_lba.py device 2 2>&1 | grep "ovs_" | tail -2
