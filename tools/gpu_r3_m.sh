#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3m.log
python tools/fast_phases.py 256 5 - OVS_DESCRIBE_XCD=0 - >> gpurun_out/r3m.log 2>&1
cat gpurun_out/r3m.log
