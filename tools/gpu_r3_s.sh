#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r3s.log
timeout 120 python tools/time_pose.py 2000 100 >> gpurun_out/r3s.log 2>&1
timeout 120 python tools/time_pose.py 500 100 >> gpurun_out/r3s.log 2>&1
timeout 900 python -m pytest tests/test_gpu_pose.py tests/test_cpp_shim.py tests/test_gpu_knife_edge.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/r3s.log
timeout 600 python tools/fuzz_parity.py --cases 60 --seed 77 2>&1 | grep -i "pose\|MISMATCH\|# seed" | tail -25 >> gpurun_out/r3s.log
python tools/class_latency.py 1080 1920 2000 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['tracking_per_frame']))" >> gpurun_out/r3s.log
grep -v amdgpu.ids gpurun_out/r3s.log
