#!/bin/bash
# round 4, call D: run-time variants (trig / stereo / pose / angle) on the device, fuse landmark staging, TSAN build beside HIP (setarch -R)
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04d_pytest_gpu.txt
cat gpurun_out/r04d_pytest_gpu.txt
timeout 300 python -m pytest tests/test_cpp_shim.py -m gpu -q -s -k "residency" 2>&1 | grep -E "^ok|^FAIL|^skip|reference|ThreadSanitizer|passed|failed" > gpurun_out/r04d_threads.txt
cat gpurun_out/r04d_threads.txt
timeout 400 python tools/class_latency.py 1080 1920 2000 200 > gpurun_out/r04d_class_latency.json 2> gpurun_out/r04d_class_latency.err
python -c "
import json
d=json.load(open('gpurun_out/r04d_class_latency.json'))
print(json.dumps(d.get('mapping_fuse')))"
