#!/bin/bash
# round 4, call C: keyframe residency (resident twins == host forms == oracle), shared caches under two threads (+ TSAN build), mapping-side
# latency, fixed contention test
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04c_pytest_gpu.txt
cat gpurun_out/r04c_pytest_gpu.txt
timeout 300 python -m pytest tests/test_cpp_shim.py -m gpu -q -s -k "residency" 2>&1 | grep -E "^ok|^FAIL|^skip|reference|ThreadSanitizer|passed|failed" > gpurun_out/r04c_threads.txt
cat gpurun_out/r04c_threads.txt
timeout 400 python tools/class_latency.py 1080 1920 2000 200 > gpurun_out/r04c_class_latency.json 2> gpurun_out/r04c_class_latency.err
python -c "
import json
d=json.load(open('gpurun_out/r04c_class_latency.json'))
print(json.dumps(d.get('mapping_fuse')))
print(json.dumps(d.get('tracking_per_frame')))"
