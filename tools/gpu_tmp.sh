#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/tmp.log
for b in 1 4 16 32 256; do python tools/fast_phases.py $b 10 - 2>&1 | grep -v amdgpu.ids >> gpurun_out/tmp.log; done
timeout 600 python -m pytest tests/test_gpu_orb.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -E "passed|failed" >> gpurun_out/tmp.log
python tools/class_latency.py 1080 1920 2000 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['pageable_no_pyramid']); print(d['tracking_per_frame'])" >> gpurun_out/tmp.log
cat gpurun_out/tmp.log
