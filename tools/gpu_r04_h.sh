#!/bin/bash
# round 4, call H: k_tree<1024> with the candidate list cached in LDS (single-frame launches): parity + single-frame / tracked-frame latency
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_stereo.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/fuzz_parity.py --cases 60 --seed 4201 --out gpurun_out/r04h_fuzz.txt > /dev/null 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r04h_fuzz.txt
timeout 200 python tools/time_single_frame.py > gpurun_out/r04h_single_frame.txt 2>&1; cat gpurun_out/r04h_single_frame.txt | tail -6
timeout 400 python tools/class_latency.py 1080 1920 2000 200 > gpurun_out/r04h_class_latency.json 2> gpurun_out/r04h_class_latency.err
python -c "
import json
d=json.load(open('gpurun_out/r04h_class_latency.json'))
print(json.dumps(d.get('tracking_per_frame'))[:420]); print({k:v for k,v in d.items() if k.startswith('staged') or k.startswith('pageable')})"
