#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py tests/test_gpu_stereo.py -x -q -m gpu 2>&1 | grep -v "^Extension\|amdgpu.ids" | tail -4 > gpurun_out/r3u.log
python tools/fast_phases.py 256 5 - - >> gpurun_out/r3u.log 2>&1
python tools/fast_phases.py 1 20 - >> gpurun_out/r3u.log 2>&1
timeout 600 python tools/fuzz_parity.py --cases 60 --seed 91 2>&1 | grep "MISMATCH\|# seed" | tail -5 >> gpurun_out/r3u.log
grep -v amdgpu.ids gpurun_out/r3u.log
