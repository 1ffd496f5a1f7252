#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_gpu_match.py tests/test_vocab_io.py tests/test_map_io.py tests/test_cpp_shim.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3i.log
cat gpurun_out/r3i.log
