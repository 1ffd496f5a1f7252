#!/bin/bash
mkdir -p gpurun_out/r3e
python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r3e/bench_gpus2.txt 2>&1; echo "rc=$?" >> gpurun_out/r3e/bench_gpus2.txt
tail -3 gpurun_out/r3e/bench_gpus2.txt
python tools/ba_multi_bench.py 1 > gpurun_out/r3e/ba_multi_1.txt 2>&1; tail -2 gpurun_out/r3e/ba_multi_1.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r3e/pytest_gpu.txt
cat gpurun_out/r3e/pytest_gpu.txt
