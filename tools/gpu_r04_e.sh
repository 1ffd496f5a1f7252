#!/bin/bash
# round 4, call E: full GPU suite after the variants (no -x), TSAN log of the two-thread test
cd /root/repo
mkdir -p gpurun_out
OVS_TSAN_LOG=$PWD/gpurun_out/r04e_tsan.txt timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r04e_pytest_gpu.txt
cat gpurun_out/r04e_pytest_gpu.txt
grep -c "WARNING: ThreadSanitizer" gpurun_out/r04e_tsan.txt
grep -A 30 "WARNING: ThreadSanitizer" gpurun_out/r04e_tsan.txt | head -120
