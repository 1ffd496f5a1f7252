#!/bin/bash
# round 4, call F: k_describe with four keypoints per wave + one-multiply deg2rad: parity, stage times, bench line; local-BA wall-clock trace
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_detmath.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python tools/fuzz_parity.py --cases 40 --seed 4101 --out gpurun_out/r04f_fuzz.txt > /dev/null 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r04f_fuzz.txt
timeout 300 python tools/ab_extract.py 256 6 "" > gpurun_out/r04f_ab.txt 2>&1; cat gpurun_out/r04f_ab.txt
timeout 200 python tools/ab_extract.py 1 20 "" >> gpurun_out/r04f_ab.txt 2>&1; tail -1 gpurun_out/r04f_ab.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err
cut -c1-300 gpurun_out/r04f_bench.json
python -c "
import json
d=json.loads(open('gpurun_out/r04f_bench.json').read().strip().splitlines()[-1])
print(d['stage_ms_per_step']); print(d['stage_ms_per_step_each_kernel_alone']); print(d['other_configs'].get('config4_local_ba_optimize'))"
OVS_BA_TRACE=1 timeout 200 python tools/time_lba.py > gpurun_out/r04f_lba_trace.txt 2>&1; tail -8 gpurun_out/r04f_lba_trace.txt
