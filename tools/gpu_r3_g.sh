#!/bin/bash
mkdir -p gpurun_out/r3g
timeout 900 python -m pytest tests/test_gpu_stereo.py tests/test_cpp_shim.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 > gpurun_out/r3g/pytest.txt
cat gpurun_out/r3g/pytest.txt
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.err
tail -c 300 gpurun_out/r3g/bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3g/bench.json'))
for k in ('value','ms_per_step','stage_ms_per_step','stage_ms_per_step_each_kernel_alone','roofline','cpu_baseline','other_configs','class_boundary_latency'):
    print(k, json.dumps(j.get(k))[:1500])
PY
