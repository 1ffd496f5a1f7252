#!/bin/bash
# Round-end check on the GPU box: the whole -m gpu suite, two fuzz seeds, the default bench line, the class-boundary latency line and the
# kernel trace of the tracked frame. Usage (repo root): tools/gpu_final.sh <tag> -> gpurun_out/<tag>_{pytest_gpu.txt,bench.json,...}
tag=${1:-r03}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|^FAILED" > gpurun_out/${tag}_pytest_gpu.txt
for sd in 1001 1002; do timeout 300 python tools/fuzz_parity.py --seed $sd --cases 80 2>&1 | tail -1 >> gpurun_out/${tag}_pytest_gpu.txt; done
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
bash tools/gpu_tracked_frame_trace.sh $tag > gpurun_out/${tag}_tracked.log 2>&1
cat gpurun_out/${tag}_pytest_gpu.txt
python - <<PY
import json
d = json.loads(open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['stage_ms_per_step_each_kernel_alone'], d['roofline']['frac'], d['roofline']['traffic'])
print(d['other_configs']['pose_optimizer_2000_obs'], d['class_boundary_latency']['tracking_per_frame'])
PY
head -3 gpurun_out/${tag}_tracked.log
