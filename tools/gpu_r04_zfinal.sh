#!/bin/bash
# round 4, call Z: full GPU suite + default bench on the final state of the round
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|^FAILED" > gpurun_out/r04z_pytest_gpu.txt; cat gpurun_out/r04z_pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r04z_bench.json 2> gpurun_out/r04z_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04z_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:60])
def find(o,k):
    if isinstance(o,dict):
        if k in o: return o[k]
        for v in o.values():
            r=find(v,k)
            if r is not None: return r
print({k:v for k,v in find(d,'config4_local_ba_optimize').items() if k!='note'})
print(find(d,'tracking_per_frame_mean_of_scenes'))
print(find(d,'pose_optimizer_2000_obs'))
PY
tail -2 gpurun_out/r04z_bench.err
