#!/bin/bash
# round 4, call S: full GPU suite + default bench + PMC refresh after the local-BA work
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r04s_pytest_gpu.txt; cat gpurun_out/r04s_pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r04s_bench.json 2> gpurun_out/r04s_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
s=d.get('side_sections') or {}
def find(o,k):
    if isinstance(o,dict):
        if k in o: return o[k]
        for v in o.values():
            r=find(v,k)
            if r is not None: return r
print(find(d,'config4_local_ba_optimize'))
print(find(d,'tracking_per_frame'))
PY
tail -2 gpurun_out/r04s_bench.err
timeout 1400 bash tools/gpu_pmc.sh r04s_pmc > /dev/null 2>&1; head -c 300 gpurun_out/r04s_pmc/pmc_traffic.json
