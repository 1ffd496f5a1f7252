#!/bin/bash
# round 4, call I: single-buffer k_fast_cells only (two-buffer build removed), bench.py with its own PMC child passes, full GPU suite,
# the fisheye / radial-division check of the threads shim test, refreshed pmc_traffic.json
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r04i_pytest_gpu.txt; cat gpurun_out/r04i_pytest_gpu.txt
(cd openvslam_amd/cpp && timeout 120 ./test_threads_shim) > gpurun_out/r04i_threads.txt 2>&1; echo "threads rc=$?"; tail -5 gpurun_out/r04i_threads.txt
timeout 900 python bench.py > gpurun_out/r04i_bench.json 2> gpurun_out/r04i_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04i_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
r=d['roofline']; print({k:(v if not isinstance(v,str) else v[:200]) for k,v in r.items()})
print(d.get('roofline_valu',{}).get('k_fast_cells'))
PY
tail -3 gpurun_out/r04i_bench.err
timeout 1400 bash tools/gpu_pmc.sh r04i_pmc > /dev/null 2>&1; tail -3 gpurun_out/r04i_pmc/summary.txt; cat gpurun_out/r04i_pmc/pmc_traffic.json | head -c 600
