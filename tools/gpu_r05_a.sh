#!/bin/bash
# round 5, call A: the state round 4 left (its last commits were made without a GPU) -- full GPU suite, default bench, kernel-trace stats, PMC refresh
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|^FAILED" > gpurun_out/r05a_pytest_gpu.txt; cat gpurun_out/r05a_pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05a_bench.json').read().strip().splitlines()[-1])
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
tail -2 gpurun_out/r05a_bench.err
timeout 1400 bash tools/gpu_pmc.sh r05a_pmc > /dev/null 2>&1; head -c 200 gpurun_out/r05a_pmc/pmc_traffic.json

# kernel-trace stats of the same bench command (short), for profiles/r05a_kernel_stats.txt
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r05a_trace -o t -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ba --live-pmc 0 > $OLDPWD/gpurun_out/r05a_trace.log 2>&1 )
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r05a_trace/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open('gpurun_out/r05a_kernel_stats.txt', 'w') as out:
        for r in rows[:25]:
            line = "%-60s calls %6s  avg %10.1f ns  total %5.1f %%" % (r['Name'][:60], r['Calls'], float(r['AverageNs']), float(r['Percentage']))
            print(line); out.write(line + "\n")
PY
find gpurun_out/r05a_trace -name '*.csv' -size +4M -delete

# the tie rules on the device (written without one at the end of round 4)
timeout 600 python tools/tie_fuzz_gpu.py 300 > gpurun_out/r05a_tie_fuzz.txt 2>&1; echo "tie fuzz rc=$?"; tail -6 gpurun_out/r05a_tie_fuzz.txt
