#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r3r.log
for args in "" "--chains 2" "--pipeline 2" "--fast-split 0" "--batch 512" "--batch 512 --chains 2" "--overlap 0"; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ba $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s value %.1f M/s  ms_per_step %.3f  frames/step %d' % ('$args', d['value']/1e6, d['ms_per_step'], d['config']['frames_per_step_per_gpu']))" >> gpurun_out/r3r.log 2>&1
done
cat gpurun_out/r3r.log
