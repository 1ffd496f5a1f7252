#!/bin/bash
# bench.py under the alternatives to its default schedule (same box, same inputs): gpurun_out/schedule_sweep.txt
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/schedule_sweep.txt
for args in "" "--chains 2" "--pipeline 2" "--pipeline 4" "--chains 2 --batch 512" ""; do
  python bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-ba --live-pmc 0 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s value %.1f M/s  ms_per_step %.3f  frames/step %d  match stages in schedule %.2f + %.2f ms' % ('$args', d['value']/1e6, d['ms_per_step'], d['config']['frames_per_step_per_gpu'], d['stage_ms_per_step']['match_near'], d['stage_ms_per_step']['match_resolve']))" >> gpurun_out/schedule_sweep.txt 2>&1
done
cat gpurun_out/schedule_sweep.txt
