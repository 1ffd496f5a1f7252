#!/bin/bash
# round 4, call B: does k_fast_cells leave room for co-resident pyramid / tree / describe workgroups when it is held to six (five) workgroups per CU?
# bench.py under schedule alternatives x OVS_FAST_PAD_LDS (extra dynamic LDS per FAST workgroup: 2400 -> 6 per CU + 22.9 KB free, 6300 -> 5 + 27 KB)
cd /root/repo
mkdir -p gpurun_out
out=gpurun_out/r04b_schedule_sweep.txt
: > $out
for pad in 0 2400 6300; do
for args in "" "--pipeline 2" "--pipeline 4" "--chains 2" "--fast-split 0 --pipeline 2"; do
  OVS_FAST_PAD_LDS=$pad timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ba $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['stage_ms_per_step']
print('pad %5s %-28s value %.1f M/s  ms_per_step %.3f  stages pyr %.2f fast %.2f tree %.2f desc %.2f  match %.2f + %.2f' % ('$pad', '$args', d['value']/1e6, d['ms_per_step'], s['pyramid'], s['fast'], s['tree'], s['describe'], s['match_near'], s['match_resolve']))" >> $out 2>&1
done
done
cat $out
tools/gpu_pmc_sq.sh r04b_pmcsq 2>&1 | tail -30
