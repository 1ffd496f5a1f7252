#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short bench run; writes gpurun_out/<tag>/ (csv) and a per-kernel summary.
tag=${1:-trace}
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag
mkdir -p $out
cmd="python $PWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ba --batch 128"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o t -- $cmd > $out/t.log 2>&1
cd - > /dev/null
python tools/trace_summary.py $out/t/t_kernel_trace.csv > $out/summary.txt 2>&1
cat $out/summary.txt
