#!/bin/bash
# Round-5 GPU calls, one parameterised script (replaces the one-off tools/gpu_r04_*.sh). Usage: tools/gpu_r05.sh <tag> <step> [<step> ...]
# steps: pytest | bench | trace | pmc | tie | fuzz | latency | lba | custom:<cmd>
cd /root/repo
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for step in "$@"; do
case "$step" in
pytest)
  timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/${tag}_pytest_gpu.txt; cat gpurun_out/${tag}_pytest_gpu.txt ;;
bench)
  timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
  python tools/bench_digest.py gpurun_out/${tag}_bench.json; tail -2 gpurun_out/${tag}_bench.err ;;
trace)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${tag}_trace -o t -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ba --live-pmc 0 > /root/repo/gpurun_out/${tag}_trace.log 2>&1 )
  python tools/trace_digest.py gpurun_out/${tag}_trace gpurun_out/${tag}_kernel_stats.txt
  find gpurun_out/${tag}_trace -name '*.csv' -size +4M -delete ;;
pmc)
  timeout 1400 bash tools/gpu_pmc.sh ${tag}_pmc > /dev/null 2>&1; head -c 300 gpurun_out/${tag}_pmc/pmc_traffic.json ;;
tie)
  timeout 600 python tools/tie_fuzz_gpu.py 300 > gpurun_out/${tag}_tie_fuzz.txt 2>&1; echo "tie fuzz rc=$?"; tail -8 gpurun_out/${tag}_tie_fuzz.txt ;;
fuzz)
  timeout 900 python tools/fuzz_parity.py --seeds 3 > gpurun_out/${tag}_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -8 gpurun_out/${tag}_fuzz.txt ;;
custom:*)
  cmd="${step#custom:}"; echo "+ $cmd"; timeout 1200 bash -c "$cmd" ;;
*) echo "unknown step $step" ;;
esac
done
