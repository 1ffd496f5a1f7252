#!/bin/bash
# Round-6 GPU calls (same steps as round 5). Usage: tools/gpu_r06.sh <tag> <step> [<step> ...]
# steps: pytest | bench | trace | pmc | tie | fuzz | lba | asan | area | custom:<cmd>
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
  for seed in 5201 5202; do
    timeout 600 python tools/fuzz_parity.py --cases 120 --seed $seed > gpurun_out/${tag}_fuzz_${seed}.txt 2>&1; echo "fuzz seed $seed rc=$? $(tail -1 gpurun_out/${tag}_fuzz_${seed}.txt | cut -c1-200)"
  done ;;
lba)
  timeout 300 python tools/time_lba.py device 6 2>&1 | tail -3
  OVS_BA_TRACE=1 timeout 120 python tools/chol_trace.py 2>&1 | grep "dense solve" | tee gpurun_out/${tag}_chol_phases.txt
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${tag}_lba_prof -o lba -- python /root/repo/tools/time_lba.py device 3 > /dev/null 2>&1 )
  f=$(find gpurun_out/${tag}_lba_prof -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY' > gpurun_out/${tag}_lba_kernel_stats.txt
import csv, sys
print("# rocprofv3 --kernel-trace --stats -- python tools/time_lba.py device 3   (3 calls of ovs_local_ba_optimize at BASELINE config 5: 15 LM trials each)")
for r in csv.DictReader(open(sys.argv[1])):
    print("%-30s calls %4s avg %9.1f us  %7s %%" % (r["Name"].split("(")[0][:30], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  head -14 gpurun_out/${tag}_lba_kernel_stats.txt
  find gpurun_out/${tag}_lba_prof -name '*.csv' -size +1M -delete; find gpurun_out/${tag}_lba_prof -name '*.db' -delete ;;
asan)
  bash tools/run_asan.sh ${tag} ;;
area)
  timeout 120 python tools/area_probe.py 2>&1 | tail -1
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${tag}_area_prof -o a -- python /root/repo/tools/area_probe.py > /dev/null 2>&1 )
  python tools/trace_digest.py gpurun_out/${tag}_area_prof gpurun_out/${tag}_area_kernel_stats.txt | head -12
  find gpurun_out/${tag}_area_prof -name '*.csv' -size +1M -delete; find gpurun_out/${tag}_area_prof -name '*.db' -delete ;;
custom:*)
  cmd="${step#custom:}"; echo "+ $cmd"; timeout 1200 bash -c "$cmd" ;;
*) echo "unknown step $step" ;;
esac
done
