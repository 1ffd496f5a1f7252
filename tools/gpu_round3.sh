#!/bin/bash
# round 3 measurement round: kernel trace, PMC passes (+ traffic json with source fingerprint), full bench line, side kernels, class latency
cd /root/repo
mkdir -p gpurun_out
tools/gpu_trace.sh r03_trace > /dev/null 2>&1
tools/gpu_pmc.sh r03_pmc > /dev/null 2>&1
cp gpurun_out/r03_pmc/pmc_traffic.json profiles/pmc_traffic.json   # for the bench line below (box-local: only gpurun_out/ travels back -- copy it into profiles/ after the call)
python bench.py --steps 20 --warmup 3 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
cat gpurun_out/r03_bench.json | cut -c1-1500
tools/gpu_profile_others.sh r03_others > /dev/null 2>&1
python tools/class_latency.py 1080 1920 2000 200 > gpurun_out/r03_class_latency_1080p.json 2>/dev/null
cat gpurun_out/r03_class_latency_1080p.json | cut -c1-1200
head -40 gpurun_out/r03_trace/summary.txt
