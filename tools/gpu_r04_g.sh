#!/bin/bash
# round 4, call G: kernel trace + PMC passes of the final kernels (traffic json with the source fingerprints bench.py checks), the bench line with
# them, k_tree step timing of a single 1080p frame (library variant built with -DOVS_TREE_TIMING), tracked-frame trace
cd /root/repo
mkdir -p gpurun_out
tools/gpu_trace.sh r04g_trace > /dev/null 2>&1
tools/gpu_pmc.sh r04g_pmc > /dev/null 2>&1
cp gpurun_out/r04g_pmc/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r04g_bench.json 2> gpurun_out/r04g_bench.err
cut -c1-1200 gpurun_out/r04g_bench.json
head -30 gpurun_out/r04g_trace/summary.txt
OVS_LIB_PATH=$PWD/openvslam_amd/libovslam_hip_tt.so timeout 200 python tools/time_single_frame.py > gpurun_out/r04g_tree_timing.txt 2>&1
grep -m 6 "k_tree n=" gpurun_out/r04g_tree_timing.txt; tail -4 gpurun_out/r04g_tree_timing.txt
timeout 500 tools/gpu_tracked_frame_trace.sh r04g 2>&1 | tail -32
