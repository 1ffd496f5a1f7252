#!/bin/bash
# round 4, call A: the LDS-DMA k_fast_cells (one / two raw-tile buffers) + merged resolver loads: parity suite, A/B with per-phase timing,
# bench line of both, a fuzz seed + the resolver-contention family, tracked-frame trace.
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04a_pytest_gpu.txt
cat gpurun_out/r04a_pytest_gpu.txt
timeout 600 python tools/ab_extract.py 256 6 "" OVS_FAST_BUFS=2 OVS_FAST_TIMING=1 OVS_FAST_BUFS=2,OVS_FAST_TIMING=1 OVS_FAST_CELLS=4 OVS_FAST_CELLS=8 OVS_FAST_BUFS=2,OVS_FAST_CELLS=8 > gpurun_out/r04a_ab.txt 2>&1
cat gpurun_out/r04a_ab.txt
timeout 300 python tools/ab_extract.py 1 20 "" OVS_FAST_BUFS=2 > gpurun_out/r04a_ab_single.txt 2>&1
cat gpurun_out/r04a_ab_single.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
cut -c1-900 gpurun_out/r04a_bench.json
OVS_FAST_BUFS=2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ba > gpurun_out/r04a_bench_bufs2.json 2> gpurun_out/r04a_bench_bufs2.err
cut -c1-400 gpurun_out/r04a_bench_bufs2.json
timeout 500 python tools/fuzz_parity.py --cases 50 --seed 4001 --out gpurun_out/r04a_fuzz.txt > /dev/null 2>&1; echo "fuzz rc=$?"
tail -2 gpurun_out/r04a_fuzz.txt
OVS_FAST_BUFS=2 timeout 400 python tools/fuzz_parity.py --cases 30 --seed 4002 --out gpurun_out/r04a_fuzz_bufs2.txt > /dev/null 2>&1; echo "fuzz bufs2 rc=$?"
tail -2 gpurun_out/r04a_fuzz_bufs2.txt
timeout 400 python tools/fuzz_parity.py --contention 40 --seed 4003 --out gpurun_out/r04a_fuzz_contention.txt > /dev/null 2>&1; echo "contention rc=$?"
tail -2 gpurun_out/r04a_fuzz_contention.txt
timeout 500 tools/gpu_tracked_frame_trace.sh r04a 2>&1 | tail -50
