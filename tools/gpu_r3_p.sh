#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
OVS_FAST_TIMING=1 python tools/fast_phases.py 64 2 - > gpurun_out/r3p_timing.log 2>&1
tail -40 gpurun_out/r3p_timing.log
