#!/bin/bash
mkdir -p gpurun_out/r3c
timeout 600 python tools/fast_phases.py 64 20 OVS_FAST_CELLS=6 OVS_FAST_CELLS=6,OVS_FAST_DBG=1 OVS_FAST_CELLS=6,OVS_FAST_DBG=4 OVS_FAST_CELLS=6,OVS_FAST_DBG=5 OVS_FAST_CELLS=6,OVS_FAST_DBG=8 OVS_FAST_CELLS=6,OVS_FAST_DBG=16 OVS_FAST_CELLS=1,OVS_FAST_DBG=8 OVS_FAST_CELLS=6 > gpurun_out/r3c/probe.txt 2>&1
cat gpurun_out/r3c/probe.txt
