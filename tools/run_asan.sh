#!/bin/bash
# The sanitised builds on a GPU box: every shim program of tests/test_cpp_shim.py in its AddressSanitizer + UndefinedBehaviorSanitizer build (clang:
# the class shims AND the host side of libovslam_hip_asan.so instrumented; the device code objects are the product's), on the tests' own inputs with the
# tests' own comparisons against the oracle. A report aborts the program, i.e. fails its test.
# usage: tools/run_asan.sh <tag> -> gpurun_out/<tag>_asan_shims.txt ; build first (CPU is enough): make -C openvslam_amd/csrc asan && make -C openvslam_amd/cpp asan
cd /root/repo
tag=${1:-r05}
mkdir -p gpurun_out
# the HIP runtime maps device memory into what ASan calls the shadow gap; leaks inside the runtime are not ours to check
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:protect_shadow_gap=0:abort_on_error=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
OVS_SHIM_SUFFIX=_asan timeout 1500 python -m pytest tests/test_cpp_shim.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/${tag}_asan_shims.txt
tail -5 gpurun_out/${tag}_asan_shims.txt
( cd openvslam_amd/cpp && ./test_policy_host_asan > /root/repo/gpurun_out/${tag}_asan_policy_host.txt 2>&1; echo "test_policy_host_asan rc=$? $(tail -1 /root/repo/gpurun_out/${tag}_asan_policy_host.txt)" )
