#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python -c "
from openvslam_amd.synth import synth_frame
synth_frame(480,752,seed=21).tofile('/tmp/a.raw'); synth_frame(480,752,seed=21,shift=(4,3),noise_seed=5).tofile('/tmp/b.raw')"
timeout 300 openvslam_amd/cpp/test_fault_shim 480 752 1000 /tmp/a.raw /tmp/b.raw > gpurun_out/r3j_fault.log 2>&1; echo "rc=$?" >> gpurun_out/r3j_fault.log
cat gpurun_out/r3j_fault.log
timeout 1500 python -m pytest tests/test_cpp_shim.py tests/test_gpu_window.py tests/test_gpu_stereo.py tests/test_map_io.py tests/test_vocab_io.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3j.log
cat gpurun_out/r3j.log
