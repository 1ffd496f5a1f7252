#!/bin/bash
# diagnostic: the sanitised test_shim under different allocator fill settings (does an uninitialised read decide a result?)
cd /root/repo
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo")
from openvslam_amd.synth import synth_frame
import os
os.makedirs("/tmp/asan_in", exist_ok=True)
synth_frame(480, 752, seed=1).tofile("/tmp/asan_in/a.raw")
synth_frame(480, 752, seed=1, shift=(5, 0), noise_seed=99).tofile("/tmp/asan_in/b.raw")
PY
cd openvslam_amd/cpp
base="detect_leaks=0:halt_on_error=1:protect_shadow_gap=0"
echo "plain:"; ./test_shim 480 752 1000 /tmp/asan_in/a.raw /tmp/asan_in/b.raw /tmp/asan_in/out0.bin 2>&1 | tail -2
for opt in "" ":max_malloc_fill_size=0" ":max_malloc_fill_size=268435456:malloc_fill_byte=0" ":max_malloc_fill_size=268435456:malloc_fill_byte=255" ":quarantine_size_mb=0"; do
  echo "ASAN_OPTIONS=$base$opt"; ASAN_OPTIONS="$base$opt" ./test_shim_asan 480 752 1000 /tmp/asan_in/a.raw /tmp/asan_in/b.raw /tmp/asan_in/out1.bin 2>&1 | tail -3
done
