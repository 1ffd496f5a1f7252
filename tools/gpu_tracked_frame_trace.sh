#!/bin/bash
# rocprofv3 kernel trace of the tracked-frame loop of openvslam_amd/cpp/bench_shim (the classes with upstream's signatures:
# extract, match_current_and_last_frames, pose_optimizer::optimize, match_frame_and_landmarks) + the class-boundary latency line.
# Usage (GPU box, repo root): tools/gpu_tracked_frame_trace.sh <tag> -> gpurun_out/<tag>_{class_latency.json,tracked_frame_trace.txt}
tag=${1:-r03}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out
mkdir -p $out/tf
timeout 300 python tools/class_latency.py > $out/${tag}_class_latency.json 2> $out/tf/cl.err
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from openvslam_amd.synth import synth_frame
synth_frame(1080, 1920, seed=31).tofile("/tmp/tf_a.raw")
synth_frame(1080, 1920, seed=31, shift=(3, 2), noise_seed=7).tofile("/tmp/tf_b.raw")
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tf/t -o t -- $root/openvslam_amd/cpp/bench_shim 1080 1920 2000 /tmp/tf_a.raw /tmp/tf_b.raw 50 > $out/tf/t.log 2>&1
cd $root
python tools/trace_summary.py $(ls $out/tf/t/*kernel_trace.csv $out/tf/t/*/*kernel_trace.csv 2>/dev/null | head -1) > $out/${tag}_tracked_frame_trace.txt 2>&1
find $out/tf -name '*.csv' -size +8M -delete
find $out/tf -name '*.db' -delete
cat $out/${tag}_class_latency.json | python -c "import json,sys; d=json.load(sys.stdin); print(json.dumps(d['tracking_per_frame']))"
head -45 $out/${tag}_tracked_frame_trace.txt
