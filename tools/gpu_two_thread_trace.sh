#!/bin/bash
# rocprofv3 kernel trace of openvslam_amd/cpp/bench_shim (its two_threads section: two extractors on two std::threads) and the digest of
# tools/two_thread_trace.py. Usage (GPU box, repo root): tools/gpu_two_thread_trace.sh <tag>
tag=${1:-r06}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out
mkdir -p $out/tt
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from openvslam_amd.synth import synth_frame
synth_frame(1080, 1920, seed=31).tofile("/tmp/tf_a.raw")
synth_frame(1080, 1920, seed=31, shift=(3, 2), noise_seed=7).tofile("/tmp/tf_b.raw")
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/tt/t -o t -- $root/openvslam_amd/cpp/bench_shim 1080 1920 2000 /tmp/tf_a.raw /tmp/tf_b.raw 50 > $out/tt/t.log 2>&1
cd $root
python -c "import json,sys; d=json.loads(open('$out/tt/t.log').read().strip().splitlines()[-1]); print(json.dumps(d.get('two_threads')))" 2>/dev/null | tee $out/${tag}_two_thread_trace.txt
python tools/two_thread_trace.py $(ls $out/tt/t/*kernel_trace.csv $out/tt/t/*/*kernel_trace.csv 2>/dev/null | head -1) | tee -a $out/${tag}_two_thread_trace.txt
find $out/tt -name '*.csv' -size +4M -delete; find $out/tt -name '*.db' -delete
