#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 500 python tools/class_latency.py 1080 1920 2000 200 > gpurun_out/r04aa_class_latency.json 2> gpurun_out/r04aa_class_latency.err
python -c "
import json
d=json.load(open('gpurun_out/r04aa_class_latency.json'))
print(d['two_threads'])"
tail -2 gpurun_out/r04aa_class_latency.err
