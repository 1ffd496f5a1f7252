#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
for g in 1 4; do
OVS_POSE_GROUPS=$g timeout 500 python tools/class_latency.py 1080 1920 2000 100 > gpurun_out/r04x_class_latency_g$g.json 2> gpurun_out/r04x_class_latency.err
python -c "
import json
d=json.load(open('gpurun_out/r04x_class_latency_g$g.json'))
print('groups=$g', json.dumps(d.get('tracking_per_frame_mean_of_scenes')))
for p in d['tracking_per_frame_scenes']: print('   ', p)"
done
