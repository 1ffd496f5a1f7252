import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from openvslam_amd import feature, synth
rows, cols = 1080, 1920
img = synth.synth_frame(rows, cols, seed=31)
ex = feature.orb_extractor(feature.orb_params(2000), max_rows=rows, max_cols=cols, max_batch=1)
for _ in range(20): k, d = ex.extract(img)
ts = []
for _ in range(300):
    t = time.perf_counter(); k, d = ex.extract(img); ts.append(time.perf_counter() - t)
ts.sort()
import hashlib
print("ZERO_COPY_OUT=%s host extract median %.4f ms p10 %.4f n=%d sha %s" % (os.environ.get("OVS_ORB_ZERO_COPY_OUT", "1"), ts[150] * 1e3, ts[30] * 1e3, len(k), hashlib.sha256(k.tobytes() + d.tobytes()).hexdigest()[:12]))
