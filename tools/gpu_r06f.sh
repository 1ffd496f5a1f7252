# round 6, late: k_reduce_scalars with one-barrier trees (same bits expected: chi2 785251.1567330412 -> 58513.97509433431)
set -x
export TMPDIR=/tmp
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from openvslam_amd import ba, synth
d = synth.synth_local_ba(seed=0, pose_noise=0.03, point_noise=0.03)
r = ba.local_ba_optimize(d["poses"], d["pose_fixed"], d["points"], d["edges"], d["cam"])
print("info", [repr(float(x)) for x in r["info"]])
PY
bash tools/gpu_r06.sh r06at lba 2>&1 | grep -E "local_ba_optimize|k_reduce|k_schur|k_lm|k_lin"
