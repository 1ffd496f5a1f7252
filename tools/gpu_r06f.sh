# round 6, late: k_schur_l XCD-aware pair order A/B
set -x
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ba.py -x -q -m gpu > /tmp/pt.log 2>&1; grep -E "passed|failed|error" /tmp/pt.log | tail -3
bash tools/gpu_r06.sh r06ap lba 2>&1 | grep -E "local_ba_optimize|k_schur"
OVS_BA_SCHUR_XCD=0 bash tools/gpu_r06.sh r06aq lba 2>&1 | grep -E "local_ba_optimize|k_schur"
