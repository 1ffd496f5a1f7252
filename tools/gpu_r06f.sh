export TMPDIR=/tmp
python -m pytest tests/test_gpu_ba.py -x -q -m gpu -k "shapes or more_edges or config5 or edge_counts" 2>&1 | grep -E "passed|failed"
bash tools/gpu_r06.sh r06ax lba 2>&1 | grep -E "local_ba_optimize|k_schur"
