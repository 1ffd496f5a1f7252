# round 6, late: k_schur on per-graph pair lists, cooperative record gather; per-kernel times of a call
set -x
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python -m pytest tests/test_gpu_ba.py tests/test_ba_dist.py tests/test_cpp_shim.py -x -q -m gpu > /tmp/pt.log 2>&1; grep -E "passed|failed|error" /tmp/pt.log | tail -3
OVS_BA_TRACE=1 python tools/time_lba.py device 5 2>&1 | grep -E "total" | tail -2
OVS_BA_SCHUR_COOP=0 python tools/time_lba.py device 4 2>&1 | tail -2
python tools/fuzz_parity.py --cases 30 --seed 612 2>&1 | tail -2
