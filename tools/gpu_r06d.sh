# round 6, late: k_linearize2 (both halves in one launch) A/B, graph build with the landmark-order pass on the device
set -x
python -m pytest tests/test_gpu_ba.py tests/test_ba_dist.py tests/test_equirect_opt.py -x -q -m gpu 2>&1 | tail -6
OVS_BA_TRACE=1 python tools/time_lba.py device 5 2>&1 | grep -E "total" | tail -3
python tools/lba_lin_sizes.py 2>&1 | tail -6
OVS_BA_LIN_MERGED=0 python tools/lba_lin_sizes.py 2>&1 | tail -6
