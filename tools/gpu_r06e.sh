# round 6, late: Hpl records staged through LDS (coalesced stores) in both forms of the keyframe side
set -x
python -m pytest tests/test_gpu_ba.py tests/test_ba_dist.py tests/test_equirect_opt.py -x -q -m gpu 2>&1 | tail -4
python tools/lba_lin_sizes.py 2>&1 | tail -6
OVS_BA_LIN_MERGED=0 python tools/lba_lin_sizes.py 2>&1 | tail -6
OVS_BA_LIN_MERGED=1 python tools/lba_lin_sizes.py 2>&1 | tail -3
OVS_BA_TRACE=1 python tools/time_lba.py device 5 2>&1 | grep -E "total" | tail -2
