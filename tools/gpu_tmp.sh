export TMPDIR=/tmp
python -m pytest tests/test_gpu_ba.py tests/test_ba_dist.py tests/test_equirect_opt.py tests/test_cpp_shim.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
python tools/lba_lin_sizes.py 2>&1 | tail -6
python tools/time_lba.py device 5 2>&1 | tail -3
