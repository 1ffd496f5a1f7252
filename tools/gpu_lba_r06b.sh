set -x
python -m pytest tests/test_gpu_ba.py -x -q -m gpu 2>&1 | tail -15
OVS_BA_TRACE=1 python tools/time_lba.py device 6 2>&1 | grep -v graph_create | tail -12
OVS_BA_TRACE=1 python tools/time_lba.py device 4 2>&1 | grep graph_create | tail -3
OVS_BA_BACKSUB_EDGES=0 python tools/time_lba.py device 5 2>&1 | tail -3
OVS_BA_DEV_OUTLIERS=0 python tools/time_lba.py device 5 2>&1 | tail -3
