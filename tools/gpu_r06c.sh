# round 6, late: fuse kernel with sixteen lanes per landmark, graph build passes
set -x
make -s -C openvslam_amd/cpp 2>&1 | tail -3
python -m pytest tests/test_gpu_window.py tests/test_gpu_edge_cases.py tests/test_gpu_knife_edge.py tests/test_cpp_shim.py -x -q -m gpu 2>&1 | tail -6
python - <<'PY'
import json, sys
sys.path.insert(0, "tools")
import class_latency
r = class_latency.measure(iters=100, scenes=1)
print(json.dumps(r.get("mapping_fuse")))
PY
OVS_BA_TRACE=1 python tools/time_lba.py device 5 2>&1 | grep -E "graph_create|total" | tail -4
python tools/fuzz_parity.py --cases 40 --seed 611 2>&1 | tail -4
