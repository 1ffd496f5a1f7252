"""bench.py's launch contract: the driver runs plain `python bench.py --gpus N`; the script starts its own ranks under torch.distributed.run
when WORLD_SIZE is unset, and stops with a clear message -- before allocating anything -- when the node has fewer devices than asked for."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=env)


def test_refuses_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the GPU-less tier's test")
    for n in ("1", "2"):
        r = _run("--gpus", n, "--steps", "1", "--warmup", "0")
        assert r.returncode != 0 and "needs a HIP device" in r.stderr and r.stdout.strip() == ""   # no JSON line, no CPU fallback


@pytest.mark.gpu
def test_asking_for_more_devices_than_the_node_has_stops_early():
    import torch
    have = torch.cuda.device_count()
    r = _run("--gpus", str(have + 1), "--steps", "1", "--warmup", "0", timeout=300)
    assert r.returncode != 0 and ("needs %d HIP devices on this node, found %d" % (have + 1, have)) in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_single_gpu_line_has_the_contract_fields():
    """One short run (2 steps, small side sections off): ONE JSON line on stdout with the fields the driver and the judge read."""
    r = _run("--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-ba", "--batch", "32", timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "u8" and d["vs_baseline"] is None and d["value"] > 0
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "hbm" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert "workload" in d["config"] and "model" not in d["config"]


@pytest.mark.gpu
def test_parity_field_is_a_measurement_of_the_timed_step():
    """`parity` in the bench line is computed, not printed: the last timed step's keypoint records, descriptors and match pairs (64-frame 1080p
    batch, level-0 split, matcher overlapped on the second stream) against the CPU oracle on >= 8 of those frames; the popcount-form run of the
    same schedule must leave the same outputs. A mismatch makes bench.py exit non-zero."""
    r = _run("--steps", "3", "--warmup", "1", "--no-ba", "--batch", "64", "--live-pmc", "0", "--cpu-budget-s", "1", timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    par = d["parity"]
    assert par["bit_exact"] is True and par["checked_frames"] >= 8 and par["checked_match_problems"] >= 8
    assert d["popcount_near_path_same_outputs_as_value_run"] is True and d["value_popcount_near_path"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_two_ranks_code_path_on_one_device():
    """`bench.py --gpus 2` end to end on a one-GPU box: the script starts its two ranks itself (torch.distributed.run on 127.0.0.1); the
    OVS_BENCH_ONE_DEVICE test hook puts both on device 0 over gloo. Checks what the driver's scaling run relies on: ONE JSON line, from rank 0,
    n_gpus = 2, weak scaling (twice the frames of a rank), the sharded local-BA sections present with a non-zero exchange size."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OVS_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "32", "--no-cpu-baseline"], capture_output=True,
                       text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["frames_per_step_per_gpu"] == 32 and "test_hook" in d
    assert abs(d["frames_per_sec"] * d["ms_per_step"] / 1e3 - 64) < 0.5   # both ranks' frames in the whole-job rate
    assert d["keypoints_per_frame"] > 1500 and d["value"] > 0
    for sec in ("local_ba", "local_ba_large"):
        assert d[sec]["allreduce_bytes"] > 0 and d[sec]["ms_per_linearisation"] > 0 and "all-reduce" in d[sec]["exchange"]


def test_live_pmc_parses_the_profilers_counter_csv(tmp_path, monkeypatch):
    """bench.py's own PMC passes (roofline.traffic measured in the run): the parser against a stand-in `rocprofv3` that writes the csv layout of
    rocprofv3 --pmc (one row per dispatch and counter); (2 * FETCH_SIZE + WRITE_SIZE) KiB per k_fast_cells launch, other kernels ignored; a
    profiler that fails or produces nothing gives (None, reason), never an exception."""
    import stat
    import sys
    sys.path.insert(0, ROOT)
    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text("""#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
out, counter = a[a.index("-d") + 1], a[a.index("--pmc") + 1]
if os.environ.get("FAKE_ROCPROF_FAIL"):
    sys.exit(1)
os.makedirs(os.path.join(out, "host", "1234"), exist_ok=True)
val = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 48.0, "SQ_INSTS_VALU": 7.0e6}[counter]
with open(os.path.join(out, "host", "1234", "p_counter_collection.csv"), "w") as f:
    f.write('"Correlation_Id","Dispatch_Id","Agent_Id","Kernel_Name","Counter_Name","Counter_Value"\\n')
    for i in range(3):
        f.write('%d,%d,"Agent 4","void ovs::k_fast_cells<false>(ovs::FrameGeo const*, int)","%s",%f\\n' % (i, i, counter, val + i - 1))
        f.write('%d,%d,"Agent 4","ovs::k_describe(ovs::FrameGeo const*)","%s",%f\\n' % (i, i, counter, 5.0e9))
""")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    got, note = bench.live_pmc_traffic("ovs::k_fast_cells", batch=64, timeout_s=30)
    assert got == {"bytes": int((2 * 1000.0 + 48.0) * 1024), "insts_valu": 7000000, "batch": 64} and "64 frames" in note
    monkeypatch.setenv("FAKE_ROCPROF_FAIL", "1")
    got, note = bench.live_pmc_traffic("ovs::k_fast_cells", batch=64, timeout_s=30)
    assert got is None and "FETCH_SIZE" in note


def test_committed_pmc_summary_was_collected_from_these_kernel_sources():
    """profiles/pmc_traffic.json is bench.py's fall-back for roofline.traffic and the source of the other stages' counters: its per-stage
    fingerprints (a stage's .hip file + the shared headers of csrc/) must be those of the tree, i.e. the PMC passes were re-run
    (tools/gpu_pmc.sh) after the last change to a kernel's sources."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    src = os.path.join(ROOT, "openvslam_amd", "csrc")
    pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for stage, fp in pm["csrc_sha16_by_stage"].items():
        assert fp == bench.pmc_stage_fingerprint(src, stage), "stale counters for stage %s: re-run tools/gpu_pmc.sh" % stage
