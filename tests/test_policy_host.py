"""The failure policy of the C++ class shims (openvslam_amd/cpp/openvslam/util/device_policy.h) on a box without a device: run_guarded's
retry / empty-result rule under scripted statuses, and -- when no HIP device is present, as in the CPU container -- the classes with
upstream's signatures answering with their empty results instead of throwing (the hot-path functions cannot fail).
The injected-failure variant on a real device is tests/test_cpp_shim.py::test_shims_never_throw_on_device_failures."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_policy_on_the_host():
    cpp = os.path.join(ROOT, "openvslam_amd", "cpp")
    subprocess.check_call(["make", "-s", "-C", cpp, "test_policy_host"])
    r = subprocess.run([os.path.join(cpp, "test_policy_host")], capture_output=True, text=True, timeout=120)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines[-1] == "ALL OK", r.stdout + r.stderr
    assert not any(l.startswith("FAIL") for l in lines)
    assert sum(l.startswith("ok") for l in lines) >= 7          # the run_guarded checks always run
    # nothing but the policy's own log lines on stderr
    assert all(l.startswith("[openvslam_amd]") for l in r.stderr.strip().splitlines() if l)
