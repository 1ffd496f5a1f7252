"""SURVEY section 5 (sanitizer builds): the CPU oracle and the host side of the class shims under AddressSanitizer + UndefinedBehaviorSanitizer.
  * the oracle: `make -C oracle asan` (gcc) -> liboracle_asan.so, loaded into a python subprocess that preloads gcc's libasan and runs the oracle's own
    CPU tests (known-answer tests, golden vectors, the second matcher family, the BA restatement) against it: no report, same results;
  * the shims' failure policy without a device: test_policy_host built by clang with the sanitised library (`make -C openvslam_amd/cpp asan`).
The device-side programs (test_shim_asan, test_fault_shim_asan, test_threads_shim_asan, test_lba_shim_asan) need a GPU: tools/run_asan.sh runs them there
(profiles/r05_asan_*.txt)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT_MARKS = ("ERROR: AddressSanitizer", "runtime error:", "ERROR: LeakSanitizer", "AddressSanitizer:DEADLYSIGNAL")


def _gcc_runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_under_asan_and_ubsan():
    asan, ubsan = _gcc_runtime("libasan.so"), _gcc_runtime("libubsan.so")
    if not asan:
        pytest.skip("gcc's libasan is not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ)
    env["OVS_ORACLE_LIB"] = "liboracle_asan.so"
    env["LD_PRELOAD"] = asan + ((":" + ubsan) if ubsan else "")
    # python itself leaks by design (interned objects, arenas): leak detection off; everything else aborts the process with a report
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["OMP_NUM_THREADS"] = "2"
    tests = ["tests/test_oracle_kat.py", "tests/test_golden.py", "tests/test_oracle_match2.py", "tests/test_ba.py", "tests/test_pattern.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu"] + tests, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert not any(m in out for m in REPORT_MARKS), out[-4000:]
    assert r.returncode == 0 and " passed" in r.stdout, out[-4000:]


def test_shim_failure_policy_under_asan_and_ubsan():
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang) or not shutil.which("make"):
        pytest.skip("no clang to build the sanitised shims with")
    cpp = os.path.join(ROOT, "openvslam_amd", "cpp")
    subprocess.check_call(["make", "-s", "-C", cpp, "test_policy_host_asan"])
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=1:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([os.path.join(cpp, "test_policy_host_asan")], capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout + r.stderr
    assert not any(m in out for m in REPORT_MARKS), out[-4000:]
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "ALL OK", out[-4000:]
