"""SURVEY.md §5 sanitizer hook: the CPU oracle, the device math of csrc/ compiled for the host (tests/host_shim) and the host build of the
projected-Newton kernel source, all under AddressSanitizer + UndefinedBehaviorSanitizer (gcc).  GPU ASan is not available on this pool;
the device side has the red-zone guard mode instead (TRAJOPT_GUARD=1, tests/test_gpu_guard.py)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"]
CSRC = ROOT / "trajectoryoptimization.jl_amd" / "csrc"
SHIM = ROOT / "tests" / "host_shim"


def _clean(stderr):
    assert "AddressSanitizer" not in stderr and "runtime error:" not in stderr, stderr[-4000:]


@pytest.mark.parametrize("harness,extra", [("ls_round_harness.cpp", []), ("device_math_harness.cpp", ["-ffp-contract=off"])])
def test_host_shims_under_asan_ubsan(harness, extra, tmp_path):
    exe = tmp_path / "harness"
    subprocess.run(["g++", "-std=c++17", "-O1", *extra, *SAN, "-I", str(SHIM), "-I", str(CSRC), str(SHIM / harness), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    _clean(r.stderr)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])


def test_oracle_and_pn_kernel_source_under_asan_ubsan(tmp_path):
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "asan"], check=True, capture_output=True)
    pn_so = tmp_path / "libpn_host_asan.so"
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-fPIC", "-shared", *SAN, "-I", str(SHIM), "-I", str(CSRC), "-o", str(pn_so),
                    str(SHIM / "pn_harness.cpp")], check=True)
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "sanitizer_driver.py"), str(ROOT / "oracle" / "build" / "liboracle_asan.so"), str(pn_so)],
                       capture_output=True, text=True, env=env, timeout=600)
    _clean(r.stderr)
    assert r.returncode == 0 and "SANITIZER_DRIVER_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
