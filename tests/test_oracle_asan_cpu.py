"""SURVEY section 5's stance on sanitizers: the C restatement (oracle/irbfn_oracle.c, the checker behind every
full-size parity test and the CPU baseline) is built with ``-fsanitize=address,undefined`` and the C-oracle suite is
run against that build in a child process (libasan must be the first DSO, hence LD_PRELOAD).  CPU only -- GPU
sanitizers are not available on this pool."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    gcc = shutil.which("gcc")
    if not gcc:
        return None
    p = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return os.path.realpath(p) if p and os.path.sep in p and os.path.exists(p) else None


def test_c_oracle_suite_under_asan_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("gcc / libasan not available")
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(ROOT, "oracle", "_build", "libirbfn_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan, IRBFN_ORACLE_LIB=lib, OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_oracle_cpu.py", "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
