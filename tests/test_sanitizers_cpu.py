"""The CPU-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (VERDICT r3 #7c):
oracle/bsn_oracle.c (the C restatement every parity test leans on) and tests/native (the backend-independent SVD
driver, orth_small.hpp, dense_small.hpp: the SAME sources the product instantiates with the HIP backend) are built
once with -fsanitize=address,undefined and the oracle's golden tests plus the driver tests run on those builds in a
child interpreter with the sanitizer runtimes preloaded.  Any report aborts the child (halt_on_error / abort_on_error)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_and_driver_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes are not installed")
    env = dict(os.environ, BSN_SANITIZE="1", LD_PRELOAD=asan + ":" + ubsan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="4")
    base = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"]
    # two children side by side (about 1.5 min each under the sanitizers); the one golden test that takes a minute on
    # its own there (bed == FBM clumping on three window units) stays out: the same clumping code runs in the PLINK one
    jobs = [base + ["tests/test_oracle_golden.py", "--deselect",
                    "tests/test_oracle_golden.py::test_clumping_bed_equals_fbm_and_unit_invariance"],
            base + ["tests/test_svd_driver_cpu.py"]]
    procs = [subprocess.Popen(j, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for j in jobs]
    for p in procs:
        out, _ = p.communicate(timeout=1500)
        tail = out[-4000:]
        assert p.returncode == 0, tail
        assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    # the children really ran on the sanitized builds
    assert os.path.exists(os.path.join(ROOT, "oracle", "libbsn_oracle_san.so"))
    assert os.path.exists(os.path.join(ROOT, "tests", "native", "libnative_test_san.so"))
