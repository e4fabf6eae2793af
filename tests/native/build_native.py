import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# BSN_SANITIZE=1 (tests/test_sanitizers_cpu.py): the same sources under -fsanitize=address,undefined
SAN = bool(os.environ.get("BSN_SANITIZE"))
SO = os.path.join(HERE, "libnative_test_san.so" if SAN else "libnative_test.so")


def build():
    src = os.path.join(HERE, "native_test.cpp")
    deps = [src] + [os.path.join(ROOT, "bigsnpr_amd", "csrc", f) for f in ("svd_driver.hpp", "dense_small.hpp", "orth_small.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if SAN else ["-O2"]
        subprocess.check_call(["g++"] + flags + ["-std=c++17", "-fPIC", "-shared", "-Wall",
                               "-I", os.path.join(ROOT, "bigsnpr_amd", "csrc"), src, "-o", SO])
    return SO


MOCK_RCCL = os.path.join(HERE, "libmock_rccl.so")


def build_mock_rccl():
    """the shared-memory stand-in for librccl.so used by the two-rank tests on a one-GPU box (mock_rccl.cpp)"""
    src = os.path.join(HERE, "mock_rccl.cpp")
    if not os.path.exists(MOCK_RCCL) or os.path.getmtime(src) > os.path.getmtime(MOCK_RCCL):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared",
                               src, "-o", MOCK_RCCL, "-lpthread", "-lrt"])
    return MOCK_RCCL
