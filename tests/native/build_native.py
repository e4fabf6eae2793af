import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "libnative_test.so")


def build():
    src = os.path.join(HERE, "native_test.cpp")
    deps = [src] + [os.path.join(ROOT, "bigsnpr_amd", "csrc", f) for f in ("svd_driver.hpp", "dense_small.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall",
                               "-I", os.path.join(ROOT, "bigsnpr_amd", "csrc"), src, "-o", SO])
    return SO
