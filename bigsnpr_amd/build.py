"""In-tree build of libbigsnpr_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m bigsnpr_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbigsnpr_hip.so")
SOURCES = ["image.hip", "matvec.hip", "svd.hip", "ld.hip", "api.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC]
LINK = []


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "bigsnpr_hip.h"))
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + LINK)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
