"""In-tree build of libbigsnpr_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m bigsnpr_amd.build [--force] [--ablation]

--ablation builds a SEPARATE library, libbigsnpr_hip_abl.so, with -DBSN_ABLATION: it additionally
contains the profiling variants of the two streaming kernels (no MFMA / no decode / no loads ...,
selected by BSN_TUNE; they compute wrong numbers by construction) that profiles/*ablation*.txt were
measured with.  The product library never contains them; the probes load the ablation build through
BSN_LIB_PATH.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbigsnpr_hip.so")
SOURCES = ["image.hip", "matvec.hip", "svd.hip", "ld.hip", "api.hip", "comm.hip", "tcross.hip", "robust.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC]
LINK = ["-ldl"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, ablation=False):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "bigsnpr_hip.h"))
    objs, jobs = [], []
    lib = LIB.replace(".so", "_abl.so") if ablation else LIB
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".abl.o" if ablation else ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC] + FLAGS + (["-DBSN_ABLATION"] if ablation else []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(lib, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + LINK)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, ablation="--ablation" in sys.argv))
