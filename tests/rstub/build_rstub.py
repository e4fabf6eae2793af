"""Builds the test-only shared object: bindings/R/bigsnprhip/src/bigsnpr_hip_shim.c + the stand-in R runtime (rstub.c),
linked against the product library."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SHIM = os.path.join(ROOT, "bindings", "R", "bigsnprhip", "src", "bigsnpr_hip_shim.c")
SO = os.path.join(HERE, "libshim_rstub.so")
WARN = ["-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
        "-Wno-cast-function-type"]   # the DL_FUNC casts of every R registration table
INC = ["-I", os.path.join(HERE, "include"), "-I", os.path.join(ROOT, "include")]


def syntax_check():
    return subprocess.run(["gcc", "-fsyntax-only", "-std=c11", "-D_GNU_SOURCE"] + WARN + INC + [SHIM],
                          capture_output=True, text=True)


def build():
    deps = [SHIM, os.path.join(HERE, "rstub.c"), os.path.join(HERE, "include", "Rinternals.h"),
            os.path.join(ROOT, "include", "bigsnpr_hip.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        libdir = os.path.join(ROOT, "bigsnpr_amd")
        subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-fPIC", "-shared"] + WARN + INC +
                              [SHIM, os.path.join(HERE, "rstub.c"), "-o", SO, "-L", libdir, "-lbigsnpr_hip",
                               "-Wl,-rpath," + libdir, "-lm"])
    return SO
