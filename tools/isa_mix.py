#!/usr/bin/env python3
"""Instruction mix of the hot loops of the streaming kernels (runs without a GPU).

    python tools/isa_mix.py [substring-of-mangled-name ...]

Compiles matvec.hip to gfx950 assembly and prints, per kernel whose mangled name contains one of
the substrings, the instruction counts of its largest loop body (between the loop label and the
backward branch)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bigsnpr_amd", "csrc", os.environ.get("ISA_SRC", "matvec.hip"))


def main():
    pats = sys.argv[1:] or ["k_prodILi1ELb1ELb1ELb1E", "k_cprodILi1ELi2ELi512ELb1ELb0E"]
    out = "/tmp/bsn_matvec.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I",
                           os.path.join(ROOT, "include"), "-I", os.path.dirname(SRC), "-S",
                           "--cuda-device-only", "-o", out, SRC] + os.environ.get("EXTRA", "").split(),
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN3bsn\w+):", lines[i])
        if m and any(p in m.group(1) for p in pats):
            name = m.group(1)
            j = i + 1
            body = []
            while j < len(lines) and not lines[j].startswith("\t.section") and not lines[j].startswith(".Lfunc_end"):
                body.append(lines[j])
                j += 1
            # loops: label .LBBx_y ... s_cbranch* .LBBx_y (backward)
            labels = {}
            for k, l in enumerate(body):
                mm = re.match(r"^(\.LBB\d+_\d+):", l)
                if mm:
                    labels[mm.group(1)] = k
            best = None
            for k, l in enumerate(body):
                mm = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
                if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
                    span = (labels[mm.group(1)], k)
                    if best is None or span[1] - span[0] > best[1] - best[0]:
                        best = span
            print(name[:90])
            if best:
                ops = collections.Counter()
                for l in body[best[0]:best[1]]:
                    mm = re.match(r"\s+([vsd]\w+|buffer_\w+|global_\w+)", l)
                    if mm:
                        ops[mm.group(1)] += 1
                tot_v = sum(v for k, v in ops.items() if k.startswith("v_") and "mfma" not in k)
                print("   loop of %d lines; VALU %d, MFMA %d" % (best[1] - best[0], tot_v,
                                                             sum(v for k, v in ops.items() if "mfma" in k)))
                for k, v in ops.most_common(22):
                    print("     %-28s %d" % (k, v))
            i = j
        else:
            i += 1


if __name__ == "__main__":
    main()
