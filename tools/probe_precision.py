#!/usr/bin/env python3
"""How many int8 slices does bed_randomSVD need?  Compares d against a 56-bit (slices=7),
tol=1e-9 solve on the same matrix."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
n, m, k = int(sys.argv[1]), int(sys.argv[2]), 20
gb = ba.bed.synthetic(n, m)
ref = ba.bed_randomSVD(gb, k=k, tol=1e-9, slices=7, return_uv=False)
print("ref: niter", ref["niter"], "nops", ref["nops"], "conv", ref["converged"])
for S, blk in ((4, 8), (3, 8), (3, 5), (2, 8), (4, 4)):
    r = ba.bed_randomSVD(gb, k=k, tol=1e-4, slices=S, block=blk, return_uv=False)
    err = np.abs(r["d"] / ref["d"] - 1).max()
    print(json.dumps(dict(slices=S, block=blk, niter=r["niter"], nops=r["nops"], conv=r["converged"],
                          resid=r["max_rel_resid"], max_rel_err_d=err, gpu_ms=r["gpu_ms"])))
