#!/usr/bin/env python3
"""Accuracy of d, u, v against a 56-bit tol=1e-10 solve for several (block, slices)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
n, m, k = int(sys.argv[1]), int(sys.argv[2]), 20
gb = ba.bed.synthetic(n, m)
ref = ba.bed_randomSVD(gb, k=k, tol=1e-10, slices=7, block=5)
print("ref: niter", ref["niter"], "conv", ref["converged"], flush=True)
def sub_err(a, b):   # max column-wise distance after sign alignment
    s = np.sign((a * b).sum(0))
    return float(np.abs(a * s - b).max()), float(np.linalg.norm(a * s - b, axis=0).max())
for blk, S in ((5, 3), (8, 4), (8, 2), (8, 3)):
    t0 = time.perf_counter(); r = ba.bed_randomSVD(gb, k=k, tol=1e-4, slices=S, block=blk); t = time.perf_counter() - t0
    print(json.dumps(dict(block=blk, slices=S, niter=r["niter"], nops=r["nops"], ms=1e3 * t, conv=r["converged"],
                          d_rel=float(np.abs(r["d"] / ref["d"] - 1).max()), u_err=sub_err(r["u"], ref["u"]),
                          v_err=sub_err(r["v"], ref["v"]))), flush=True)

# true residuals of the (block 8, 2 slices) and (block 5, 3 slices) solves through the 56-bit products
for blk, S in ((8, 2), (5, 3), (8, 4)):
    r = ba.bed_randomSVD(gb, k=k, tol=1e-4, slices=S, block=blk)
    worst = 0.0
    for t in (0, 9, 19):
        av = ba.bed_prodVec(gb, r["v"][:, t], center=r["center"], scale=r["scale"])
        atu = ba.bed_cprodVec(gb, r["u"][:, t], center=r["center"], scale=r["scale"])
        # eigen-residual of A A' for the pair (d^2, u):  || A A' u - d^2 u || / d^2
        aatu = ba.bed_prodVec(gb, atu, center=r["center"], scale=r["scale"])
        res = np.linalg.norm(aatu - r["d"][t] ** 2 * r["u"][:, t]) / r["d"][t] ** 2
        worst = max(worst, res)
        print(json.dumps(dict(block=blk, slices=S, t=t, eig_resid=float(res),
                              av_du=float(np.linalg.norm(av - r["d"][t] * r["u"][:, t]) / r["d"][t]),
                              atu_dv=float(np.linalg.norm(atu - r["d"][t] * r["v"][:, t]) / r["d"][t]))), flush=True)
    print("block", blk, "slices", S, "reported max_rel_resid", r["max_rel_resid"], "true worst of sampled", worst, flush=True)
