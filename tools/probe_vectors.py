#!/usr/bin/env python3
"""Accuracy / cost frontier of bed_randomSVD's singular vectors (round 5): per-vector angle of u and v to a 56-bit
tol-1e-10 solve (another block size), for the precision schedule and for uniform panels of 16 / 24 / 32 / 56 bits.
    python tools/probe_vectors.py [n m [k]]          (default 400000 1000000 20)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
gb = ba.bed.synthetic(n, m)
t0 = time.perf_counter()
ref = ba.bed_randomSVD(gb, k=k, tol=1e-10, slices=7, block=4)
print(json.dumps(dict(ref=dict(niter=ref["niter"], converged=ref["converged"], s=round(time.perf_counter() - t0, 2),
                               relgap=[float(x) for x in np.round(np.abs(np.diff(ref["d"] ** 2)) / ref["d"][1:] ** 2, 4)]))), flush=True)


def angles(a, b):   # per column: || a sign - b ||  (= 2 sin(theta / 2) for unit vectors)
    s = np.sign((a * b).sum(0))
    return np.linalg.norm(a * s - b, axis=0)


cfgs = [("default (schedule)", dict()),
        ("uniform 16 bit (round 4)", dict(vec_floor=-1.0)),
        ("uniform 24 bit", dict(slices=3, block=16)),
        ("uniform 32 bit, block 8", dict(slices=4, block=8)),
        ("uniform 56 bit, block 4", dict(slices=7, block=4)),
        ("schedule, floor 1e-9 (32 bit early)", dict(vec_floor=1e-9))]
if len(sys.argv) > 4:
    cfgs = [c for c in cfgs if any(t in c[0] for t in sys.argv[4].split(","))]
for name, kw in cfgs:
    r = ba.bed_randomSVD(gb, k=k, **kw)          # (first call: allocations, copies)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = ba.bed_randomSVD(gb, k=k, **kw); ts.append(time.perf_counter() - t0)
    au, av = angles(r["u"], ref["u"]), angles(r["v"], ref["v"])
    h = (k + 1) // 2
    print(json.dumps(dict(cfg=name, ms=round(1e3 * min(ts), 2), niter=r["niter"], nops=r["nops"], block=r["block"],
                          slices=r["slices"], slices_max=r["slices_max"], wide_steps=r["wide_steps"],
                          n_wide=[r["n_wide_cprod"], r["n_wide_prod"]],
                          wide_ms=[round(r["wide_cprod_ms"] / max(1, r["n_wide_cprod"]), 2), round(r["wide_prod_ms"] / max(1, r["n_wide_prod"]), 2)],
                          narrow_ms=[round(r["cprod_ms"] / max(1, r["n_cprod"]), 2), round(r["prod_ms"] / max(1, r["n_prod"]), 2),
                                     round(r["cprod_stats_ms"] / max(1, r["n_cprod_stats"]), 2)],
                          resid=[r["lead_rel_resid"], r["max_rel_resid"]],
                          d_rel=float(np.abs(r["d"] / ref["d"] - 1).max()),
                          u_lead=float(au[:h].max()), u_all=float(au.max()), v_lead=float(av[:h].max()), v_all=float(av.max()),
                          u=[float("%.1e" % x) for x in au], v=[float("%.1e" % x) for x in av])), flush=True)
