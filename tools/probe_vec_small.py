#!/usr/bin/env python3
"""the oracle's small shapes, where the leading vectors of the default solve sit at 1.5 - 1.8e-6.  Per shape:
angles of the leading half to the oracle's dense SVD for the default, for other floors of the precision schedule and for
uniform panels at the DEFAULT's block (the same Krylov trajectory)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
from oracle import oracle as orc

def ang(ref, x, k):
    s = np.sign(np.sum(x * ref[:, :k], axis=0))
    return np.linalg.norm(x * s - ref[:, :k], axis=0)

gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cases = [("example", ba.bed(os.path.join(gold, "example.bed")), orc.BedFile(os.path.join(gold, "example.bed")), None, 10)]
for n, m, k in ((1500, 4000, 20), (3000, 900, 10)):
    ob, gb = orc.fake_bed(n, m, seed=21), ba.bed.synthetic(n, m, seed=21)
    cases.append(("synth_%dx%d" % (n, m), gb, ob, np.nonzero(orc.bed_scaleBinom(ob)["scale"] > 0)[0], k))
for name, gb, ob, ic, k in cases:
    ref = orc.dense_svd(ob, None, ic, k=k + 1)
    lam = ref["d"] ** 2
    amp = np.array([lam[i] / np.min(np.abs(lam[i] - np.delete(lam, i))) for i in range(k)])
    h = (k + 1) // 2
    d0 = ba.bed_randomSVD(gb, ind_col=ic, k=k)
    blk = d0["block"]
    print(json.dumps({"case": name, "k": k, "block": blk, "amp_lead": [round(float(a), 1) for a in amp[:h]]}))
    for tag, kw in [("default", {})] + [("floor %g" % f, dict(vec_floor=f)) for f in (1e-7, 5e-8, 2e-8, 5e-9, 1e-9)] + \
                   [("uniform %d bit" % (8 * s), dict(slices=s, block=blk)) for s in (2, 3, 4, 5, 7)]:
        r = ba.bed_randomSVD(gb, ind_col=ic, k=k, **kw)
        au, av = ang(ref["u"], r["u"], k), ang(ref["v"], r["v"], k)
        print(json.dumps({"case": name, "solve": tag, "niter": r["niter"], "slices_max": r["slices_max"], "wide_steps": r["wide_steps"],
                          "resid_lead": float("%.2e" % r["lead_rel_resid"]), "u_lead": float("%.2e" % au[:h].max()), "v_lead": float("%.2e" % av[:h].max()),
                          "u_lead_each": [float("%.1e" % x) for x in au[:h]]}), flush=True)
