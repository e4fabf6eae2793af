#!/usr/bin/env python3
"""what limits u / v of the DEFAULT bed_randomSVD at C3 (400K x 1M, k = 20, tol 1e-4)?

Per solve: the angles || x sign - x_ref || to two tight references (56-bit panels, tol 1e-10, block 4 and block 5), for
  * the default solve on a FRESH handle and again on the warm handle,
  * the same Krylov trajectory on 56-bit panels (slices = 7, block = 16, tol 1e-4): what an fp64 solve stopped at the
    same tol leaves — the comparator the test needs,
  * 32-bit panels at block 16, and the 16-bit solve.
Also prints the gap amplification lam_i / min_j |lam_i - lam_j| of every triplet.  One JSON line per row."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400000)
ap.add_argument("--m", type=int, default=1000000)
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--seeds", default="20250905,7")
a = ap.parse_args()
k, h = a.k, (a.k + 1) // 2


def ang(x, ref):
    s = np.sign(np.sum(x * ref, axis=0))
    return np.linalg.norm(x * s - ref, axis=0)


for seed in [int(s) for s in a.seeds.split(",")]:
    gb = ba.bed.synthetic(a.n, a.m, seed=seed)
    rows = []
    t0 = time.perf_counter(); r1 = ba.bed_randomSVD(gb, k=k); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); r2 = ba.bed_randomSVD(gb, k=k); t2 = time.perf_counter() - t0
    rows.append(("default, fresh handle", r1, t1)); rows.append(("default, warm handle", r2, t2))
    print(json.dumps({"seed": seed, "first_equals_second": bool(np.array_equal(r1["u"], r2["u"]) and np.array_equal(r1["d"], r2["d"]))}), flush=True)
    refs = {}
    for blk in (4, 5):
        t0 = time.perf_counter()
        refs[blk] = ba.bed_randomSVD(gb, k=k, tol=1e-10, slices=7, block=blk)
        print(json.dumps({"seed": seed, "reference_block": blk, "niter": refs[blk]["niter"], "converged": refs[blk]["converged"],
                          "s": round(time.perf_counter() - t0, 2)}), flush=True)
    lam = refs[4]["d"] ** 2
    amp = np.array([lam[i] / np.min(np.abs(lam[i] - np.delete(lam, i))) for i in range(k)])
    print(json.dumps({"seed": seed, "d": [float(x) for x in refs[4]["d"]], "gap_amplification": [round(float(x), 2) for x in amp],
                      "refs_agree_u": float(ang(refs[4]["u"], refs[5]["u"]).max()), "refs_agree_v": float(ang(refs[4]["v"], refs[5]["v"]).max())}), flush=True)
    def show(tag, r, secs):
        out = {"seed": seed, "solve": tag, "s": round(secs, 3), "niter": r["niter"], "slices_max": r["slices_max"], "wide_steps": r["wide_steps"],
               "resid_lead": r["lead_rel_resid"], "resid_all": r["max_rel_resid"]}
        for blk in (4, 5):
            au, av = ang(r["u"], refs[blk]["u"]), ang(r["v"], refs[blk]["v"])
            out["vs_block%d" % blk] = {"u_lead": float(au[:h].max()), "u_all": float(au.max()), "v_lead": float(av[:h].max()), "v_all": float(av.max())}
        au = ang(r["u"], refs[4]["u"])
        out["u_per_vector"] = [float("%.2e" % x) for x in au]
        print(json.dumps(out), flush=True)
    for tag, r, secs in rows:
        show(tag, r, secs)
    for tag, kw in (("56-bit panels, block 16, tol 1e-4 (same trajectory)", dict(slices=7, block=16)),
                    ("32-bit panels, block 16, tol 1e-4", dict(slices=4, block=16)),
                    ("16-bit panels at every step", dict(vec_floor=-1.0)),
                    ("56-bit panels, block 4, tol 1e-4", dict(slices=7, block=4)),
                    ("floor 5e-8", dict(vec_floor=5e-8)),
                    ("floor 1e-6", dict(vec_floor=1e-6))):
        t0 = time.perf_counter()
        try:
            r = ba.bed_randomSVD(gb, k=k, **kw)
        except Exception as e:
            print(json.dumps({"solve": tag, "error": str(e)}), flush=True)
            continue
        try:
            show(tag, r, time.perf_counter() - t0)
        except Exception as e:
            print(json.dumps({"solve": tag, "error": str(e)}), flush=True)
        del r
    gb.close()
