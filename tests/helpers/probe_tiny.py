import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bigsnpr_amd as ba
from oracle import oracle as orc
n, m, k = 39, 17, 16
ob = orc.fake_bed(n, m, seed=487, na16=0)
gb = ba.bed.synthetic(n, m, seed=487, na16=0)
sc = orc.bed_scaleBinom(ob)
ic = np.nonzero(sc["scale"] > 0)[0]
ref = orc.dense_svd(ob, None, ic, k=k)
for block, slices, verbose in ((8, 0, 2), (8, 7, 0), (4, 0, 0), (16, 0, 0), (1, 0, 0)):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = ba.bed_randomSVD(gb, ind_col=ic, k=k, block=block, slices=slices, seed=11, verbose=verbose)
    err = np.abs(res["d"] - ref["d"]) / ref["d"]
    print("block", block, "slices", slices, "converged", res["converged"], "niter", res["niter"], "basis", res["basis"],
          "max rel err %.2e" % err.max(), "resid %.2e" % res["max_rel_resid"], [str(x.message)[:80] for x in w])
