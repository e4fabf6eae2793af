import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bigsnpr_amd as ba
from oracle import oracle as orc
ob2 = orc.fake_bed(1500, 4000, seed=21)
gb2 = ba.bed.synthetic(1500, 4000, seed=21)
sc = orc.bed_scaleBinom(ob2)
ic = np.nonzero(sc["scale"] > 0)[0]
ref2 = orc.dense_svd(ob2, None, ic, k=20)
for kw in (dict(block=16, max_basis=96), dict(block=8, max_basis=64), dict(block=8, max_basis=64, slices=7, tol=1e-8)):
    res = ba.bed_randomSVD(gb2, ind_col=ic, k=20, verbose=1, **kw)
    print(kw, res["converged"], res["niter"], res["basis"], "err %.2e" % np.abs(res["d"] / ref2["d"] - 1).max(), flush=True)
