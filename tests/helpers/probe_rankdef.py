import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bigsnpr_amd as ba
from oracle import oracle as orc
n, m0, rep = 300, 12, 5
base = orc.fake_bed(n, m0, seed=9, na16=0)
payload = np.tile(base.payload.reshape(m0, -1), (rep, 1)).reshape(-1)
gb = ba.bed.from_payload(payload, n, m0 * rep)
res = ba.bed_randomSVD(gb, k=20, verbose=2)
print("d1 %.4f" % res["d"][0])
