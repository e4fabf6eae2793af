#!/usr/bin/env python3
"""Times the windowed-LD path at BASELINE config C5 scale (one chromosome, 400K x 100K,
window ~2000 variants)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
gb = ba.bed.synthetic(n, m)
pos = np.arange(m, dtype=np.float64)
for rep in range(2):
    t0 = time.perf_counter(); ld = ba.bed_ld_scores(gb, size=W / 1000.0, infos_pos=pos); t1 = time.perf_counter()
    pairs = m * W - W * (W + 1) / 2
    print(json.dumps(dict(op="bed_ld_scores", s=t1 - t0, pairs=pairs, pair_per_s=pairs / (t1 - t0),
                          int8_TOPS=pairs * n * 6 * 2 / (t1 - t0) / 1e12, ld_mean=float(ld.mean()))), flush=True)
t0 = time.perf_counter(); c = ba.bed_cor(gb, size=W / 1000.0, infos_pos=pos, thr_r2=0.01); t1 = time.perf_counter()
print(json.dumps(dict(op="bed_cor thr_r2=0.01", s=t1 - t0, nnz=int(c.p[-1]))), flush=True)
