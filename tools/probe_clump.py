#!/usr/bin/env python3
"""Times bed_clumping / snp_grid_clumping-style work at config C5 scale (400K x 100K)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
n, m = 400000, int(sys.argv[1]) if len(sys.argv) > 1 else 100000
gb = ba.bed.synthetic(n, m)
chr_ = np.ones(m, dtype=np.int64)
pos = 1000.0 * np.arange(m)
for size in (100, 500, 2000):
    t0 = time.perf_counter()
    keep = ba.bed_clumping(gb, thr_r2=0.2, size=size, infos_chr=chr_, infos_pos=pos)
    print("bed_clumping size=%d kb (window %d variants): %.2f s, kept %d" % (size, size, time.perf_counter() - t0, keep.size), flush=True)
