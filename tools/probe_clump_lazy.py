#!/usr/bin/env python3
"""Wide-window clumping at config C5 scale (400K x 100K): the dense band against the batched
candidate-vs-kept path (forced through BSN_CLUMP_BAND_BUDGET), same kept set required."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
n, m = 400000, int(sys.argv[1]) if len(sys.argv) > 1 else 100000
gb = ba.bed.synthetic(n, m)
chr_ = np.ones(m, dtype=np.int64)
pos = 1000.0 * np.arange(m)
for size, thr in ((5000, 0.01), (20000, 0.01), (5000, 0.05)):
    out = {}
    for mode in ("dense", "lazy"):
        if mode == "lazy":
            os.environ["BSN_CLUMP_BAND_BUDGET"] = "1e6"
        else:
            os.environ.pop("BSN_CLUMP_BAND_BUDGET", None)
        t0 = time.perf_counter()
        keep = ba.bed_clumping(gb, thr_r2=thr, size=size, infos_chr=chr_, infos_pos=pos)
        out[mode] = (time.perf_counter() - t0, keep)
    same = np.array_equal(out["dense"][1], out["lazy"][1])
    print("bed_clumping window %d variants, thr %.2f: dense %.2f s, lazy %.2f s, kept %d, identical %s"
          % (2 * size, thr, out["dense"][0], out["lazy"][0], out["dense"][1].size, same), flush=True)
