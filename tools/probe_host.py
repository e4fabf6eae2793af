#!/usr/bin/env python3
"""Host-side time around one bed_randomSVD call (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import _lib
L = _lib.load()
n, m = 400000, int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
gb = ba.bed.synthetic(n, m)
os.environ["BSN_TIMING"] = "1"
for rep in range(3):
    t0 = time.perf_counter(); st = ba.bed_colstats(gb); t1 = time.perf_counter()
    ms = ba.bed_scaleBinom(gb); t2 = time.perf_counter()
    r = ba.bed_randomSVD(gb, fun_scaling=lambda *a, **k: ms, k=20, return_uv=False, verbose=2 if rep == 2 else 0)
    t3 = time.perf_counter()
    print("colstats %.1f ms  scaleBinom %.1f ms  randomSVD(prescaled) %.1f ms  streaming %.1f ms" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, r["cprod_ms"] + r["prod_ms"]), flush=True)
