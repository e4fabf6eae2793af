#!/usr/bin/env python3
"""bed_randomSVD with and without the download of u (n x k) and v (m x k) to host memory."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bigsnpr_amd as ba
n, m = 400000, int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
gb = ba.bed.synthetic(n, m)
for uv in (False, True, False, True):
    t0 = time.perf_counter(); r = ba.bed_randomSVD(gb, k=20, return_uv=uv); t1 = time.perf_counter()
    print("return_uv=%s: %.1f ms" % (uv, (t1 - t0) * 1e3), flush=True)
