#!/usr/bin/env python3
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import _lib
L = _lib.load()
n, m = 400000, int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
gb = ba.bed.synthetic(n, m)
for rep in range(3):
    t0 = time.perf_counter(); ms = ba.bed_scaleBinom(gb); t1 = time.perf_counter()
    r = ba.bed_randomSVD(gb, fun_scaling=lambda *a, **k: ms, k=20, return_uv=False, verbose=2 if rep == 2 else 0)
    t2 = time.perf_counter()
    stream = r["cprod_ms"] + r["prod_ms"]
    print(json.dumps(dict(scaling_ms=(t1 - t0) * 1e3, svd_call_ms=(t2 - t1) * 1e3, gpu_ms=r["gpu_ms"], streaming_ms=stream,
                          niter=r["niter"], nops=r["nops"])))
