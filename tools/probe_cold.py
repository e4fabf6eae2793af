#!/usr/bin/env python3
"""the call the reference makes — ONE bed_randomSVD on a freshly opened object — timed in a FRESH process.

Prints one JSON line: library load, image ready (synthetic generation on the device), then the wall time of the
first, second and third solve on the handle and what the library says about the first (BSN_TIMING / BSN_ALLOC_TRACE on
stderr)."""
import argparse, json, os, sys, time
T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400000)
ap.add_argument("--m", type=int, default=1000000)
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--solves", type=int, default=3)
a = ap.parse_args()
import numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import _lib
L = _lib.load()
t_import = time.perf_counter() - T0
ba.selftest()
t_selftest = time.perf_counter() - T0
t0 = time.perf_counter()
gb = ba.bed.synthetic(a.n, a.m)
L.bsn_device_sync()
t_image = time.perf_counter() - t0
times, infos = [], []
for i in range(a.solves):
    t0 = time.perf_counter()
    r = ba.bed_randomSVD(gb, k=a.k)
    times.append(1e3 * (time.perf_counter() - t0))
    infos.append({"niter": r["niter"], "gpu_ms": r["gpu_ms"], "tiled": r["tiled"], "n_prod": r["n_prod"], "n_wide_prod": r["n_wide_prod"],
                  "prod_ms": r["prod_ms"], "wide_prod_ms": r["wide_prod_ms"], "sigma1": float(r["d"][0])})
    del r
print(json.dumps({"import_s": t_import, "selftest_s": t_selftest - t_import, "image_ready_s": t_image, "solve_ms": times, "info": infos}), flush=True)
