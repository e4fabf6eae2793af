#!/usr/bin/env python3
"""bed_tcrossprodSelf wall time and fp64 rate (run on the GPU box): python tools/probe_tcross.py --n 16384 --m 65536"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--m", type=int, default=65536)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
L = _lib.load()
gb = ba.bed.synthetic(a.n, a.m)
sc = ba.bed_scaleBinom(gb)
fs = lambda obj, ind_row, ind_col, ncores=1: sc
K, _ = ba.bed_tcrossprodSelf(gb, fun_scaling=fs)
t0 = time.perf_counter()
for _ in range(a.reps):
    K, _ = ba.bed_tcrossprodSelf(gb, fun_scaling=fs)
dt = (time.perf_counter() - t0) / a.reps
flops = 2.0 * a.n * a.n * a.m
print(json.dumps(dict(entry="bed_tcrossprodSelf", n=a.n, m=a.m, s=round(dt, 4), TFLOPs_full=round(flops / dt / 1e12, 2),
                      frac_of_78_6=round(flops / dt / 78.6e12, 3), trace=float(np.trace(K)))), flush=True)
