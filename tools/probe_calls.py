#!/usr/bin/env python3
"""Wall time per call of the one-shot entries on a 2-bit image (default: BASELINE config 2, 50 000 x 200 000);
run it under `rocprofv3 --kernel-trace --stats` to see how much of each call is kernel time."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--m", type=int, default=200000)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default="")
a = ap.parse_args()
L = _lib.load()
gb = ba.bed.synthetic(a.n, a.m)
sc = ba.bed_scaleBinom(gb)
rng = np.random.default_rng(1)
x, y = rng.normal(size=a.m), rng.normal(size=a.n)
entries = {
    "bed_counts": lambda: ba.bed_counts(gb),
    "bed_prodVec": lambda: ba.bed_prodVec(gb, x, center=sc["center"], scale=sc["scale"]),
    "bed_cprodVec": lambda: ba.bed_cprodVec(gb, y, center=sc["center"], scale=sc["scale"]),
}
nb = ((a.n + 3) // 4) * a.m
for name, fn in entries.items():
    if a.only and name != a.only:
        continue
    fn(); fn()
    L.bsn_device_sync()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    dt = (time.perf_counter() - t0) / a.reps
    print(json.dumps(dict(entry=name, ms=round(dt * 1e3, 3), GBps=round(nb / dt / 1e9, 1))), flush=True)
