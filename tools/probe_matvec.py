#!/usr/bin/env python3
"""Times the streaming kernels on a synthetic matrix (run on the GPU box)."""
import argparse, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400000)
ap.add_argument("--m", type=int, default=100000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--nvecs", type=str, default="1,4,8")
ap.add_argument("--slices", type=int, default=4)
ap.add_argument("--subset", type=float, default=0.0, help="random sorted subset of this fraction of the variants (non-contiguous ind.col)")
ap.add_argument("--xkind", default="normal", help="normal | ones (a 0/1 panel: constant digits, low operand toggling)")
ap.add_argument("--dosage", action="store_true", help="byte image of a dosage FBM (uploaded from the host) instead of the 2-bit image")
a = ap.parse_args()
L = _lib.load()
ba.selftest()
t0 = time.time()
if a.dosage:
    rng0 = np.random.default_rng(3)
    base_m = min(a.m, 10000)
    base = rng0.integers(7, 208, size=(base_m, a.n), dtype=np.uint8)      # CODE_DOSAGE grid 0, 0.01, ..., 2
    host = np.concatenate([base] * ((a.m + base_m - 1) // base_m), axis=0)[:a.m]
    G = ba.FBM_code256(host.T, code=ba.CODE_DOSAGE)
    gb = G._bed
    bytes_pass = a.n * a.m
else:
    gb = ba.bed.synthetic(a.n, a.m)
    bytes_pass = ((a.n + 3) // 4) * a.m
if os.environ.get("BSN_PROBE_TILE") and not a.dosage:
    print("tiled copy:", gb.tile(), flush=True)
L.bsn_device_sync()
print("generate %.2fs, image %.2f GB" % (time.time() - t0, gb.hbm_bytes() / 1e9), flush=True)

def timed(fn, reps):
    fn(); L.bsn_device_sync()
    ms = C.c_double()
    L.bsn_timer_start(gb.handle)
    for _ in range(reps):
        fn()
    L.bsn_timer_stop(gb.handle, C.byref(ms))
    return ms.value / reps

if a.dosage:
    st = ba.snp_colstats(G)
    sc = dict(center=st["sumX"] / a.n, scale=np.sqrt(st["denoX"] / (a.n - 1)))
else:
    t = timed(lambda: ba.bed_counts(gb), 1)
    print(json.dumps(dict(kernel="counts(host api)", ms=t, GBps=bytes_pass / t / 1e6)), flush=True)
    sc = ba.bed_scaleBinom(gb)
rng = np.random.default_rng(0)
ic = None
if a.subset > 0:
    ic = np.sort(rng.choice(a.m, int(a.m * a.subset), replace=False))
    sc = dict(center=sc["center"][ic], scale=sc["scale"][ic])
    bytes_pass = bytes_pass // a.m * ic.size
    a.m = ic.size
    print("subset of %d variants" % ic.size, flush=True)
op = ba.ScaledOp(gb, None, ic, sc["center"], sc["scale"], slices=a.slices)
for nv in [int(v) for v in a.nvecs.split(",")]:
    if a.xkind == "ones":
        X = ba.DeviceArray.from_numpy((rng.random(size=(a.m, nv)) < 0.5).astype(float))
        R = ba.DeviceArray.from_numpy((rng.random(size=(a.n, nv)) < 0.5).astype(float))
    else:
        X = ba.DeviceArray.from_numpy(rng.normal(size=(a.m, nv)))
        R = ba.DeviceArray.from_numpy(rng.normal(size=(a.n, nv)))
    Y = ba.DeviceArray(a.n, nv); Z = ba.DeviceArray(a.m, nv)
    t = timed(lambda: op.cprod(R, Z), a.reps)
    import hashlib
    print(json.dumps(dict(kernel="cprod", nvec=nv, ms=t, GBps=bytes_pass / t / 1e6,
                          sha=hashlib.sha1(Z.to_numpy().tobytes()).hexdigest()[:12])), flush=True)
    t = timed(lambda: op.prod(X, Y), a.reps)
    print(json.dumps(dict(kernel="prod", nvec=nv, ms=t, GBps=bytes_pass / t / 1e6,
                          sha=hashlib.sha1(Y.to_numpy().tobytes()).hexdigest()[:12])), flush=True)
