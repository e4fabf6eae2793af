#!/usr/bin/env python3
"""Host-API wall time of the secondary §8 rows at realistic sizes (run on the GPU box).

    python tools/probe_rows.py [--n 400000 --m 250000] [--fbm-n 50000 --fbm-m 200000]

Every line is one public entry called the way the reference's R function would be (host vectors in,
host vectors out), so the time includes quantise / finalize kernels and the transfers of the vectors;
GB/s = algorithmic bytes of ONE pass over the image it streams / wall time.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400000)
ap.add_argument("--m", type=int, default=250000)
ap.add_argument("--fbm-n", type=int, default=50000)
ap.add_argument("--fbm-m", type=int, default=200000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--skip-bed", action="store_true")
ap.add_argument("--skip-fbm", action="store_true")
a = ap.parse_args()
L = _lib.load()
ba.selftest()
rng = np.random.default_rng(5)


def timed(fn, reps=a.reps):
    for _ in range(3):       # the first calls of an entry allocate its work buffers
        fn()
    L.bsn_device_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    L.bsn_device_sync()
    return (time.perf_counter() - t0) / reps


def line(image, entry, sec, nbytes, **kw):
    print(json.dumps(dict(image=image, entry=entry, ms=round(sec * 1e3, 3), GBps=round(nbytes / sec / 1e9, 1),
                          frac_of_8TBps=round(nbytes / sec / 8e12, 3), **kw)), flush=True)


for n, m in ([] if a.skip_bed else [(50000, 200000), (a.n, a.m)]):
    gb = ba.bed.synthetic(n, m)
    L.bsn_device_sync()
    nb = ((n + 3) // 4) * m
    tag = "2-bit %dx%d (%.1f GB)" % (n, m, nb / 1e9)
    line(tag, "bed_counts", timed(lambda: ba.bed_counts(gb)), nb)
    line(tag, "bed_colstats", timed(lambda: ba.bed_colstats(gb)), nb)
    sc = ba.bed_scaleBinom(gb)
    x = rng.normal(size=m)
    y = rng.normal(size=n)
    line(tag, "bed_prodVec", timed(lambda: ba.bed_prodVec(gb, x, center=sc["center"], scale=sc["scale"])), nb)
    line(tag, "bed_cprodVec", timed(lambda: ba.bed_cprodVec(gb, y, center=sc["center"], scale=sc["scale"])), nb)
    V = rng.normal(size=(m, 10))
    ir = ba.rows_along(gb)
    ic = ba.cols_along(gb)
    line(tag, "prod_and_rowSumsSq (10 PCs)",
         timed(lambda: ba.prod_and_rowSumsSq(gb, ir, ic, sc["center"], sc["scale"], V)), nb)
    U = rng.normal(size=(n, 10))
    line(tag, "bed_pcadapt statistics (multLinReg, 10 PCs)",
         timed(lambda: ba.multLinReg(gb, ir, ic, U)), nb)
    gb.close()

if not a.skip_fbm:
    n, m = a.fbm_n, a.fbm_m
    # a dosage FBM: bytes 7..207 = the dosage grid 0, 0.01, ..., 2 of CODE_DOSAGE, no missing values
    t0 = time.time()
    base_m = min(m, 20000)
    base = rng.integers(7, 208, size=(base_m, n), dtype=np.uint8)   # variant-major = column-major n x m
    reps = (m + base_m - 1) // base_m
    host = np.concatenate([base] * reps, axis=0)[:m]
    G_bytes = host.T                                                # n x m view, Fortran order
    print("host FBM %.1f GB built in %.1f s" % (host.nbytes / 1e9, time.time() - t0), flush=True)
    t0 = time.perf_counter()
    G = ba.FBM_code256(G_bytes, code=ba.CODE_DOSAGE)
    G.handle
    L.bsn_device_sync()
    up = time.perf_counter() - t0
    nb = n * m
    tag = "byte %dx%d dosage (%.1f GB)" % (n, m, nb / 1e9)
    line(tag, "ingest (host FBM -> byte image)", up, nb)
    line(tag, "snp_colstats", timed(lambda: ba.snp_colstats(G)), nb)
    st = ba.snp_colstats(G)
    center = st["sumX"] / n
    scale = np.sqrt(st["denoX"] / (n - 1))
    x = rng.normal(size=m)
    y = rng.normal(size=n)
    line(tag, "big_prodVec", timed(lambda: ba.big_prodVec(G, x, center=center, scale=scale)), nb)
    line(tag, "big_cprodVec", timed(lambda: ba.big_cprodVec(G, y, center=center, scale=scale)), nb)
    beta = rng.normal(size=m)
    lp = rng.uniform(0, 10, size=m)
    line(tag, "snp_PRS (11 thresholds)",
         timed(lambda: ba.snp_PRS(G, beta, lpS_keep=lp, thr_list=np.linspace(0, 9, 11))), nb)
