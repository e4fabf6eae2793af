#!/usr/bin/env python3
"""snp_autoSVD end to end on a synthetic 2-bit image with 22 chromosomes (run on the GPU box):
where the time of the whole pipeline goes (MAF filter, clumping, SVD rounds + outlier detection)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import ld as ldm, autosvd

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400000)
ap.add_argument("--m", type=int, default=250000)
ap.add_argument("--bed", action="store_true", help="bed_autoSVD (1 %% missing values: the six-product LD kernels in the clumping) instead of snp_autoSVD")
a = ap.parse_args()
gb = ba.bed.synthetic(a.n, a.m)
chrom = np.repeat(np.arange(1, 23), (a.m + 21) // 22)[:a.m]
pos = np.arange(a.m) * 2000.0                       # 250 variants per 500-kb window side
T = {}
def timed(name, fn):
    def w(*x, **k):
        t0 = time.perf_counter(); r = fn(*x, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
# inside snp_clumping (round 6): the per-chromosome column statistics, the clumping call, and of it the HIP-event time of the
# pair-statistics kernels (bsn_ld_last_stats)
K = {}
_clump_chr0 = ldm._clump_chr
def _clump_chr(*x, **k):
    t0 = time.perf_counter(); r = _clump_chr0(*x, **k); K["clumping_chr calls"] = K.get("clumping_chr calls", 0.0) + time.perf_counter() - t0
    st = ldm.last_stats()
    K["of which pair-statistics kernels"] = K.get("of which pair-statistics kernels", 0.0) + st["stats_ms"] * 1e-3
    K["kernel"] = st["kernel"][:40]
    return r
ldm._clump_chr = _clump_chr
_colstats0 = ldm.snp_colstats
def _colstats(*x, **k):
    t0 = time.perf_counter(); r = _colstats0(*x, **k); K["snp_colstats calls"] = K.get("snp_colstats calls", 0.0) + time.perf_counter() - t0
    return r
ldm.snp_colstats = _colstats
if a.bed:
    gb._map = dict(chromosome=chrom, physical_pos=pos)   # (a synthetic handle has no .bim)
    from bigsnpr_amd import bed as bedm
    _clump0 = ldm._clump_chr
    autosvd.bed_MAF = timed("bed_MAF", autosvd.bed_MAF)
    autosvd.bed_clumping = timed("bed_clumping", autosvd.bed_clumping)
    autosvd.bed_randomSVD = timed("bed_randomSVD", autosvd.bed_randomSVD)
    _bcs0 = ldm.bed_colstats
    def _bcs(*x, **k):
        t0 = time.perf_counter(); r = _bcs0(*x, **k); K["bed_colstats calls"] = K.get("bed_colstats calls", 0.0) + time.perf_counter() - t0
        return r
    ldm.bed_colstats = _bcs
autosvd.snp_MAF = timed("snp_MAF", autosvd.snp_MAF)
autosvd.snp_clumping = timed("snp_clumping", autosvd.snp_clumping)
autosvd.big_randomSVD = timed("big_randomSVD", autosvd.big_randomSVD)
autosvd.dist_ogk = timed("dist_ogk", autosvd.dist_ogk)
autosvd.rollmean_groups = timed("rollmean", autosvd.rollmean_groups)
autosvd.tukey_mc_up = timed("tukey_mc_up", autosvd.tukey_mc_up)
for rep in ("first call (imports, first launches, allocations)", "second call", "third call"):
    T.clear(); K.clear()
    t0 = time.perf_counter()
    res = (ba.bed_autoSVD(gb, k=10, verbose=rep.startswith("first")) if a.bed
           else ba.snp_autoSVD(gb, chrom, pos, k=10, verbose=rep.startswith("first")))
    tot = time.perf_counter() - t0
    print("%s: total %.3f s; %s; rest of the host loop %.3f s; kept %d of %d variants"
          % (rep, tot, ", ".join("%s %.3f s" % kv for kv in T.items()), tot - sum(T.values()), res["subset"].size, a.m))
    print("    inside snp_clumping: " + ", ".join("%s %s" % (k_, ("%.3f s" % v) if isinstance(v, float) else v) for k_, v in K.items()))
